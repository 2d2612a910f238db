// how fast can one file take N GB from T threads on this box: buffered pwrite() of disjoint ranges against memcpy into a shared
// mapping (g++ -O2 -pthread tools/dbg/write_bench.cpp -o /tmp/write_bench; /tmp/write_bench <dir> <GB>)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 4.0) << 30;
    std::vector<char> src(256u << 20);
    for (size_t i = 0; i < src.size(); i++) src[i] = (char)(i * 131u);
    using clk = std::chrono::steady_clock;
    for (int mode = 0; mode < 4; mode++) for (int T : {1, 4, 8, 16, 32}) {
        const std::string path = dir + "/write_bench.bin";
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0666);
        const auto t0 = clk::now();
        char* m = nullptr;
        if (mode >= 1) {
            if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
            if (mode == 2 && posix_fallocate(fd, 0, (off_t)total) != 0) { perror("fallocate"); }
            if (mode != 3) { m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (m == MAP_FAILED) { perror("mmap"); return 1; } }
        }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            const size_t a = total * t / T, b = total * (t + 1) / T;
            for (size_t o = a; o < b;) {
                const size_t n = std::min<size_t>(b - o, 64u << 20);
                if (mode == 0 || mode == 3) { size_t d = 0; while (d < n) { ssize_t w = pwrite(fd, src.data() + d, n - d, (off_t)(o + d)); if (w <= 0) { perror("pwrite"); return; } d += (size_t)w; } }
                else memcpy(m + o, src.data(), n);
                o += n;
            }
        });
        for (auto& x : th) x.join();
        if (m) munmap(m, total);
        close(fd);
        const double s = std::chrono::duration<double>(clk::now() - t0).count();
        printf("%-28s T=%2d  %.2f GB/s\n", mode == 0 ? "pwrite (growing file)" : mode == 1 ? "mmap (ftruncate'd hole)" : mode == 2 ? "mmap (fallocate'd)" : "pwrite (ftruncate'd hole)", T, total / s / 1e9);
        fflush(stdout);
        unlink(path.c_str());
    }
    return 0;
}
