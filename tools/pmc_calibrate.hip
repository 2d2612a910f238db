// pmc_calibrate.hip -- known-byte-count kernels in the access patterns of the FCZ kernels, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: only the wide coalesced read is
// calibrated there; "calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate tools/pmc_calibrate.hip
//   rocprofv3 --kernel-include-regex fczcal --pmc FETCH_SIZE -- /tmp/pmc_calibrate
// Every kernel moves exactly N_BYTES (1 GiB) in and N_BYTES out; buffers are 2 x 1 GiB (>> 256 MiB L3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr size_t N_BYTES = 1ull << 30;

// 16 B per lane, fully coalesced (k_compress_tiled atom loads / flush stores)
__global__ void fczcal_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 4 B per lane, coalesced (k_sidechain coordinate stores are 4 B, nearly contiguous across lanes)
__global__ void fczcal_copy4(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// every lane streams its own 2800-byte record with 8 B loads/stores (k_backbone: lane = chain, packed words)
__global__ void fczcal_lane_stream8(const uint64_t* __restrict__ a, uint64_t* __restrict__ b, size_t n_rec) {
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    for (int k = 0; k < 350; k++) b[r * 350 + k] = a[r * 350 + k];
}
// 4 B per lane with a 100-byte lane stride (k_sidechain: lane = residue, one atom component per store)
__global__ void fczcal_stride100(const uint8_t* __restrict__ a, uint8_t* __restrict__ b, size_t n_rec) {
    const size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    for (int k = 0; k < 25; k++) {
        uint32_t v; __builtin_memcpy(&v, a + r * 100 + 4 * k, 4); __builtin_memcpy(b + r * 100 + 4 * k, &v, 4);
    }
}

int main() {
    void *a, *b;
    if (hipMalloc(&a, N_BYTES) != hipSuccess || hipMalloc(&b, N_BYTES) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, N_BYTES); hipMemset(b, 0, N_BYTES);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(fczcal_copy16, dim3(8192), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, N_BYTES / 16);
        hipLaunchKernelGGL(fczcal_copy4, dim3(8192), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, N_BYTES / 4);
        const size_t n8 = N_BYTES / 2800;
        hipLaunchKernelGGL(fczcal_lane_stream8, dim3((n8 + 63) / 64), dim3(64), 0, 0, (const uint64_t*)a, (uint64_t*)b, n8);
        const size_t n100 = N_BYTES / 100;
        hipLaunchKernelGGL(fczcal_stride100, dim3((n100 + 255) / 256), dim3(256), 0, 0, (const uint8_t*)a, (uint8_t*)b, n100);
    }
    hipDeviceSynchronize();
    printf("bytes_per_kernel %zu %zu %zu %zu\n", N_BYTES, N_BYTES, (N_BYTES / 2800) * 2800, (N_BYTES / 100) * 100);
    return 0;
}
