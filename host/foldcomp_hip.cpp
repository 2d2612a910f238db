// foldcomp_hip.cpp -- the C++ host of the MI355X codec: a `foldcomp`-style command line over the C-ABI of include/fcz_hip.h.
//
//   foldcomp-hip compress   [-t threads] [--gpus N] [-b N] [-y] [-r] [-d|-z] [--skip-discontinuous] [--json-stats] <pdb|cif file|dir|tar(.gz)|db> [<fcz file|dir|tar|db>]
//   foldcomp-hip decompress [--gpus N] [-a] [-y] [-r] [-d|-z] [--json-stats] <fcz file|dir|tar(.gz)|db> [<pdb file|dir|tar|db>]
//   foldcomp-hip extract    [--plddt|--fasta|--amino-acid] [-p digits] [--use-title] [--no-merge] [-d|-z] [-r] <fcz file|dir|tar(.gz)|db> [<out>]
//   foldcomp-hip check      [-r] <fcz file|dir|tar(.gz)|db>
//   foldcomp-hip rmsd       <pdb|cif> <pdb|cif>
//   no GPU needed (used by the tests):
//   foldcomp-hip dump-batch [-b N] <pdb file>          the host-side batch of a file as text
//   foldcomp-hip plan-dump [--shard R/N] <input>       the items of a run's input (files, database entries, tar members), one per line
//   foldcomp-hip db-pack <dir> <db> / db-unpack <db> <dir>   files <-> database container
//   foldcomp-hip tar-pack <dir> <tar>                  files -> tar archive with the reference writer's headers
//   foldcomp-hip db-splice --shard R/N [--key0 K --off0 B] <part db> <final db>   exchange step of a sharded run (see run_db_splice)
//   sharded runs:  compress|decompress -d --shard R/N --device D ...   rank R of N takes its byte-balanced range of the inputs
//
// It mirrors the reference's own driver (src/main.cpp:438-536 compress, :612-689 decompress, :780-795 extract, :912-926
// check) for PDB text files and directories: structures are parsed on the host threads, fragments (one chain without gaps
// = one FCZ record, src/main.cpp:465-508) are batched, and every batch makes ONE trip through the GPU
// (fcz_compress_batch / fcz_decompress_pdb_* / fcz_extract). Host logic restated here, each with its reference:
//   fixed-column ATOM parser            foldcomp/foldcomp.cxx:259-278, gemmi number parsing = strtod -> float
//   removeAlternativePosition           src/atom_coordinate.cpp:362-370
//   identifyChains                      src/atom_coordinate.cpp:469-497
//   identifyDiscontinousResInd          src/atom_coordinate.cpp:506-530
//   splitAtomByResidue                  src/atom_coordinate.cpp:304-328
//   getFileParts / isCompressible       src/utility.cpp:118-140
//   database container (-d)            src/database_reader.cpp, src/database_writer.cpp
//   mmCIF _atom_site loop, .gz         src/structure_reader.cpp:31-61 (gemmi), zlib
//   tar archives in and out (-z)       src/input_processor.h:109-198 (TarProcessor), lib/microtar, src/main.cpp:333-342, :519-523, :666-674
#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <climits>
#include <unordered_map>
#include <unordered_set>
#include <string_view>
#include <zlib.h>
#include <omp.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

extern "C" {
#include "../include/fcz_hip.h"
#include "../include/fcz_host.h"
}

namespace {

constexpr size_t BATCH_CHAINS = 16384;
constexpr size_t JOB_CHAINS_MAX = 4096;     // fragments per job of the compress pipeline: bounds the host memory in flight

// ---- atoms of one input file (the reference's std::vector<AtomCoordinate>, as parallel arrays) ----
// Atom and residue names are kept as packed integers: up to four characters, one per byte (a PDB name field is four columns
// wide, a residue name three); a longer name (mmCIF allows it) is interned in the table's side list and carries 0xff in the top
// byte, which no ASCII name has. Equal names <=> equal integers inside one table.
inline uint32_t pack_name(const char* s, size_t n) { uint32_t v = 0; for (size_t i = 0; i < n; i++) v |= (uint32_t)(unsigned char)s[i] << (8 * i); return v; }
constexpr uint32_t PK_N = 'N', PK_CA = 'C' | ('A' << 8), PK_C = 'C';
struct AtomTable {
    std::vector<uint32_t> atom, residue;          // packed names
    std::vector<std::string> long_names;          // names of more than four characters
    std::vector<char> chain;
    std::vector<uint16_t> chain_key;               // the chain NAME (columns 21-22 trimmed, up to two characters packed) where the reader
                                                   // that filled the table knows it; empty = `chain` is the whole name
    std::vector<std::string> chain_names;          // mmCIF: chain names of any length; chain_key then indexes this list
    std::vector<int> atom_index, res_index;
    std::vector<float> x, y, z, bfac;
    // filled by the parse threads (name -> code once per atom, in parallel): atom code (fcz_atom_code_from_name) and the
    // residue code of the atom's residue name (fcz_res_code_from_name, -1 = not a name the codec takes); empty = not computed
    std::vector<uint8_t> atom_code;
    std::vector<int8_t> res_code;
    size_t size() const { return atom.size(); }
    uint32_t intern(const char* s, size_t n) {
        if (n <= 4) return pack_name(s, n);
        for (size_t i = 0; i < long_names.size(); i++) if (long_names[i].size() == n && memcmp(long_names[i].data(), s, n) == 0) return 0xff000000u | (uint32_t)i;
        long_names.emplace_back(s, n);
        return 0xff000000u | (uint32_t)(long_names.size() - 1);
    }
    uint32_t intern(const std::string& s) { return intern(s.data(), s.size()); }
    bool same_chain(size_t i, size_t j) const { return chain_key.size() == size() ? chain_key[i] == chain_key[j] : chain[i] == chain[j]; }
    uint16_t intern_chain(const std::string& nm) {
        for (size_t i = 0; i < chain_names.size(); i++) if (chain_names[i] == nm) return (uint16_t)i;
        chain_names.push_back(nm);
        return (uint16_t)(chain_names.size() - 1);
    }
    std::string chain_name(size_t i) const {       // what the reference appends to a record's name (src/main.cpp:489-491)
        if (chain_key.size() != size()) return std::string(1, chain[i]);
        if (!chain_names.empty()) return chain_names[chain_key[i]];
        std::string o; if (chain_key[i] & 0xff) o.push_back((char)(chain_key[i] & 0xff)); if (chain_key[i] >> 8) o.push_back((char)(chain_key[i] >> 8)); return o;
    }
    std::string name(uint32_t pk) const {
        if ((pk >> 24) == 0xffu) return long_names[pk & 0xffffffu];
        std::string o;
        for (int i = 0; i < 4 && ((pk >> (8 * i)) & 0xffu); i++) o.push_back((char)((pk >> (8 * i)) & 0xffu));
        return o;
    }
    void clear() {   // keeps every capacity (TablePool)
        atom.clear(); residue.clear(); long_names.clear(); chain.clear(); chain_key.clear(); chain_names.clear(); atom_index.clear(); res_index.clear();
        x.clear(); y.clear(); z.clear(); bfac.clear(); atom_code.clear(); res_code.clear();
    }
    void reserve(size_t n) {
        atom.reserve(n); residue.reserve(n); chain.reserve(n); atom_index.reserve(n); res_index.reserve(n);
        x.reserve(n); y.reserve(n); z.reserve(n); bfac.reserve(n); atom_code.reserve(n); res_code.reserve(n);
    }
    AtomTable slice(size_t a, size_t b) const {
        AtomTable t;
        t.long_names = long_names; t.chain_names = chain_names;
        t.atom.assign(atom.begin() + a, atom.begin() + b); t.residue.assign(residue.begin() + a, residue.begin() + b);
        t.chain.assign(chain.begin() + a, chain.begin() + b);
        if (chain_key.size() == size()) t.chain_key.assign(chain_key.begin() + a, chain_key.begin() + b);
        t.atom_index.assign(atom_index.begin() + a, atom_index.begin() + b); t.res_index.assign(res_index.begin() + a, res_index.begin() + b);
        t.x.assign(x.begin() + a, x.begin() + b); t.y.assign(y.begin() + a, y.begin() + b); t.z.assign(z.begin() + a, z.begin() + b);
        t.bfac.assign(bfac.begin() + a, bfac.begin() + b);
        if (atom_code.size() == size()) { t.atom_code.assign(atom_code.begin() + a, atom_code.begin() + b); t.res_code.assign(res_code.begin() + a, res_code.begin() + b); }
        return t;
    }
    void push_from(const AtomTable& o, size_t i) {   // o shares this table's long_names (remove_alternative_position copies them first)
        atom.push_back(o.atom[i]); residue.push_back(o.residue[i]); chain.push_back(o.chain[i]);
        if (o.chain_key.size() == o.size()) chain_key.push_back(o.chain_key[i]);
        atom_index.push_back(o.atom_index[i]); res_index.push_back(o.res_index[i]);
        x.push_back(o.x[i]); y.push_back(o.y[i]); z.push_back(o.z[i]); bfac.push_back(o.bfac[i]);
        if (o.atom_code.size() == o.size()) { atom_code.push_back(o.atom_code[i]); res_code.push_back(o.res_code[i]); }
    }
};

// Tables go round: the parse threads take an emptied table (with the capacity it grew to) instead of allocating eleven
// vectors per file, the compress workers hand the tables of a finished job back. Fresh allocations of that size by hundreds of
// threads (heap growth by mprotect, first-touch page faults) all take the process's address-space lock.
struct TablePool {
    std::mutex m; std::vector<AtomTable> free_;
    AtomTable get() {
        std::lock_guard<std::mutex> g(m);
        if (free_.empty()) return AtomTable();
        AtomTable t = std::move(free_.back()); free_.pop_back();
        return t;
    }
    void put(AtomTable&& t) {
        t.clear();
        std::lock_guard<std::mutex> g(m);
        if (free_.size() < 4 * JOB_CHAINS_MAX) free_.push_back(std::move(t));   // beyond that the tables are freed, not hoarded
    }
};
TablePool& table_pool() { static TablePool p; return p; }

std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
bool ends_with(const std::string& s, const std::string& p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }

int parse_int(const std::string& f) {
    const std::string s = strip(f);
    char* end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    if (s.empty() || *end) throw std::runtime_error("invalid integer field '" + f + "'");
    return (int)v;
}
float parse_float(const std::string& f) {   // text -> double -> float, as gemmi / the Python binding do
    const std::string s = strip(f);
    char* end = nullptr;
    const double v = strtod(s.c_str(), &end);
    if (s.empty() || *end) throw std::runtime_error("invalid number field '" + f + "'");
    return (float)v;
}


// ---- the same parser over the raw file image, without a string per line or per field (the parse threads are what bounds a
//      disk -> database run: 240 KB of text per 350-residue chain against 20 us of GPU time). Number fields take the exact
//      fast path of decimal -> double (integer mantissa / power of ten, both exact in double, one correctly rounded division
//      = strtod's result for up to 15 digits) and fall back to strtod for anything else. ----
inline bool fast_decimal(const char* p, const char* e, double& out) {
    while (p < e && (*p == ' ' || *p == '\t')) p++;
    while (e > p && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) e--;
    if (p == e) return false;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = *p == '-'; p++; }
    uint64_t m = 0; int digits = 0, frac = 0; bool dot = false;
    for (; p < e; p++) {
        if (*p >= '0' && *p <= '9') { m = m * 10 + (uint64_t)(*p - '0'); digits++; if (dot) frac++; }
        else if (*p == '.' && !dot) dot = true;
        else return false;
    }
    if (digits == 0 || digits > 15) return false;
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const double v = (double)m / p10[frac];
    out = neg ? -v : v;
    return true;
}
// The layout every writer of the format uses for a number field: right-aligned, fixed decimals ("%8.3f" coordinates, "%6.2f"
// B-factors). W = field width, F = decimals. The float is (float)(m / 10^F) with m the digits as an integer; for m < 10^8 and
// F <= 3, (float)(m * 10^-F) is the same float (the product is within one double ulp of the quotient, and a quotient with
// 125 not dividing m is at least 2^-25 / 10^F away, relatively, from every float rounding boundary; checked exhaustively for
// F = 2, 3), so the division becomes a multiplication. Anything else about the field -> false, the general path decides.
template <int W, int F>
inline bool fixed_field(const char* f, float& out) {
    constexpr int IP = W - F - 1;                      // characters before the point
    if (f[IP] != '.') return false;
    uint32_t frac = 0;
    for (int i = 0; i < F; i++) { const unsigned c = (unsigned)(f[IP + 1 + i] - '0'); if (c > 9u) return false; frac = frac * 10u + c; }
    int i = 0;
    while (i < IP && f[i] == ' ') i++;
    bool neg = false;
    if (i < IP && f[i] == '-') { neg = true; i++; }
    if (i >= IP) return false;                         // no digit before the point
    uint32_t ip = 0;
    for (; i < IP; i++) { const unsigned c = (unsigned)(f[i] - '0'); if (c > 9u) return false; ip = ip * 10u + c; }
    constexpr uint32_t P10 = F == 3 ? 1000u : (F == 2 ? 100u : (F == 1 ? 10u : 1u));
    constexpr double INV = F == 3 ? 0.001 : (F == 2 ? 0.01 : (F == 1 ? 0.1 : 1.0));
    const float v = (float)((double)(ip * P10 + frac) * INV);
    out = neg ? -v : v;
    return true;
}
inline float field_float(const char* line, size_t len, size_t a, size_t b) {
    if (a >= len) throw std::runtime_error("invalid number field ''");
    b = std::min(b, len);
    double v;
    if (fast_decimal(line + a, line + b, v)) return (float)v;
    return parse_float(std::string(line + a, b - a));
}
inline int field_int(const char* line, size_t len, size_t a, size_t b) {
    if (a >= len) throw std::runtime_error("invalid integer field ''");
    b = std::min(b, len);
    const char* p = line + a; const char* e = line + b;
    while (p < e && *p == ' ') p++;
    while (e > p && (e[-1] == ' ' || e[-1] == '\r')) e--;
    bool neg = false; const char* q = p;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    long v = 0; bool ok = q < e;
    for (; q < e; q++) { if (*q < '0' || *q > '9') { ok = false; break; } v = v * 10 + (*q - '0'); }
    if (!ok) return parse_int(std::string(line + a, b - a));
    return (int)(neg ? -v : v);
}
inline bool is_space(char c) { return c == ' ' || (unsigned)(c - 9) < 5u; }   // isspace of the C locale, without the table call
inline uint32_t field_pack(const char* line, size_t len, size_t a, size_t b) {   // the stripped field as a packed name (b - a <= 4)
    if (a >= len) return 0;
    b = std::min(b, len);
    while (a < b && is_space(line[a])) a++;
    while (b > a && is_space(line[b - 1])) b--;
    return pack_name(line + a, b - a);
}
inline bool field_blank(const char* line, size_t len, size_t a, size_t b) {
    if (a >= len) return true;
    b = std::min(b, len);
    for (; a < b; a++) if (!is_space(line[a])) return false;
    return true;
}
inline std::string field_strip(const char* line, size_t len, size_t a, size_t b) {
    if (a >= len) return std::string();
    b = std::min(b, len);
    while (a < b && isspace((unsigned char)line[a])) a++;
    while (b > a && isspace((unsigned char)line[b - 1])) b--;
    return std::string(line + a, b - a);
}
// name -> code tables of the codec on the packed names: a small open-addressed hash (the parse threads look up every atom)
struct NameCodes {
    struct Slot { uint32_t key; int code; };
    Slot atoms[256], residues[128];
    static uint32_t h(uint32_t k) { return (k * 0x9E3779B1u) >> 24; }
    static void put(Slot* tab, uint32_t mask, uint32_t k, int code) { uint32_t i = h(k) & mask; while (tab[i].code != -2) i = (i + 1) & mask; tab[i] = {k, code}; }
    static int get(const Slot* tab, uint32_t mask, uint32_t k, int miss) {
        for (uint32_t i = h(k) & mask; tab[i].code != -2; i = (i + 1) & mask) if (tab[i].key == k) return tab[i].code;
        return miss;
    }
    NameCodes() {
        for (Slot& x : atoms) x = {0, -2};
        for (Slot& x : residues) x = {0, -2};
        for (int i = 0; i < 64; i++) { const char* n = fcz_atom_code_name(i); if (n && strlen(n) <= 4 && fcz_atom_code_from_name(n) == i) put(atoms, 255, pack_name(n, strlen(n)), i); }
        for (int i = 0; i < 32; i++) { const char* n = fcz_res_code_name(i); if (n && strlen(n) <= 4 && fcz_res_code_from_name(n) == i) put(residues, 127, pack_name(n, strlen(n)), i); }
    }
    int atom(uint32_t pk) const { return get(atoms, 255, pk, FCZ_ATOM_CODE_OTHER); }       // a long name is no name of the codec
    int residue(uint32_t pk) const { return get(residues, 127, pk, -1); }
};
const NameCodes& name_codes() { static const NameCodes c; return c; }

// ---- the reference's reader for PDB text: gemmi 0.5.1 read_pdb as StructureReader uses it -------------------------------------
// (src/structure_reader.cpp:31-61, lib/gemmi/pdb.hpp:262-365; restated rule by rule in foldcomp_amd/structure.py parse_pdb_gemmi,
// which tests/test_ingest_vs_reference.py checks against the live reference on mutated files, and this function against that):
// records matched on their first four letters case-insensitively, END stops the reading, MODEL / ENDMDL end a chain run, a line
// is at most 120 characters, an ATOM / HETATM line shorter than 54 characters + line end fails the file, numbers = the longest
// valid prefix of their field (0 when there is none), B-factor 20 when the line ends before column 65, atoms grouped by residue
// (number, insertion code, name, segment) inside a run of lines with one chain name, title = HEADER id code else the TITLE texts.
inline bool g_space(unsigned char c) { return c == ' ' || (unsigned)(c - 9) < 5u; }
inline double g_double(const char* p, size_t n) {          // fast_float::from_chars after blanks and one '+'
    const char* e = p + n;
    while (p < e && g_space((unsigned char)*p)) p++;
    if (p < e && *p == '+') p++;
    const char* q = p;
    if (q < e && *q == '-') q++;
    const char* d0 = q;
    while (q < e && *q >= '0' && *q <= '9') q++;
    size_t nd = (size_t)(q - d0);
    if (q < e && *q == '.') { q++; const char* f0 = q; while (q < e && *q >= '0' && *q <= '9') q++; nd += (size_t)(q - f0); }
    if (nd == 0) {                                          // no digits: inf / infinity / nan (any case), else nothing
        const char* w = p; bool neg = false;
        if (w < e && *w == '-') { neg = true; w++; }
        auto is = [&](const char* word) { size_t k = strlen(word); if ((size_t)(e - w) < k) return false; for (size_t i = 0; i < k; i++) if (((unsigned char)w[i] | 0x20) != (unsigned char)word[i]) return false; return true; };
        if (is("inf")) return neg ? -HUGE_VAL : HUGE_VAL;
        if (is("nan")) return neg ? -NAN : NAN;
        return 0.0;
    }
    if (q < e && (*q == 'e' || *q == 'E')) {
        const char* x = q + 1;
        if (x < e && (*x == '+' || *x == '-')) x++;
        if (x < e && *x >= '0' && *x <= '9') { while (x < e && *x >= '0' && *x <= '9') x++; q = x; }
    }
    char tmp[64]; const size_t len = std::min<size_t>((size_t)(q - p), sizeof tmp - 1);
    memcpy(tmp, p, len); tmp[len] = 0;
    return strtod(tmp, nullptr);
}
inline int g_int(const char* p, size_t n) {                 // string_to_int(p, false, n): blanks, sign, digits; wraps like int
    size_t i = 0;
    while (i < n && g_space((unsigned char)p[i])) i++;
    bool neg = false;
    if (i < n && p[i] == '-') { neg = true; i++; } else if (i < n && p[i] == '+') i++;
    uint32_t v = 0;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; i++) v = v * 10u + (uint32_t)(p[i] - '0');
    return (int)(neg ? 0u - v : v);
}
inline long g_base36(const char* p, size_t n) { char z[8] = {0}; memcpy(z, p, std::min<size_t>(n, 7)); return strtol(z, nullptr, 36); }
inline uint32_t g_pack(const char* p, size_t n, AtomTable* t = nullptr) {   // read_string: left trim, stop at the line end, right trim
    size_t a = 0;
    while (a < n && g_space((unsigned char)p[a])) a++;
    size_t b = a;
    while (b < n && p[b] != '\n' && p[b] != '\r' && p[b] != '\0') b++;
    while (b > a && g_space((unsigned char)p[b - 1])) b--;
    (void)t;
    return pack_name(p + a, b - a);
}
inline uint32_t g_id4(const char* s) { return (((uint32_t)(unsigned char)s[0] << 24) | ((uint32_t)(unsigned char)s[1] << 16) | ((uint32_t)(unsigned char)s[2] << 8) | (uint32_t)(unsigned char)s[3]) & ~0x20202020u; }

AtomTable parse_pdb_gemmi(const char* data, size_t size, std::string& title) {
    AtomTable t = table_pool().get();
    t.reserve(size / 78 + 8);
    const NameCodes& nc = name_codes();
    struct Rid { int seq; char icode; uint32_t resn, seg; bool operator==(const Rid& o) const { return seq == o.seq && icode == o.icode && resn == o.resn && seg == o.seg; } };
    struct RidHash { size_t operator()(const Rid& r) const { return (size_t)((uint32_t)r.seq * 0x9E3779B1u) ^ ((size_t)r.resn << 7) ^ ((size_t)r.seg << 17) ^ (size_t)(unsigned char)r.icode; } };
    std::vector<uint64_t> order;                       // (run << 32 | residue ordinal inside the run) of every atom
    std::vector<uint8_t> aniso;                        // the atom has an ANISOU record with u11 != 0
    std::unordered_map<Rid, uint32_t, RidHash> resmap; std::vector<Rid> run_rids; bool map_built = false; long long max_key = 0;
    std::vector<std::string> model_names; std::vector<bool> model_has_chains;
    int model = -1; bool have_chain = false; uint16_t chain_key = 0; uint32_t run = 0, n_res_in_run = 0;
    bool have_resi = false; Rid cur{}; uint32_t cur_ord = 0; long last_atom_of_cur = -1; bool regroup = false;
    std::vector<long> last_atom_of;                    // last atom read of every residue of the run (for ANISOU)
    std::string entry_id; title.clear();
    uint32_t last_res = 0xfffffffeu; int8_t last_rc = -1;
    const char* p = data; const char* end = data + size;
    char line[128];
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl + 1 : end;
        size_t len = std::min<size_t>((size_t)(le - p), 120);
        memcpy(line, p, len); memset(line + len, 0, 8);               // (reads past the end of a line see its terminator)
        p = le;
        if (memchr(line, 0, len)) { len = strnlen(line, len); memset(line + len, 0, sizeof line - len); if (!len) break; }
        const uint32_t id = g_id4(line);
        if (id == g_id4("ATOM") || id == g_id4("HETA")) {
            if (len < 55) throw std::runtime_error("The line is too short to be correct");
            const uint32_t cn = g_pack(line + 20, 2);
            Rid rid;
            rid.icode = (line[26] != '\r' && line[26] != '\n') ? line[26] : '\0';
            if ((unsigned char)line[22] < 'A') {
                rid.seq = INT_MIN;
                for (int i = 0; i < 4; i++) if (!g_space((unsigned char)line[22 + i])) { rid.seq = g_int(line + 22 + i, (size_t)(4 - i)); break; }
            } else rid.seq = (int)(g_base36(line + 22, 4) - 466560 + 10000);
            rid.resn = g_pack(line + 17, 3);
            rid.seg = len > 72 ? g_pack(line + 72, 4) : 0u;
            if (!have_chain || (uint16_t)cn != chain_key) {
                if (model < 0) {
                    const std::string name = std::to_string(model_names.size() + 1);
                    for (const std::string& m : model_names) if (m == name) throw std::runtime_error("ATOM/HETATM between models");
                    model_names.push_back(name); model_has_chains.push_back(false); model = (int)model_names.size() - 1;
                }
                model_has_chains[(size_t)model] = true;
                have_chain = true; chain_key = (uint16_t)cn; run++; n_res_in_run = 0; run_rids.clear(); map_built = false; last_atom_of.clear(); have_resi = false;
            }
            if (!have_resi || !(cur == rid)) {
                // a residue whose (number, insertion code) is beyond every one of the run so far is new: the usual file never
                // needs the map, which is only built (from the run's list) when a residue could be an earlier one coming back
                const long long key = ((long long)rid.seq << 8) | (unsigned char)rid.icode;
                if (run_rids.empty() || key > max_key) { cur_ord = n_res_in_run++; run_rids.push_back(rid); if (map_built) resmap.emplace(rid, cur_ord); last_atom_of.push_back(-1); max_key = key; }
                else {
                    if (!map_built) { resmap.clear(); for (uint32_t k = 0; k < (uint32_t)run_rids.size(); k++) resmap.emplace(run_rids[k], k); map_built = true; }
                    auto it = resmap.find(rid);
                    if (it == resmap.end()) { cur_ord = n_res_in_run++; resmap.emplace(rid, cur_ord); run_rids.push_back(rid); last_atom_of.push_back(-1); if (key > max_key) max_key = key; }
                    else { cur_ord = it->second; if (cur_ord + 1 != n_res_in_run) regroup = true; }
                }
                cur = rid; have_resi = true;
            }
            const uint32_t an = g_pack(line + 12, 4);
            t.atom.push_back(an); t.residue.push_back(rid.resn);
            t.atom_code.push_back((uint8_t)nc.atom(an));
            if (rid.resn != last_res) { last_res = rid.resn; last_rc = (int8_t)nc.residue(rid.resn); }
            t.res_code.push_back(last_rc);
            t.chain.push_back((cn & 0xff) ? (char)(cn & 0xff) : ' '); t.chain_key.push_back((uint16_t)cn);
            t.atom_index.push_back((unsigned char)line[6] < 'A' ? g_int(line + 6, 5) : (int)(g_base36(line + 6, 5) - 16796160 + 100000));
            t.res_index.push_back(rid.seq);
            float fx, fy, fz, fb;
            if (fixed_field<8, 3>(line + 30, fx) && fixed_field<8, 3>(line + 38, fy) && fixed_field<8, 3>(line + 46, fz)) { t.x.push_back(fx); t.y.push_back(fy); t.z.push_back(fz); }
            else { t.x.push_back((float)g_double(line + 30, 8)); t.y.push_back((float)g_double(line + 38, 8)); t.z.push_back((float)g_double(line + 46, 8)); }
            if (len > 64) { if (len >= 67 && fixed_field<6, 2>(line + 60, fb)) t.bfac.push_back(fb); else t.bfac.push_back((float)g_double(line + 60, 6)); }
            else t.bfac.push_back(20.0f);                       // gemmi's default B-factor: the line ends before the field
            if (len > 78) {
                // read_charge (lib/gemmi/pdb.hpp:85-98): a digit in columns 79-80 needs a sign (or nothing) beside it
                char digit = line[78], sign = line[79];
                if (!(digit == ' ' && sign == ' ')) {
                    if (sign >= '0' && sign <= '9') std::swap(digit, sign);
                    if (digit >= '0' && digit <= '9' && sign != '+' && sign != '-' && sign != '\0' && !g_space((unsigned char)sign))
                        throw std::runtime_error("Wrong format for charge");
                }
            }
            order.push_back(((uint64_t)run << 32) | cur_ord); aniso.push_back(0);
            last_atom_of[cur_ord] = (long)t.atom.size() - 1; last_atom_of_cur = (long)t.atom.size() - 1;
        } else if (id == g_id4("ANIS")) {
            if (model < 0 || !have_chain || !have_resi || last_atom_of[cur_ord] < 0) throw std::runtime_error("ANISOU record not directly after ATOM/HETATM.");
            uint8_t& a = aniso[(size_t)last_atom_of[cur_ord]];
            if (a) throw std::runtime_error("Duplicated ANISOU record or not directly after ATOM/HETATM.");
            a = len > 28 && ((float)g_int(line + 28, 7) * 1e-4f) != 0.f;
        } else if (id == g_id4("HEAD")) {
            if (len > 66) { size_t b = 66; while (b > 62 && (line[b - 1] == ' ' || line[b - 1] == '\r' || line[b - 1] == '\n' || line[b - 1] == '\t')) b--; if (b > 62) entry_id.assign(line + 62, b - 62); }
        } else if (id == g_id4("TITL")) {
            if (len > 10) { size_t b = len - 1; while (b > 10 && (line[b - 1] == ' ' || line[b - 1] == '\r' || line[b - 1] == '\n' || line[b - 1] == '\t')) b--; title.append(line + 10, b - 10); }
        } else if (id == g_id4("CRYS")) {
            // UnitCell::set -> calculate_properties: a cell whose gamma is given and whose alpha or beta is exactly zero fails the file
            if (len > 54 && g_double(line + 47, 7) != 0.0 && (g_double(line + 33, 7) == 0.0 || g_double(line + 40, 7) == 0.0))
                throw std::runtime_error("Impossible angle - N*180deg.");
        } else if (id == g_id4("MODE")) {
            if (model >= 0 && have_chain) throw std::runtime_error("MODEL without ENDMDL?");
            const std::string name = std::to_string(g_int(line + 10, 4));
            model = -1;
            for (size_t m = 0; m < model_names.size(); m++) if (model_names[m] == name) model = (int)m;
            if (model < 0) { model_names.push_back(name); model_has_chains.push_back(false); model = (int)model_names.size() - 1; }
            if (model_has_chains[(size_t)model]) throw std::runtime_error("duplicate MODEL number: " + name);
            have_chain = false;
        } else if (id == g_id4("ENDM")) {
            model = -1; have_chain = false;
        } else if ((id & ~0xfu) == (g_id4("END") & ~0xfu) && (id >> 8) == (g_id4("END") >> 8)) {
            break;
        } else if (id == g_id4("data") && line[4] == '_' && model < 0) {
            throw std::runtime_error("Incorrect file format (perhaps it is cif not pdb?)");
        } else if (id == g_id4("{\"da") && ((line[4] & ~0x20) == ('t' & ~0x20)) && ((line[5] & ~0x20) == ('a' & ~0x20)) && ((line[6] & ~0x20) == ('_' & ~0x20)) && model < 0) {
            throw std::runtime_error("Incorrect file format (perhaps it is mmJSON not pdb?)");
        }
    }
    (void)last_atom_of_cur;
    if (regroup) {
        // atoms of a residue whose lines were apart: the reader hands them on residue by residue
        std::vector<size_t> idx(t.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return order[a] < order[b]; });
        AtomTable o = table_pool().get();
        o.reserve(t.size());
        for (size_t i : idx) o.push_from(t, i);
        table_pool().put(std::move(t));
        t = std::move(o);
    }
    if (!entry_id.empty()) title = entry_id;
    return t;
}

AtomTable remove_alternative_position(const AtomTable& t) {
    AtomTable o;
    o.long_names = t.long_names; o.chain_names = t.chain_names;
    o.reserve(t.size());
    bool have = false; uint32_t prev = 0;
    for (size_t i = 0; i < t.size(); i++) {
        if (have && t.atom[i] == prev) continue;
        o.push_from(t, i);
        prev = t.atom[i]; have = true;
    }
    return o;
}

// ---- .gz inputs (zlib) ----
std::string gunzip(const std::string& z) {
    z_stream st{};
    if (inflateInit2(&st, 16 + MAX_WBITS) != Z_OK) throw std::runtime_error("zlib init failed");
    st.next_in = (Bytef*)z.data(); st.avail_in = (uInt)z.size();
    std::string out; char buf[1 << 16];
    int rc;
    do {
        st.next_out = (Bytef*)buf; st.avail_out = sizeof buf;
        rc = inflate(&st, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&st); throw std::runtime_error("not a valid gzip stream"); }
        out.append(buf, sizeof buf - st.avail_out);
    } while (rc != Z_STREAM_END);
    inflateEnd(&st);
    return out;
}

// the first bytes of a gzip member's text (what decides PDB / mmCIF for a database entry whose member is inflated on the device);
// -1: the stream does not start like a gzip member
long gunzip_head(const uint8_t* z, size_t n, char* out, size_t cap) {
    z_stream st{};
    if (inflateInit2(&st, 16 + MAX_WBITS) != Z_OK) return -1;
    st.next_in = (Bytef*)z; st.avail_in = (uInt)n;
    st.next_out = (Bytef*)out; st.avail_out = (uInt)cap;
    const int rc = inflate(&st, Z_NO_FLUSH);
    const long got = (long)(cap - st.avail_out);
    inflateEnd(&st);
    return (rc == Z_OK || rc == Z_STREAM_END || (rc == Z_BUF_ERROR && got > 0)) ? got : -1;
}

// ---- minimal mmCIF reader: the _atom_site loop (what gemmi hands to StructureReader::updateStructure, reference
//      src/structure_reader.cpp:31-61) and _entry.id ----
// ---- mmCIF as the reference's reader takes it: gemmi 0.5.1's grammar (lib/gemmi/cif.hpp:37-148), its table look-ups
//      (cifdoc.hpp Block::find / find_values) and make_structure_from_block's atoms (mmcif.hpp:560-680). Restated rule by rule in
//      foldcomp_amd/structure.py (parse_cif_gemmi), where the rules are checked against the live reference on mutated files ----
using sv = std::string_view;
struct CifItem { int type = 0; sv tag, value; std::vector<sv> tags, values; sv name; std::vector<CifItem> items; };   // 0 pair, 1 loop, 2 frame
struct CifBlock { sv name; bool global = false; std::vector<CifItem> items; };
inline bool cif_ordinary(unsigned char c) {           // char_table(c) == 1
    static const char* ord = "!%&()*+,-./0123456789:<=>?@ABCDEFGHIJKLMNOPQRSTUVWXYZ\\^`abcdefghijklmnopqrstuvwxyz{|}~";
    static bool tab[256], init = false;
    if (!init) { for (const char* p = ord; *p; p++) tab[(unsigned char)*p] = true; init = true; }
    return tab[c];
}
inline bool cif_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
struct CifScanner {
    const char* d; size_t n, i = 0;
    bool ws() {
        const size_t i0 = i;
        while (i < n) {
            if (cif_ws((unsigned char)d[i])) i++;
            else if (d[i] == '#') { const void* q = memchr(d + i, '\n', n - i); i = q ? (size_t)((const char*)q - d) + 1 : n; }
            else break;
        }
        return i > i0;
    }
    bool ws_or_eof() { return ws() || i >= n; }
    int keyword_at(size_t at) const {                  // 1 data_ 2 loop_ 3 global_ 4 save_ 5 stop_
        static const char* kw[] = {"data_", "loop_", "global_", "save_", "stop_"};
        for (int k = 0; k < 5; k++) {
            const size_t l = strlen(kw[k]);
            if (at + l > n) continue;
            bool eq = true;
            for (size_t q = 0; q < l && eq; q++) eq = tolower((unsigned char)d[at + q]) == kw[k][q];
            if (eq) return k + 1;
        }
        return 0;
    }
    size_t nonblank_run(size_t at) const { while (at < n && (unsigned char)d[at] >= 0x21 && (unsigned char)d[at] <= 0x7e) at++; return at; }
    bool tag(sv& out) {
        if (i < n && d[i] == '_') { const size_t j = nonblank_run(i + 1); if (j > i + 1) { out = sv(d + i, j - i); i = j; return true; } }
        return false;
    }
    bool value(sv& out) {                              // throws on an unterminated string / text field
        if (i >= n) return false;
        const char c = d[i];
        size_t j = i;
        while (j < n && cif_ordinary((unsigned char)d[j])) j++;
        if (j > i && j < n && cif_ws((unsigned char)d[j])) { out = sv(d + i, j - i); i = j; return true; }
        if (c == '\'' || c == '"') {
            for (j = i + 1;; j++) {
                if (j >= n || d[j] == '\n') throw std::runtime_error("unterminated string");
                if (d[j] == c && (j + 1 >= n || d[j + 1] == ' ' || d[j + 1] == '\n' || d[j + 1] == '\r' || d[j + 1] == '\t' || d[j + 1] == '#')) { out = sv(d + i, j + 1 - i); i = j + 1; return true; }
            }
        }
        if (c == ';' && (i == 0 || d[i - 1] == '\n')) {
            const sv rest(d + i, n - i);
            const size_t k = rest.find("\n;");
            if (k == sv::npos) throw std::runtime_error("unterminated text field");
            out = sv(d + i, k + 2); i += k + 2; return true;
        }
        if (keyword_at(i) || c == '_' || c == '$' || c == '#') return false;
        j = nonblank_run(i);
        if (j > i) { out = sv(d + i, j - i); i = j; return true; }
        return false;
    }
};
void cif_items(CifScanner& sc, bool in_frame, std::vector<CifItem>& items) {
    for (;;) {
        sv t;
        if (sc.tag(t)) {
            if (!sc.ws()) throw std::runtime_error("parse error");
            CifItem it; it.type = 0; it.tag = t;
            if (!sc.value(it.value) || !sc.ws_or_eof()) throw std::runtime_error(std::string(t) + " has no value");
            items.push_back(std::move(it));
            continue;
        }
        const int k = sc.keyword_at(sc.i);
        if (k == 2) {
            sc.i += 5;
            if (!sc.ws()) throw std::runtime_error("parse error");
            CifItem it; it.type = 1;
            while (sc.tag(t)) { if (!sc.ws()) throw std::runtime_error("parse error"); it.tags.push_back(t); }
            if (it.tags.empty()) throw std::runtime_error("parse error");
            for (;;) {
                const size_t at = sc.i; sv v;
                if (!sc.value(v)) break;
                if (!sc.ws_or_eof()) { sc.i = at; break; }
                it.values.push_back(v);
            }
            if (it.values.empty() && !(sc.i >= sc.n || sc.keyword_at(sc.i))) throw std::runtime_error("parse error");
            if (sc.keyword_at(sc.i) == 5) { const size_t at = sc.i; sc.i += 5; if (!sc.ws_or_eof()) sc.i = at; }
            if (it.values.size() % it.tags.size() != 0) throw std::runtime_error("Wrong number of values in the loop");
            items.push_back(std::move(it));
            continue;
        }
        if (k == 4 && !in_frame) {
            const size_t j = sc.nonblank_run(sc.i + 5);
            if (j == sc.i + 5) return;
            CifItem it; it.type = 2; it.name = sv(sc.d + sc.i + 5, j - sc.i - 5); sc.i = j;
            if (!sc.ws()) throw std::runtime_error("parse error");
            cif_items(sc, true, it.items);
            if (sc.keyword_at(sc.i) != 4) throw std::runtime_error("parse error");
            sc.i += 5;
            if (!sc.ws_or_eof()) throw std::runtime_error("parse error");
            items.push_back(std::move(it));
            continue;
        }
        return;
    }
}
std::string cif_lower(sv s) { std::string o(s); for (char& c : o) c = (char)tolower((unsigned char)c); return o; }
std::vector<CifBlock> cif_document(const char* data, size_t size) {
    CifScanner sc{data, size};
    sc.ws();
    std::vector<CifBlock> blocks;
    if (sc.i >= sc.n) return blocks;
    for (;;) {
        const int k = sc.keyword_at(sc.i);
        CifBlock b;
        if (k == 1) { const size_t j = sc.nonblank_run(sc.i + 5); b.name = j > sc.i + 5 ? sv(sc.d + sc.i + 5, j - sc.i - 5) : sv("#"); sc.i = j; }
        else if (k == 3) { b.global = true; sc.i += 7; }
        else break;
        if (!sc.ws_or_eof()) throw std::runtime_error("parse error");
        cif_items(sc, false, b.items);
        blocks.push_back(std::move(b));
    }
    if (blocks.empty()) throw std::runtime_error("expected block header (data_)");
    if (sc.i < sc.n) throw std::runtime_error("parse error");
    std::unordered_set<std::string> seen;
    for (const CifBlock& b : blocks) { if (!seen.insert(cif_lower(b.name)).second && !b.name.empty()) throw std::runtime_error("duplicate block name"); }
    for (const CifBlock& b : blocks) {
        std::unordered_set<std::string> tags, frames;
        for (const CifItem& it : b.items) {
            if (it.type == 0) { if (!tags.insert(cif_lower(it.tag)).second) throw std::runtime_error("duplicate tag " + std::string(it.tag)); }
            else if (it.type == 1) { for (sv t : it.tags) if (!tags.insert(cif_lower(t)).second) throw std::runtime_error("duplicate tag " + std::string(t)); }
            else if (!frames.insert(cif_lower(it.name)).second) throw std::runtime_error("duplicate save_" + std::string(it.name));
        }
    }
    return blocks;
}
inline bool cif_null(sv v) { return v.size() == 1 && (v[0] == '?' || v[0] == '.'); }
sv cif_string(sv v) {                                  // cif::as_string
    if (v.empty() || cif_null(v)) return sv();
    if (v[0] == '"' || v[0] == '\'') return v.substr(1, v.size() - 2);
    if (v[0] == ';' && v.size() > 2 && v[v.size() - 2] == '\n') return v.substr(1, v.size() - (v[v.size() - 3] == '\r' ? 4 : 3));
    return v;
}
double cif_number(sv v, double dflt) {                 // cif::as_number: the whole value is a number (+ an uncertainty in brackets)
    if (!v.empty() && v[0] == '+') v.remove_prefix(1);
    // the usual value, "-12.345": digits and a point, read exactly (fast_decimal = strtod's result for up to 15 digits)
    if (!v.empty() && v[0] != '+' && v.back() != ' ' && v.back() != '\t' && v.back() != '\r' && v[0] != ' ' && v[0] != '\t') {
        double d;
        if (fast_decimal(v.data(), v.data() + v.size(), d)) return d;
    }
    size_t i = 0; const size_t n = v.size();
    if (i < n && v[i] == '-') i++;
    size_t d0 = i; while (i < n && isdigit((unsigned char)v[i])) i++;
    size_t nd = i - d0;
    if (i < n && v[i] == '.') { i++; const size_t f0 = i; while (i < n && isdigit((unsigned char)v[i])) i++; nd += i - f0; }
    if (nd == 0) return dflt;
    if (i < n && (v[i] == 'e' || v[i] == 'E')) {
        size_t j = i + 1; if (j < n && (v[j] == '+' || v[j] == '-')) j++;
        const size_t e0 = j; while (j < n && isdigit((unsigned char)v[j])) j++;
        if (j > e0) i = j;
    }
    const size_t num_end = i;
    if (i < n && v[i] == '(') { size_t j = i + 1; while (j < n && isdigit((unsigned char)v[j])) j++; if (j < n && v[j] == ')') i = j + 1; }
    if (i != n) return dflt;
    return strtod(std::string(v.substr(0, num_end)).c_str(), nullptr);
}
int cif_int_checked(sv v) {                            // string_to_int(str, true): what it throws the reference does not survive
    size_t i = 0; const size_t n = v.size();
    auto sp = [](char c) { return c == ' ' || (c >= 9 && c <= 13); };
    while (i < n && sp(v[i])) i++;
    bool neg = false;
    if (i < n && (v[i] == '-' || v[i] == '+')) { neg = v[i] == '-'; i++; }
    const size_t d0 = i; int64_t acc = 0;
    while (i < n && isdigit((unsigned char)v[i])) { acc = (acc * 10 + (v[i] - '0')) & 0xffffffffll; i++; }
    const bool has = i > d0;
    while (i < n && sp(v[i])) i++;
    if (!has || i != n) throw std::runtime_error("not an integer: " + std::string(v));
    const uint32_t u = (uint32_t)acc;
    return neg ? (int)(0u - u) : (int)u;
}
// Block::find_values: the first loop that has the tag (any case) or the first pair that IS the tag (this case)
const CifItem* cif_find(const std::vector<CifItem>& items, sv tag, size_t& col) {
    const std::string low = cif_lower(tag);
    for (const CifItem& it : items) {
        if (it.type == 1) { for (size_t c = 0; c < it.tags.size(); c++) if (cif_lower(it.tags[c]) == low) { col = c; return &it; } }
        else if (it.type == 0 && it.tag == tag) { col = 0; return &it; }
    }
    return nullptr;
}

AtomTable parse_cif_gemmi(const char* data, size_t size, std::string& title) {
    std::vector<CifBlock> blocks = cif_document(data, size);
    if (blocks.empty()) throw std::runtime_error("empty file");
    size_t c0 = 0;
    auto has_tag = [&](const CifBlock& b, sv tag) { size_t c; return cif_find(b.items, tag, c) != nullptr; };
    // monomer-library and CCD files take another route in gemmi (chemcomp_xyz.hpp:106-120): no protein chain comes out of those
    if ((blocks.size() == 2 && blocks[0].name == "comp_list" && !blocks[0].global) || (blocks.size() == 3 && blocks[0].global && blocks[1].name == "comp_list") ||
        (blocks.size() == 1 && !has_tag(blocks[0], "_atom_site.id") && has_tag(blocks[0], "_chem_comp_atom.atom_id")))
        throw std::runtime_error("a chemical-component dictionary, not a structure");
    for (size_t b = 1; b < blocks.size(); b++) if (has_tag(blocks[b], "_atom_site.id")) throw std::runtime_error("2+ blocks are ok if only the first one has coordinates");
    const std::vector<CifItem>& items = blocks[0].items;
    auto pair_value = [&](sv tag, sv& out) { for (const CifItem& it : items) if (it.type == 0 && it.tag == tag) { out = it.value; return true; } return false; };
    // the cell: six values as ONE row; a zero alpha or beta beside a gamma fails (UnitCell::set -> calculate_properties)
    {
        static const char* ct[6] = {"_cell.length_a", "_cell.length_b", "_cell.length_c", "_cell.angle_alpha", "_cell.angle_beta", "_cell.angle_gamma"};
        const CifItem* it0 = cif_find(items, ct[0], c0);
        sv cell[6]; bool have = false;
        if (it0 && it0->type == 1) {
            bool all = true; size_t cc[6];
            for (int k = 0; k < 6 && all; k++) { all = false; for (size_t c = 0; c < it0->tags.size(); c++) if (cif_lower(it0->tags[c]) == cif_lower(ct[k])) { cc[k] = c; all = true; break; } }
            if (all) {
                const size_t rows = it0->values.size() / it0->tags.size();
                if (rows != 1) throw std::runtime_error("Expected one value, found " + std::to_string(rows));
                for (int k = 0; k < 6; k++) cell[k] = it0->values[cc[k]];
                have = true;
            }
        } else {
            have = true;
            for (int k = 0; k < 6 && have; k++) have = pair_value(ct[k], cell[k]);
        }
        if (have && !cif_null(cell[0]) && !cif_null(cell[1]) && !cif_null(cell[2])) {
            const double al = cif_number(cell[3], NAN), be = cif_number(cell[4], NAN), ga = cif_number(cell[5], NAN);
            if (ga != 0.0 && (al == 0.0 || be == 0.0)) throw std::runtime_error("Impossible angle - N*180deg.");
        }
    }
    auto info = [&](sv tag) {
        std::string out; size_t c;
        const CifItem* it = cif_find(items, tag, c);
        if (!it) return out;
        bool first = true;
        auto add = [&](sv v) { if (cif_null(v)) return; if (!first) out += "; "; out += std::string(cif_string(v)); first = false; };
        if (it->type == 0) add(it->value); else for (size_t r = c; r < it->values.size(); r += it->tags.size()) add(it->values[r]);
        return out;
    };
    title = info("_entry.id");
    if (title.empty()) title = info("_struct.title");
    static const char* want[23] = {"id", "?group_PDB", "type_symbol", "?label_atom_id", "label_alt_id", "?label_comp_id", "label_asym_id", "?label_entity_id",
        "?label_seq_id", "?pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "?pdbx_formal_charge", "auth_seq_id", "?auth_comp_id",
        "?auth_asym_id", "?auth_atom_id", "?pdbx_PDB_model_num", "?calc_flag", "?pdbx_tls_group_id"};
    enum { kId, kGroup, kSymbol, kLabelAtom, kAlt, kLabelComp, kLabelAsym, kLabelEntity, kLabelSeq, kIns, kX, kY, kZ, kOcc, kB, kCharge, kAuthSeq, kAuthComp, kAuthAsym,
           kAuthAtom, kModel, kCalc, kTls };
    const CifItem* loop = cif_find(items, "_atom_site.id", c0);
    int pos[23]; bool ok = false; size_t n_rows = 0; size_t width = 0;
    std::vector<sv> one;
    if (loop && loop->type == 1) {
        ok = true;
        for (int k = 0; k < 23 && ok; k++) {
            const bool opt = want[k][0] == '?';
            const std::string full = cif_lower(std::string("_atom_site.") + (want[k] + (opt ? 1 : 0)));
            pos[k] = -1;
            for (size_t c = 0; c < loop->tags.size(); c++) if (cif_lower(loop->tags[c]) == full) { pos[k] = (int)c; break; }
            if (pos[k] < 0 && !opt) ok = false;
        }
        if (ok) { width = loop->tags.size(); n_rows = loop->values.size() / width; }
    } else {
        ok = true;
        for (int k = 0; k < 23 && ok; k++) {
            const bool opt = want[k][0] == '?';
            const std::string full = std::string("_atom_site.") + (want[k] + (opt ? 1 : 0));
            sv v;
            if (pair_value(full, v)) { pos[k] = (int)one.size(); one.push_back(v); }
            else if (opt) pos[k] = -1;
            else ok = false;
        }
        if (ok) { width = one.size(); n_rows = 1; }
    }
    AtomTable t = table_pool().get();
    if (!ok || n_rows == 0) return t;
    const std::vector<sv>& vals = (loop && loop->type == 1) ? loop->values : one;
    const int kAsym = pos[kAuthAsym] >= 0 ? kAuthAsym : kLabelAsym, kComp = pos[kAuthComp] >= 0 ? kAuthComp : kLabelComp, kAtom = pos[kAuthAtom] >= 0 ? kAuthAtom : kLabelAtom;
    if (pos[kComp] < 0) throw std::runtime_error("Neither _atom_site.label_comp_id nor auth_comp_id found");
    if (pos[kAtom] < 0) throw std::runtime_error("Neither _atom_site.label_atom_id nor auth_atom_id found");
    // models / chains / residues in the order they are made; every atom gets (model, chain, residue) ordinals and is sorted by them
    struct Res { int num; char icode; sv name; };
    struct Ch { sv name; std::vector<Res> res; uint32_t ord; };
    struct Mo { std::string name; std::vector<Ch> chains; };
    std::vector<Mo> models;
    auto model_named = [&](sv nm) -> size_t { for (size_t m = 0; m < models.size(); m++) if (models[m].name == nm) return m; models.push_back(Mo{std::string(nm), {}}); return models.size() - 1; };
    size_t model = model_named(pos[kModel] >= 0 ? cif_string(vals[(size_t)pos[kModel]]) : sv("1"));
    long chain = -1, resi = -1;
    std::vector<uint64_t> order; order.reserve(n_rows);
    std::vector<uint32_t> chain_ord_of_model;             // running count of chains, for the sort key
    uint32_t n_chain_total = 0;
    bool sorted = true; uint64_t last_key = 0;
    t.reserve(n_rows);
    const NameCodes& nc = name_codes();
    for (size_t r = 0; r < n_rows; r++) {
        const sv* row = vals.data() + r * width;
        if (pos[kModel] >= 0 && row[pos[kModel]] != sv(models[model].name)) { model = model_named(cif_string(row[pos[kModel]])); chain = -1; }
        const sv asym = cif_string(row[pos[kAsym]]);
        if (chain < 0 || asym != models[model].chains[(size_t)chain].name) {
            models[model].chains.push_back(Ch{asym, {}, n_chain_total++}); chain = (long)models[model].chains.size() - 1; resi = -1;
        }
        Ch& ch = models[model].chains[(size_t)chain];
        const sv seqs = cif_string(row[pos[kAuthSeq]]);
        char icode = ' ';
        if (pos[kIns] >= 0) {
            const sv v = row[pos[kIns]];
            if (cif_null(v)) icode = ' ';
            else if (v.size() < 2) icode = v[0];
            else { const sv s2 = cif_string(v); if (s2.size() >= 2) throw std::runtime_error("Not a single character"); icode = s2.empty() ? '\0' : s2[0]; }
        }
        int num = -999;                                    // SeqId::OptionalNum::None
        if (!seqs.empty()) {
            if ((unsigned char)seqs.back() >= 'A') {
                if (icode == ' ') icode = seqs.back();
                else if (icode != seqs.back()) throw std::runtime_error("Inconsistent insertion code in " + std::string(seqs));
                num = cif_int_checked(seqs.substr(0, seqs.size() - 1));
            } else num = cif_int_checked(seqs);
        }
        const sv comp = cif_string(row[pos[kComp]]);
        auto matches = [&](const Res& q) { return q.num == num && (q.icode | 0x20) == (icode | 0x20) && q.name == comp; };
        if (resi < 0 || !matches(ch.res[(size_t)resi])) {
            resi = -1;
            for (size_t q = 0; q < ch.res.size(); q++) if (matches(ch.res[q])) { resi = (long)q; break; }
            const bool fresh = resi < 0;
            if (fresh) { ch.res.push_back(Res{num, icode, comp}); resi = (long)ch.res.size() - 1; }
            // (label_seq_id is read when the residue gets its first atom: a residue found again has atoms already)
            if (fresh && pos[kLabelSeq] >= 0 && !cif_null(row[pos[kLabelSeq]])) (void)cif_int_checked(row[pos[kLabelSeq]]);
        }
        const sv alt = row[pos[kAlt]];
        if (!cif_null(alt) && alt.size() >= 2 && cif_string(alt).size() >= 2) throw std::runtime_error("Not a single character");
        if (pos[kCharge] >= 0 && !cif_null(row[pos[kCharge]])) (void)cif_int_checked(row[pos[kCharge]]);
        int serial = 0;
        {
            const sv v = row[pos[kId]]; size_t i = 0; const size_t n = v.size();
            while (i < n && (v[i] == ' ' || (v[i] >= 9 && v[i] <= 13))) i++;
            bool neg = false; if (i < n && (v[i] == '-' || v[i] == '+')) { neg = v[i] == '-'; i++; }
            uint32_t u = 0; while (i < n && isdigit((unsigned char)v[i])) { u = u * 10u + (uint32_t)(v[i] - '0'); i++; }
            serial = neg ? (int)(0u - u) : (int)u;
        }
        const sv an = cif_string(row[pos[kAtom]]);
        const uint32_t a_id = t.intern(an.data(), an.size()), r_id = t.intern(comp.data(), comp.size());
        t.atom.push_back(a_id); t.residue.push_back(r_id);
        t.atom_code.push_back((uint8_t)nc.atom(a_id)); t.res_code.push_back((int8_t)nc.residue(r_id));
        t.chain.push_back(asym.empty() ? ' ' : asym[0]); t.chain_key.push_back(t.intern_chain(std::string(asym)));
        t.atom_index.push_back(serial); t.res_index.push_back(num);
        t.x.push_back((float)cif_number(row[pos[kX]], NAN)); t.y.push_back((float)cif_number(row[pos[kY]], NAN)); t.z.push_back((float)cif_number(row[pos[kZ]], NAN));
        t.bfac.push_back((float)cif_number(row[pos[kB]], 50.0));
        const uint64_t key = ((uint64_t)model << 48) | ((uint64_t)ch.ord << 24) | (uint64_t)resi;
        if (!order.empty() && key < last_key) sorted = false;
        if (order.empty() || key >= last_key) last_key = key;
        order.push_back(key);
    }
    if (!sorted) {
        std::vector<size_t> idx(t.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return order[a] < order[b]; });
        AtomTable o = table_pool().get();
        o.long_names = t.long_names; o.chain_names = t.chain_names;
        o.reserve(t.size());
        for (size_t i : idx) o.push_from(t, i);
        table_pool().put(std::move(t));
        t = std::move(o);
    }
    return t;
}

// gemmi::coor_format_from_content (lib/gemmi/mmread.hpp:31-47): what StructureReader::loadFromBuffer -- the reader of every
// `compress` input, src/main.cpp:457 -- goes by; the file's name only says whether it is gzipped. 0 unknown, 1 pdb, 2 mmcif, 3 mmjson
int coor_format_from_content(const char* d, size_t size) {
    size_t i = 0; const long end = (long)size - 8;
    while ((long)i < end) {
        const unsigned char c = (unsigned char)d[i];
        if (c == ' ' || (c >= 9 && c <= 13)) i++;
        else if (c == '#') { while ((long)i < end && d[i] != '\n') i++; }
        else if (c == '{') return 3;
        else if ((d[i] & ~0x20) == 'D' && (d[i + 1] & ~0x20) == 'A' && (d[i + 2] & ~0x20) == 'T' && (d[i + 3] & ~0x20) == 'A' && d[i + 4] == '_') return 2;
        else return 1;
    }
    return 0;
}
// the same decision from the first `have` bytes of a text of `full` bytes; -1: the prefix does not reach the deciding characters
int coor_format_from_prefix(const char* d, size_t have, size_t full) {
    size_t i = 0; const long end = (long)full - 8;
    while ((long)i < end) {
        if (i + 5 > have) return have >= full ? 0 : -1;
        const unsigned char c = (unsigned char)d[i];
        if (c == ' ' || (c >= 9 && c <= 13)) i++;
        else if (c == '#') { while ((long)i < end && i < have && d[i] != '\n') i++; if (i >= have && have < full) return -1; }
        else if (c == '{') return 3;
        else if ((d[i] & ~0x20) == 'D' && (d[i + 1] & ~0x20) == 'A' && (d[i + 2] & ~0x20) == 'T' && (d[i + 3] & ~0x20) == 'A' && d[i + 4] == '_') return 2;
        else return 1;
    }
    return 0;
}
AtomTable parse_pdb_gemmi(const char* data, size_t size, std::string& title);
AtomTable parse_structure_gemmi(const char* data, size_t size, std::string& title) {
    const int fmt = coor_format_from_content(data, size);
    if (fmt == 1) return parse_pdb_gemmi(data, size, title);
    if (fmt == 2) return parse_cif_gemmi(data, size, title);
    throw std::runtime_error(fmt == 0 ? "wrong format of coordinate file" : "mmJSON input is not supported");
}

struct Range { size_t a, b; };

std::vector<Range> identify_chains(const AtomTable& t) {
    std::vector<Range> out;
    const size_t n = t.size();
    size_t start = 0, i = 1;
    while (i < n) {
        if (!t.same_chain(i, i - 1)) {
            if (t.atom[i] == PK_N) { out.push_back({start, i}); start = i; }
            else {
                size_t j = i;
                while (j < n && t.atom[j] != PK_N) j++;
                if (j == n) break;
                out.push_back({start, i});
                start = j; i = start;
            }
        }
        i++;
    }
    out.push_back({start, n});
    return out;
}

std::vector<Range> identify_discontinuous(const AtomTable& t, Range r) {
    std::vector<size_t> n_idx;
    for (size_t i = r.a; i < r.b; i++) if (t.atom[i] == PK_N) n_idx.push_back(i);
    std::vector<Range> out;
    if (n_idx.empty()) return out;
    size_t start = n_idx[0];
    for (size_t q = 0; q + 1 < n_idx.size(); q++)
        if (t.res_index[n_idx[q + 1]] - t.res_index[n_idx[q]] > 1) { out.push_back({start, n_idx[q + 1]}); start = n_idx[q + 1]; }
    out.push_back({start, r.b});
    return out;
}

// residue boundaries as atom offsets (a new residue starts where residue_index changes; the last atom always joins the
// open residue)
std::vector<uint32_t> split_residues(const AtomTable& t) {
    std::vector<uint32_t> ro{0};
    const size_t n = t.size();
    for (size_t i = 1; i < n; i++) if (t.res_index[i] != t.res_index[i - 1] && i != n - 1) ro.push_back((uint32_t)i);
    ro.push_back((uint32_t)n);
    return ro;
}

// page-locked host memory (fcz_pinned_alloc = hipHostMalloc): what a worker hands to fcz_compress_batch is copied by DMA on the
// ctx stream, so the transfers of one worker overlap the kernels of the other worker on the same GPU. Falls back to malloc when
// no device is present (the CPU-only subcommands and tests).
// ONE switch for every element type (a function-local static inside the template would be one flag per instantiation).
inline std::atomic<bool>& pinned_enabled() { static std::atomic<bool> v{false}; return v; }
inline std::atomic<uint64_t>& pinned_blocks() { static std::atomic<uint64_t> v{0}; return v; }   // page-locked blocks handed out (statistics, tests)
template <class T> struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) {
        const bool pin = pinned_enabled().load();
        void* p = pin ? fcz_pinned_alloc(n * sizeof(T) + 16) : malloc(n * sizeof(T) + 16);
        if (!p) throw std::bad_alloc();
        *(uint64_t*)p = pin ? 1 : 0;     // remember where the block came from
        if (pin) pinned_blocks()++;
        return (T*)((char*)p + 16);      // 16: the payload keeps the alignment of the block
    }
    void deallocate(T* q, size_t) { void* p = (char*)q - 16; if (*(uint64_t*)p) fcz_pinned_free(p); else free(p); }
    template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T> using pvec = std::vector<T, PinnedAlloc<T>>;

// ---- SoA batch = fcz_chain_batch ----
struct Batch {
    std::vector<uint32_t> res_off{0}, title_off{0};
    pvec<uint32_t> atom_off;
    pvec<float> x, y, z, bfac_ca;
    pvec<uint8_t> atom_code, res_code;
    std::vector<int32_t> first_res, first_atom;
    std::string chain_id, titles;
    size_t n_chains() const { return res_off.size() - 1; }
    void clear() {   // keeps the capacity: a worker reuses its (pinned) buffers for every batch
        res_off.assign(1, 0); title_off.assign(1, 0); atom_off.clear(); x.clear(); y.clear(); z.clear(); bfac_ca.clear();
        atom_code.clear(); res_code.clear(); first_res.clear(); first_atom.clear(); chain_id.clear(); titles.clear();
    }

    // what add() derives from a fragment before it touches the batch: residue boundaries, residue codes, CA B-factors. Pure (no
    // batch state), so the parse threads run it side by side; throws std::runtime_error with what the reference would abort on
    struct Prepared { std::vector<uint32_t> ro; std::vector<uint8_t> ac, rc; std::vector<float> bf; };
    static Prepared prepare(const AtomTable& t, int anchor_threshold = 25) {
        if (t.size() == 0) throw std::runtime_error("empty chain");
        Prepared p;
        p.ro = split_residues(t);
        const std::vector<uint32_t>& ro = p.ro;
        const size_t nres = ro.size() - 1;
        // the FCZ header holds nResidue in 16 bits and nAnchor in 8 (src/foldcomp.h:120-125): the reference wraps silently
        // and writes a record nobody can read; here the chain is refused
        if (nres > 65535 || (anchor_threshold > 0 && nres / (size_t)anchor_threshold + 2 > 255))
            throw std::runtime_error("chain of " + std::to_string(nres) + " residues does not fit the FCZ header (65535 residues, 255 anchors)");
        p.ac.resize(t.size()); p.rc.resize(nres); p.bf.assign(nres, 0.0f);
        const bool coded = t.atom_code.size() == t.size();
        for (size_t i = 0; i < t.size(); i++) p.ac[i] = coded ? t.atom_code[i] : (uint8_t)fcz_atom_code_from_name(t.name(t.atom[i]).c_str());
        for (size_t r = 0; r < nres; r++) {
            const int code = coded ? (int)t.res_code[ro[r]] : fcz_res_code_from_name(t.name(t.residue[ro[r]]).c_str());
            if (code < 0) throw std::runtime_error("residue name '" + t.name(t.residue[ro[r]]) + "' is not supported by the codec");
            p.rc[r] = (uint8_t)code;
            long pos[3] = {-1, -1, -1}; int cnt[3] = {0, 0, 0};
            for (uint32_t i = ro[r]; i < ro[r + 1]; i++) if (p.ac[i] < 3) { cnt[p.ac[i]]++; if (pos[p.ac[i]] < 0) pos[p.ac[i]] = i; }
            if (pos[0] < 0 || pos[1] < 0 || pos[2] < 0 || !(pos[0] < pos[1] && pos[1] < pos[2]))
                throw std::runtime_error("residue without N, CA, C backbone atoms in order");
            // the reference works on the flat list of every N / CA / C atom (filterBackbone; nResidue = their number / 3), this
            // codec on the first of each per residue: they agree when a residue has one of each. A second one would shift every
            // later residue of the reference's record: refused, not compressed differently
            if (cnt[0] != 1 || cnt[1] != 1 || cnt[2] != 1) throw std::runtime_error("residue with a second N, CA or C atom");
            p.bf[r] = t.bfac[pos[1]];
        }
        // header.lastResidue is the residue name of the chain's last ATOM (src/foldcomp.cpp:469), the residue codes those of each
        // residue's first atom: one value serves both only when they agree
        if (t.residue[t.size() - 1] != t.residue[ro[nres - 1]])
            throw std::runtime_error("the chain's last atom carries another residue name than its residue");
        return p;
    }
    // appends one prepared fragment (copies only)
    void append(const AtomTable& t, const Prepared& p, const std::string& title) {
        const size_t nres = p.ro.size() - 1, abase = x.size();
        for (size_t r = 0; r < nres; r++) atom_off.push_back((uint32_t)(abase + p.ro[r]));
        x.insert(x.end(), t.x.begin(), t.x.end()); y.insert(y.end(), t.y.begin(), t.y.end()); z.insert(z.end(), t.z.begin(), t.z.end());
        atom_code.insert(atom_code.end(), p.ac.begin(), p.ac.end());
        res_code.insert(res_code.end(), p.rc.begin(), p.rc.end());
        bfac_ca.insert(bfac_ca.end(), p.bf.begin(), p.bf.end());
        res_off.push_back(res_off.back() + (uint32_t)nres);
        first_res.push_back(t.res_index[0]); first_atom.push_back(t.atom_index[0]);
        chain_id.push_back(t.chain[0]);
        titles += title; title_off.push_back((uint32_t)titles.size());
    }
    // appends one fragment; throws before it changes the batch
    void add(const AtomTable& t, const std::string& title, int anchor_threshold = 25) { append(t, prepare(t, anchor_threshold), title); }
    fcz_chain_batch view(int anchor_threshold) {
        if (atom_off.size() == res_code.size()) atom_off.push_back((uint32_t)x.size());
        fcz_chain_batch b{};
        b.n_chains = (uint32_t)n_chains(); b.n_residues = (uint32_t)res_code.size(); b.n_atoms = (uint32_t)x.size();
        b.anchor_threshold = anchor_threshold;
        b.res_off = res_off.data(); b.atom_off = atom_off.data(); b.x = x.data(); b.y = y.data(); b.z = z.data();
        b.atom_code = atom_code.data(); b.res_code = res_code.data(); b.bfac_ca = bfac_ca.data();
        b.first_res_index = first_res.data(); b.first_atom_index = first_atom.data(); b.chain_id = chain_id.data();
        b.titles = titles.data(); b.title_off = title_off.data();
        return b;
    }
};

// ---- files ----
std::string read_file(const std::string& p) {   // POSIX read: iostream construction takes a process-wide locale reference per file
    const int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + p);
    struct stat st;
    std::string out;
    if (fstat(fd, &st) == 0 && st.st_size > 0) out.resize((size_t)st.st_size + 1);   // + 1: the read that returns 0 needs room
    size_t got = 0;
    for (;;) {
        if (got == out.size()) out.resize(out.size() + (1 << 16));
        const ssize_t n = read(fd, &out[got], out.size() - got);
        if (n < 0) { close(fd); throw std::runtime_error("cannot read " + p); }
        if (n == 0) break;
        got += (size_t)n;
    }
    close(fd);
    out.resize(got);
    return out;
}
// the same into a buffer the calling thread keeps (the parse threads read ~240 KB per file: a fresh allocation of that size is
// an mmap + page faults + munmap per file, and the address-space lock those take is what stopped the parse threads from scaling)
struct FileImage { std::unique_ptr<char[]> p; size_t cap = 0, n = 0; const char* data() const { return p.get(); } };
std::atomic<uint64_t> g_bytes_read{0};     // input bytes the parse threads have read (statistics of the compress pipeline)
void read_file_into(const std::string& path, FileImage& im) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    size_t want = (fstat(fd, &st) == 0 && st.st_size > 0) ? (size_t)st.st_size + 1 : (size_t)(1 << 16);
    auto grow = [&](size_t need) {
        if (need <= im.cap) return;
        size_t c = std::max<size_t>(need, im.cap * 2);
        std::unique_ptr<char[]> q(new char[c]);
        if (im.n) memcpy(q.get(), im.p.get(), im.n);
        im.p = std::move(q); im.cap = c;
    };
    im.n = 0;
    grow(want);
    for (;;) {
        if (im.n == im.cap) grow(im.cap + (1 << 16));
        const ssize_t k = read(fd, im.p.get() + im.n, im.cap - im.n);
        if (k < 0) { close(fd); throw std::runtime_error("cannot read " + path); }
        if (k == 0) break;
        im.n += (size_t)k;
    }
    close(fd);
    g_bytes_read += im.n;
}
bool is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
std::string base_name(const std::string& p) { const size_t i = p.find_last_of('/'); return i == std::string::npos ? p : p.substr(i + 1); }
void list_files(const std::string& dir, bool recursive, std::vector<std::string>& out) {
    std::vector<std::string> files, dirs;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n == "." || n == "..") continue;
            const std::string p = dir + "/" + n;
            // the entry type comes with the directory entry on every file system that knows it: no stat per file
            const bool dir_entry = e->d_type == DT_DIR || ((e->d_type == DT_UNKNOWN || e->d_type == DT_LNK) && is_dir(p));
            (dir_entry ? dirs : files).push_back(p);
        }
        closedir(d);
    }
    std::sort(files.begin(), files.end()); std::sort(dirs.begin(), dirs.end());
    out.insert(out.end(), files.begin(), files.end());
    if (recursive) for (const std::string& s : dirs) list_files(s, true, out);
}
// getFileParts: split at the last '.', a trailing .gz stays with the extension
// getFileParts (reference src/utility.cpp:118-126): split at the LAST dot ("test.cif.gz" -> "test.cif", "gz")
void file_parts(const std::string& base, std::string& stem, std::string& ext) {
    stem = base; ext.clear();
    const size_t i = base.rfind('.');
    if (i != std::string::npos) { stem = base.substr(0, i); ext = base.substr(i + 1); }
}
// isCompressible (src/utility.cpp:129-140): pdb, cif, and either of them gzipped (the .gz case looks one extension further in)
bool is_compressible(const std::string& stem, const std::string& ext) {
    if (ext == "pdb" || ext == "cif") return true;
    if (ext != "gz") return false;
    std::string s2, e2;
    file_parts(stem, s2, e2);
    return e2 == "pdb" || e2 == "cif";
}

bool write_out(const std::string& path, const char* data, size_t n, bool overwrite) {
    if (exists(path) && !overwrite) { fprintf(stderr, "[Error] Output file already exists: %s\n", base_name(path).c_str()); return false; }
    std::ofstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "[Error] cannot write %s\n", path.c_str()); return false; }
    f.write(data, (std::streamsize)n);
    return true;
}
void make_dir(const std::string& p) { if (!exists(p)) mkdir(p.c_str(), 0777); }


// the words of an index / lookup line: separated by blanks and tabs (src/database_reader.cpp:283-361)
inline int line_words(const char* s, size_t n, const char* w[4], size_t wl[4]) {
    int k = 0; size_t i = 0;
    while (i < n) {
        while (i < n && (s[i] == ' ' || s[i] == '\t')) i++;
        const size_t st = i;
        while (i < n && s[i] != ' ' && s[i] != '\t') i++;
        if (i > st) { if (k < 4) { w[k] = s + st; wl[k] = i - st; } k++; }
    }
    return k;
}
inline bool all_digits_u64(const char* s, size_t n, uint64_t& v) {   // a plain decimal number (what strtoull reads the same way)
    if (n == 0 || n > 19) return false;
    v = 0;
    for (size_t i = 0; i < n; i++) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (uint64_t)(s[i] - '0'); }
    return true;
}

// ---- Foldcomp / MMseqs2-style database container (reference src/database_reader.cpp, src/database_writer.cpp):
//      `<db>` concatenated entries, `<db>.index` lines "key\toffset\tlength", `<db>.lookup` lines "key\tname\t0",
//      `<db>.dbtype` = int32 12 ----
struct DbReader {
    struct Row { long long key, off, len; };
    std::vector<Row> rows;                 // sorted by key (stable), as the reference reader does
    struct Look { long long first; uint64_t pos; uint32_t len; };   // a lookup line: its key, where its name lies in `lraw`
    std::string lraw;                      // the .lookup file as it was read (214 M names are not 214 M strings)
    std::vector<Look> lookup;              // sorted by key
    std::vector<Look> lookup_file;         // in file order (empty when that is `lookup`)
    bool lookup_in_file_order = false;
    const char* data = nullptr; size_t size = 0; int fd = -1;
    explicit DbReader(const std::string& path) {
        // read_index (src/database_reader.cpp:283-311): an entry per '\n' of the file (a last line without one is not an entry),
        // words separated by blanks and tabs, numbers by strtoul / strtoull; a line of more than three words fails the read (one of
        // fewer than three is undefined there: refused here)
        // (plain decimal words -- every line a writer of this format makes -- are read in place; anything else goes through
        //  strtoul / strtoull on a copy of the word, as before)
        std::string raw;
        try { raw = read_file(path + ".index"); } catch (const std::exception&) { throw std::runtime_error("cannot open " + path + ".index"); }
        auto number = [](const char* s, size_t n, bool wide) -> unsigned long long {
            uint64_t v;
            if (all_digits_u64(s, n, v)) return v;
            const std::string word(s, n);
            return wide ? strtoull(word.c_str(), nullptr, 10) : (unsigned long long)strtoul(word.c_str(), nullptr, 10);
        };
        // both files are cut at line ends into one piece per host thread and parsed side by side (1 M entries: 0.25 s -> 0.03 s)
        auto pieces_of = [](const std::string& t) {
            const size_t T = (size_t)std::max(1, std::min(omp_get_max_threads(), (int)(t.size() >> 16) + 1));
            std::vector<size_t> cut(T + 1, t.size());
            cut[0] = 0;
            for (size_t k = 1; k < T; k++) {
                const size_t from = std::max(cut[k - 1], t.size() / T * k);
                const void* e = from < t.size() ? memchr(t.data() + from, '\n', t.size() - from) : nullptr;
                cut[k] = e ? (size_t)((const char*)e - t.data()) + 1 : t.size();
            }
            return cut;
        };
        {
            const std::vector<size_t> cut = pieces_of(raw);
            const size_t T = cut.size() - 1;
            std::vector<std::vector<Row>> part(T); std::vector<int> bad(T, -1);
#pragma omp parallel for schedule(static, 1) if (T > 1)
            for (long long t = 0; t < (long long)T; t++) {
                const char* w[4]; size_t wl[4];
                for (size_t p0 = cut[(size_t)t]; p0 < cut[(size_t)t + 1];) {
                    const void* e = memchr(raw.data() + p0, '\n', cut[(size_t)t + 1] - p0);
                    if (!e) break;                                         // (only the last piece can end without a line end)
                    const size_t nl = (size_t)((const char*)e - raw.data());
                    const int k = line_words(raw.data() + p0, nl - p0, w, wl);
                    if (k != 3) { bad[(size_t)t] = k; break; }
                    Row r; r.key = (long long)(uint32_t)number(w[0], wl[0], false); r.off = (long long)number(w[1], wl[1], true); r.len = (long long)number(w[2], wl[2], true);
                    part[(size_t)t].push_back(r);
                    p0 = nl + 1;
                }
            }
            for (size_t t = 0; t < T; t++) {
                rows.insert(rows.end(), part[t].begin(), part[t].end());
                if (bad[t] >= 0) throw std::runtime_error(path + ".index: a line of " + std::to_string(bad[t]) + " columns");
            }
        }
        // (written in key order by every writer: the sort is a check then)
        if (!std::is_sorted(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.key < b.key; }))
            std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.key < b.key; });
        // read_lookup (:345-361): std::getline lines (a last line without a line end is one), key = the first word, name = the second
        try { lraw = read_file(path + ".lookup"); } catch (const std::exception&) {}
        {
            const std::vector<size_t> cut = pieces_of(lraw);
            const size_t T = cut.size() - 1;
            std::vector<std::vector<Look>> part(T);
#pragma omp parallel for schedule(static, 1) if (T > 1)
            for (long long t = 0; t < (long long)T; t++) {
                const char* w[4]; size_t wl[4];
                for (size_t p0 = cut[(size_t)t]; p0 < cut[(size_t)t + 1];) {
                    const void* e = memchr(lraw.data() + p0, '\n', cut[(size_t)t + 1] - p0);
                    const size_t nl = e ? (size_t)((const char*)e - lraw.data()) : cut[(size_t)t + 1];
                    if (line_words(lraw.data() + p0, nl - p0, w, wl) >= 2)
                        part[(size_t)t].push_back({(long long)(uint32_t)number(w[0], wl[0], false), (uint64_t)(w[1] - lraw.data()), (uint32_t)std::min<size_t>(wl[1], UINT32_MAX)});
                    p0 = nl + 1;
                }
            }
            for (size_t t = 0; t < T; t++) lookup.insert(lookup.end(), part[t].begin(), part[t].end());
        }
        // a key or a name that comes twice: the later line wins (the reference's stable_sort with "<=" comparators, :313-321, leaves
        // equal elements in reverse order and its look-ups take the first): reversed here before the stable sort by key
        // (keys strictly increasing -- what every writer makes -- : file order IS key order and nothing comes twice)
        lookup_in_file_order = std::adjacent_find(lookup.begin(), lookup.end(), [](const auto& a, const auto& b) { return a.first >= b.first; }) == lookup.end();
        if (!lookup_in_file_order) {
            lookup_file = lookup;
            std::reverse(lookup.begin(), lookup.end());
            std::stable_sort(lookup.begin(), lookup.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        }
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st; fstat(fd, &st); size = (size_t)st.st_size;
        if (size) { void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); if (p == MAP_FAILED) throw std::runtime_error("mmap failed"); data = (const char*)p; }
    }
    ~DbReader() { if (data) munmap((void*)data, size); if (fd >= 0) close(fd); }
    size_t n() const { return rows.size(); }
    // position of a key / of a lookup name among the entries, -1 when absent (reader_get_id, reader_lookup_entry)
    long long id_of_key(long long key) const {
        auto it = std::lower_bound(rows.begin(), rows.end(), key, [](const Row& r, long long k) { return r.key < k; });
        return (it != rows.end() && it->key == key) ? (long long)(it - rows.begin()) : -1;
    }
    long long id_of_name(const std::string& nm) const {
        const auto& lf = lookup_in_file_order ? lookup : lookup_file;
        for (size_t i = lf.size(); i-- > 0;) if (lf[i].len == nm.size() && memcmp(lraw.data() + lf[i].pos, nm.data(), nm.size()) == 0) return id_of_key(lf[i].first);
        return -1;
    }
    std::string name(size_t i) const {
        if (i < lookup.size() && lookup[i].first == rows[i].key && lookup_in_file_order) return std::string(lraw.data() + lookup[i].pos, lookup[i].len);   // line i of both files
        auto it = std::lower_bound(lookup.begin(), lookup.end(), rows[i].key, [](const auto& a, long long k) { return a.first < k; });
        return (it != lookup.end() && it->first == rows[i].key) ? std::string(lraw.data() + it->pos, it->len) : std::to_string(rows[i].key);
    }
    // the stored bytes. MMseqs-made databases end every entry with a NUL: it stays -- the codec, like Foldcomp::read, takes
    // the record length from the header and ignores what follows (a record may legitimately end in zero bytes itself)
    std::string entry(size_t i) const {
        const long long o = rows[i].off, l = rows[i].len;
        if (o < 0 || l < 0 || (size_t)(o + l) > size) throw std::runtime_error("database entry out of range");
        return std::string(data + o, (size_t)l);
    }
};

struct DbWriter {
    struct Row { long long key, off, len; std::string name; };
    std::string path; std::ofstream out; std::vector<Row> rows; long long pos = 0;
    explicit DbWriter(const std::string& p) : path(p), out(p, std::ios::binary) {
        if (!out) throw std::runtime_error("cannot write " + p);
        std::ofstream t(p + ".dbtype", std::ios::binary);
        const int32_t twelve = 12; t.write((const char*)&twelve, 4);
    }
    void append(const char* d, size_t n, long long key, const std::string& name, bool nul) {
        out.write(d, (std::streamsize)n);
        if (nul) out.put('\0');
        const long long len = (long long)n + (nul ? 1 : 0);
        rows.push_back({key, pos, len, name}); pos += len;
    }
    void close() {   // free_writer: index and lookup sorted by key (src/database_writer.cpp:59-73)
        std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.key < b.key; });
        std::ofstream fi(path + ".index"), fl(path + ".lookup");
        for (const Row& r : rows) { fi << r.key << "\t" << r.off << "\t" << r.len << "\n"; fl << r.key << "\t" << r.name << "\t0\n"; }
        out.close();
    }
};
bool is_db(const std::string& p) { return exists(p + ".dbtype"); }

// ---- tar archives, in and out (the reference: lib/microtar driven by TarProcessor, src/input_processor.h:109-198, and by the
//      writers of src/main.cpp:519-523, :666-674, :832-844, src/foldcomp.cpp:1122). AFDB ships its bulk downloads as a plain tar
//      of .pdb.gz / .cif.gz members: the members of a plain tar are byte ranges of one file (InputPlan reads them like database
//      entries, a gzip member goes to the device as it lies in the archive); a gzipped tar (.tar.gz / .tgz, by the NAME as
//      TarProcessor decides) is one DEFLATE stream and is inflated by zlib on the producer thread as the run walks it.
//      What microtar reads and this reader keeps: 512-byte headers; a header whose checksum field starts with a NUL ends the
//      archive; the checksum is the byte sum of the header with its checksum field read as eight blanks, compared with the
//      field's octal number; the size is the octal number of its field; the member's name is the NAME FIELD ONLY, cut to 99
//      characters (the ustar prefix and pax records are not read); a GNU long-name record ('L', 'K') carries the next member's
//      name as its data; members of type '0', '7' and NUL are files, every other type is skipped with its data.
//      Where microtar goes on with stale state (a header that fails its checksum, an archive that ends without a NUL record)
//      this reader stops with an [Error] line: what was read before it is processed. ----
bool is_tar_name(const std::string& p) { return ends_with(p, ".tar") || ends_with(p, ".tar.gz") || ends_with(p, ".tgz"); }   // src/main.cpp:341
struct TarMember { std::string name; uint64_t off = 0, len = 0; };   // off: position of the member's bytes in the (inflated) archive
struct TarStream {
    int fd = -1; gzFile gz = nullptr; uint64_t pos = 0; std::string path, last; bool failed = false;
    ~TarStream() { if (gz) gzclose(gz); if (fd >= 0) close(fd); }
    bool open_path(const std::string& p) {
        path = p;
        if (ends_with(p, ".gz") || ends_with(p, ".tgz")) { gz = gzopen(p.c_str(), "rb"); if (gz) gzbuffer(gz, 1u << 20); return gz != nullptr; }
        fd = open(p.c_str(), O_RDONLY);
        return fd >= 0;
    }
    bool read_exact(void* dst, uint64_t n) {
        uint8_t* d = (uint8_t*)dst; uint64_t got = 0;
        while (got < n) {
            const size_t want = (size_t)std::min<uint64_t>(n - got, 1u << 30);
            const long k = gz ? (long)gzread(gz, d + got, (unsigned)want) : (long)pread(fd, d + got, want, (off_t)(pos + got));
            if (k <= 0) return false;
            got += (uint64_t)k;
        }
        pos += n;
        return true;
    }
    bool skip(uint64_t n) {
        if (gz) { if (n && gzseek(gz, (z_off_t)n, SEEK_CUR) < 0) return false; }
        pos += n;
        return true;
    }
    static uint64_t pad(uint64_t len) { return (512 - len % 512) % 512; }
    static uint64_t octal(const char* f, size_t n) {                // strtoul(field, NULL, 8); GNU's base-256 form for what octal cannot hold
        if ((unsigned char)f[0] & 0x80u) { uint64_t v = (unsigned char)f[0] & 0x7fu; for (size_t i = 1; i < n; i++) v = (v << 8) | (unsigned char)f[i]; return v; }
        size_t i = 0; while (i < n && (f[i] == ' ' || (unsigned)(f[i] - 9) < 5u)) i++;
        uint64_t v = 0; while (i < n && f[i] >= '0' && f[i] <= '7') v = v * 8 + (uint64_t)(f[i++] - '0');
        return v;
    }
    // the next file member: 1 (name and size set; its bytes start at `pos`: the caller reads or skips them, then the padding),
    // 0 at the end of the archive, -1 after an [Error] line
    int next(TarMember& m) {
        std::string long_name; bool have_long = false;
        for (;;) {
            unsigned char h[512];
            if (!read_exact(h, 512)) { fprintf(stderr, "[Error] tar truncated after entry %s\n", last.c_str()); failed = true; return -1; }
            if (h[148] == '\0') return 0;                             // a NUL record
            unsigned sum = 256;
            for (int i = 0; i < 148; i++) sum += h[i];
            for (int i = 156; i < 512; i++) sum += h[i];
            if (sum != (unsigned)octal((const char*)h + 148, 8)) { fprintf(stderr, "[Error] %s: bad tar header checksum after entry %s\n", base_name(path).c_str(), last.c_str()); failed = true; return -1; }
            const uint64_t len = octal((const char*)h + 124, 12);
            const char type = (char)h[156];
            if (type == 'L' || type == 'K') {
                std::string d((size_t)len, '\0');
                if ((len && !read_exact(&d[0], len)) || !skip_pad(len)) { fprintf(stderr, "[Error] cannot read entry %s\n", last.c_str()); failed = true; return -1; }
                long_name.assign(d.c_str()); have_long = true;       // (the name as a C string: the record carries its terminator)
                continue;
            }
            m.name = have_long ? long_name : std::string((const char*)h, strnlen((const char*)h, 99));
            m.len = len; m.off = pos; last = m.name;
            if (type == '0' || type == '7' || type == '\0') return 1;
            if (len && (!skip(len) || !skip_pad(len))) { fprintf(stderr, "[Error] cannot skip entry %s\n", m.name.c_str()); failed = true; return -1; }
            have_long = false;
        }
    }
    bool skip_pad(uint64_t len) { return skip(pad(len)); }
    bool skip_member(const TarMember& m) { return skip(m.len) && skip_pad(m.len); }
};

// a member's header as mtar_write_file_header makes it (lib/microtar: zeros; name; mode 644, owner 0, size, mtime 0 as "%o"; type
// '0'; checksum "%06o" + NUL + blank): no ustar magic, no times -- the same bytes, so the same archive for the same members
void tar_header(uint8_t* h, const std::string& name, uint64_t size) {
    memset(h, 0, 512);
    memcpy(h, name.data(), std::min<size_t>(name.size(), 99));
    snprintf((char*)h + 100, 8, "%o", 0644u);
    snprintf((char*)h + 108, 8, "%o", 0u);
    snprintf((char*)h + 124, 12, "%llo", (unsigned long long)size);
    snprintf((char*)h + 136, 12, "%o", 0u);
    h[156] = '0';
    unsigned sum = 256;
    for (int i = 0; i < 148; i++) sum += h[i];
    for (int i = 156; i < 512; i++) sum += h[i];
    snprintf((char*)h + 148, 8, "%06o", sum);
    h[155] = ' ';
}
inline uint64_t tar_record_bytes(uint64_t len) { return 512 + len + TarStream::pad(len); }

// the members of one job, header + bytes + padding each, back to back in `out` (what mtar_write_file_header + mtar_write_data leave
// in the archive): a job's members are one contiguous range of the archive, placed by the Sequencer like a database's records
template <class Vec, class GetName, class GetPtr, class GetLen>
uint64_t tar_pack_members(Vec& out, size_t n, GetName name, GetPtr ptr, GetLen len) {
    uint64_t total = 0;
    for (size_t q = 0; q < n; q++) total += tar_record_bytes(len(q));
    out.resize(total);
    uint64_t pos = 0;
    for (size_t q = 0; q < n; q++) {
        const uint64_t l = len(q), pd = TarStream::pad(l);
        tar_header(out.data() + pos, name(q), l);
        if (l) memcpy(out.data() + pos + 512, ptr(q), l);
        if (pd) memset(out.data() + pos + 512 + l, 0, pd);
        pos += 512 + l + pd;
    }
    return total;
}
// the end of an archive: two NUL records (mtar_write_finalize)
bool tar_finalize(int fd, uint64_t at) {
    uint8_t z[1024]; memset(z, 0, sizeof z);
    return pwrite(fd, z, sizeof z, (off_t)at) == (ssize_t)sizeof z;
}

struct Options {
    std::string mode, input, output;
    int brk = 25, digits = 1, ext_mode = 0;
    bool alt = false, overwrite = false, recursive = false, skip_discontinuous = false, use_title = false, db = false;
    bool tar = false;           // -z / --tar, or an output that ends in .tar: the outputs become the members of one tar archive (src/main.cpp:333-335)
    bool single = false;        // one structure / FCZ file in, one file out
    bool host_parse = false;    // --host-parse: compress parses on the host threads even where the device could (A/B, debugging)
    int job_files = 0;          // --job-files N: files per device job of `compress` (0: the default rule)
    bool host_inflate = false;  // --host-inflate: gzipped inputs are inflated by the reader threads (zlib) instead of on the device (A/B)
    bool check = false;         // --check: decompress skips entries that fail Foldcomp::checkValidity (src/main.cpp:629-636)
    bool merge = true;          // --no-merge: extract writes one file per entry instead of one merged file (src/main.cpp:171-195)
    bool file_input = false;    // -f / --file: <input> is a text file that lists the inputs, one per line (src/main.cpp:304-325)
    std::string id_list;        // --id-list FILE: only these entries of a database input (src/input_processor.h:287-299)
    int id_mode = 1;            // --id-mode 0: the list holds keys, 1: names
    std::vector<std::string> inputs;   // what <input> expands to (itself, or the lines of the -f file: containers first, then files)
    int gpus = 1;               // --gpus N: devices used by compress (0 = every visible device)
    int workers_per_gpu = 2;    // host threads (each with its own ctx and stream) per device
    int write_threads = 8;      // threads a decompress worker writes its job's text with (set from -t: threads / workers)
    bool json_stats = false;    // --json-stats: one JSON line with counts and wall times on stdout
    int threads_said = 1;       // -t as given (the reference's default: 1): only for the announcement lines
    int shard_rank = 0, shard_world = 1;   // --shard R/N: this process is rank R of N of a sharded database run (see InputPlan)
    int device = 0;             // --device D: first HIP device of this process (a rank of a sharded run drives device LOCAL_RANK)
    bool place = false;         // --place: decompress -d as a rank of a sharded run: sizes pass over the range, counts on stdout, then the
                                // placement (`key0 off0 total` on stdin) and ONE write of every record at its final offset (see run_decompress)
};

// ---- the inputs of a run, as a stream ------------------------------------------------------------------------------------------
// The reference's driver walks a directory or a database entry by entry (src/input_processor.h:85-101, :237-257). Here a run's
// inputs -- files of directories, entries of databases, in the order they are listed -- form ONE sequence of items, and a process
// takes a contiguous range of it: everything (one process), or, as rank R of N (--shard R/N), the range whose cumulative input
// bytes lie between R/N and (R+1)/N of the total (contiguous ranges balanced by bytes ~ residues). Every rank
// computes the same cuts from the same listing; no rank ever holds another rank's entries.
//
// Databases are streamed. free_writer (src/database_writer.cpp:59-73) writes .index and .lookup sorted by key, one line per entry:
// entry i of the reader (rows sorted by key, src/database_reader.cpp:109) IS line i of both files, so a rank needs only its own
// stretch of lines. One pass over both files (DbScan) checks exactly that -- keys strictly increasing, the same key on line i of
// both, three / at least two clean words per line -- and keeps the file positions and the cumulative entry bytes of every 4 096th
// line (214 M entries: 52 k marks). A database that is not of that shape (hand-edited, doubled keys, no lookup, --id-list) is read
// through DbReader instead: correct, but with every row and name in memory.
struct LineFile {                              // forward reader of a text file, line by line, with the file position of each line
    int fd = -1; std::vector<char> buf; size_t a = 0, b = 0; uint64_t pos = 0; bool eof = false;   // pos: file position of buf[a]
    ~LineFile() { if (fd >= 0) close(fd); }
    bool open_at(const std::string& path, uint64_t at = 0) {
        if (fd >= 0) close(fd);
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        if (buf.empty()) buf.resize(1 << 20);
        a = b = 0; pos = at; eof = false;
        return lseek(fd, (off_t)at, SEEK_SET) == (off_t)at;
    }
    // the next line without its line end; false at the end of the file (read_index counts line ENDS: a last line without one is
    // not an entry, src/database_reader.cpp:283-311)
    bool next(const char*& s, size_t& n) {
        for (;;) {
            if (const void* e = a < b ? memchr(buf.data() + a, '\n', b - a) : nullptr) {
                s = buf.data() + a; n = (size_t)((const char*)e - s);
                pos += n + 1; a += n + 1;
                return true;
            }
            if (eof) return false;
            if (a > 0) { memmove(buf.data(), buf.data() + a, b - a); b -= a; a = 0; }
            if (b == buf.size()) buf.resize(buf.size() * 2);
            const ssize_t k = read(fd, buf.data() + b, buf.size() - b);
            if (k < 0) throw std::runtime_error("read failed");
            if (k == 0) eof = true; else b += (size_t)k;
        }
    }
};

struct DbScan {
    static constexpr uint64_t STRIDE = 4096;
    struct Mark { uint64_t ipos, lpos, cum; };   // positions of line k x STRIDE in .index / .lookup, entry bytes before it
    std::vector<Mark> marks; uint64_t n = 0, bytes = 0; std::string path, why;
    // true: the database can be streamed (see above); false: `why` says what stands in the way
    bool scan(const std::string& db) {
        path = db; marks.clear(); n = bytes = 0;
        LineFile fi, fl;
        if (!fi.open_at(db + ".index")) throw std::runtime_error("cannot open " + db + ".index");
        if (!fl.open_at(db + ".lookup")) { why = "no .lookup file"; return false; }
        const char *s, *t; size_t sn, tn; const char* w[4]; size_t wl[4];
        uint64_t prev = 0;
        for (;;) {
            const uint64_t ipos = fi.pos, lpos = fl.pos;
            const bool hi_ = fi.next(s, sn), hl = fl.next(t, tn);
            if (!hi_ || !hl) {
                // (a lookup may end without a line end: std::getline still reads that line -- not streamable, rare)
                if (hi_ != hl || fi.a < fi.b || fl.a < fl.b) { why = "index and lookup differ in their number of lines"; return false; }
                break;
            }
            uint64_t key, off, len, lkey;
            if (line_words(s, sn, w, wl) != 3 || !all_digits_u64(w[0], wl[0], key) || !all_digits_u64(w[1], wl[1], off) || !all_digits_u64(w[2], wl[2], len) || key > UINT32_MAX) {
                why = "an index line that is not three plain numbers"; return false; }
            if (line_words(t, tn, w, wl) < 2 || !all_digits_u64(w[0], wl[0], lkey)) { why = "a lookup line without a key and a name"; return false; }
            if (lkey != key) { why = "line " + std::to_string(n) + " of index and lookup carry different keys"; return false; }
            if (n > 0 && key <= prev) { why = "keys are not strictly increasing"; return false; }
            prev = key;
            if (n % STRIDE == 0) marks.push_back({ipos, lpos, bytes});
            n++; bytes += len;
        }
        return true;
    }
};

struct InputItem { int kind = 0; std::string name; uint64_t off = 0, len = 0; int src = 0; };   // kind 0: a file (name = path, len = its size or UINT64_MAX: not asked yet); 1: a database entry (name = lookup name) or a tar member (name = the member's name)

struct InputPlan {
    struct Src {
        int kind = 0;                          // 0 files, 1 database streamed, 2 database through DbReader, 3 tar (members listed up front), 4 gzipped tar (walked as it inflates)
        std::string path;
        std::vector<TarMember> members;        // kind 3
        std::vector<uint8_t> arena; uint64_t arena0 = 0;   // kind 4: the bytes of the members handed out since the last release(); arena[0] is stream position arena0
        std::vector<std::string> files; std::vector<uint64_t> fsize;
        DbScan scan;
        std::unique_ptr<DbReader> db; std::vector<size_t> ids;
        std::vector<std::string> id_names;     // --id-list: an entry goes by the list's own line -- with --id-mode 0 that is its KEY (src/input_processor.h:262-278)
        int dfd = -1; uint64_t dsize = 0;
        uint64_t n = 0, bytes = 0, lo = 0, hi = 0;   // items, their bytes, this process's items [lo, hi)
        ~Src() { if (dfd >= 0) close(dfd); }
        uint64_t weight(uint64_t i) const { return kind == 0 ? fsize[i] : kind == 3 ? members[i].len : (uint64_t)db->rows[ids[i]].len; }   // kinds 0, 2 and 3
    };
    std::vector<std::unique_ptr<Src>> srcs;
    uint64_t n_items = 0, n_mine = 0, bytes_total = 0;
    bool streamed_all = true;
    bool unsized = false;                      // a gzipped tar among the inputs: its members are counted as the run walks it
    bool hold = false;                         // the consumer reads the entries of a gzipped tar after for_each's callback returned: it calls release()
    void release() { for (auto& sp : srcs) if (sp->kind == 4) { sp->arena0 += sp->arena.size(); sp->arena.clear(); } }

    // first item e of a source with (bytes of the items before it) >= t; n when there is none
    static uint64_t first_at_least(Src& s, uint64_t t) {
        if (t == 0) return 0;
        if (t > s.bytes) return s.n;
        if (s.kind != 1) {
            uint64_t c = 0;
            for (uint64_t i = 0; i < s.n; i++) { if (c >= t) return i; c += s.weight(i); }
            return s.n;
        }
        const auto& mk = s.scan.marks;
        size_t m = (size_t)(std::lower_bound(mk.begin(), mk.end(), t, [](const DbScan::Mark& a, uint64_t v) { return a.cum < v; }) - mk.begin());
        if (m < mk.size() && mk[m].cum >= t && (m == 0 || mk[m - 1].cum >= t)) return (uint64_t)m * DbScan::STRIDE;   // (m == 0 only: t == 0 handled above)
        const size_t from = m == 0 ? 0 : m - 1;                       // the cut lies after mark `from`
        LineFile fi;
        if (!fi.open_at(s.path + ".index", mk[from].ipos)) throw std::runtime_error("cannot open " + s.path + ".index");
        uint64_t c = mk[from].cum, e = (uint64_t)from * DbScan::STRIDE;
        const char* q; size_t qn; const char* w[4]; size_t wl[4];
        while (e < s.n) {
            if (c >= t) return e;
            uint64_t len = 0;
            if (!fi.next(q, qn) || line_words(q, qn, w, wl) != 3 || !all_digits_u64(w[2], wl[2], len)) throw std::runtime_error(s.path + ".index changed during the run");
            c += len; e++;
        }
        return s.n;
    }

    // list the inputs; cut this process's range. `quiet`: messages about ids that are not found come from rank 0 only
    void build(const Options& o) {
        const bool quiet = o.shard_rank != 0;
        for (const std::string& input : o.inputs) {
            std::unique_ptr<Src> sp(new Src()); Src& s = *sp;
            s.path = input;
            if (!is_dir(input) && is_tar_name(input)) {
                if (ends_with(input, ".gz") || ends_with(input, ".tgz")) {
                    // one DEFLATE stream: no member can be reached without inflating what lies before it
                    if (o.shard_world > 1) throw std::runtime_error(input + ": a gzipped tar cannot be cut into ranges (inflate it to a plain .tar for a sharded run)");
                    s.kind = 4; unsized = true;
                } else {
                    s.kind = 3;
                    TarStream ts;
                    if (!ts.open_path(input)) throw std::runtime_error("open tar " + input + " failed.");
                    TarMember m;
                    while (ts.next(m) == 1) { s.bytes += m.len; s.members.push_back(m); if (!ts.skip_member(m)) break; }
                    s.n = s.members.size();
                    s.dfd = open(input.c_str(), O_RDONLY);
                    if (s.dfd < 0) throw std::runtime_error("cannot open " + input);
                    struct stat st; fstat(s.dfd, &st); s.dsize = (uint64_t)st.st_size;
                }
            } else if (is_db(input)) {
                if (o.id_list.empty() && s.scan.scan(input)) { s.kind = 1; s.n = s.scan.n; s.bytes = s.scan.bytes; }
                else {
                    if (o.id_list.empty() && !quiet) fprintf(stderr, "[Info] %s is read into memory (%s)\n", input.c_str(), s.scan.why.c_str());
                    s.kind = 2; streamed_all = false;
                    s.db.reset(new DbReader(input));
                    if (!o.id_list.empty()) {
                        std::ifstream f(o.id_list);
                        if (!f && !quiet) fprintf(stderr, "[Error] user id '%s' does not exist.\n", o.id_list.c_str());
                        std::string line;
                        while (f && std::getline(f, line)) {
                            line = strip(line);
                            if (line.empty()) continue;
                            const long long id = o.id_mode == 0 ? s.db->id_of_key(atoll(line.c_str())) : s.db->id_of_name(line);
                            if (id < 0) { if (!quiet) fprintf(stderr, "[Warning] %s not found in database.\n", line.c_str()); continue; }
                            s.ids.push_back((size_t)id); s.id_names.push_back(line);
                        }
                    } else { s.ids.resize(s.db->n()); for (size_t i = 0; i < s.db->n(); i++) s.ids[i] = i; }
                    s.n = s.ids.size();
                    for (size_t i : s.ids) s.bytes += (uint64_t)std::max<long long>(0, s.db->rows[i].len);
                }
                if (s.kind == 1) {
                    s.dfd = open(input.c_str(), O_RDONLY);
                    if (s.dfd < 0) throw std::runtime_error("cannot open " + input);
                    struct stat st; fstat(s.dfd, &st); s.dsize = (uint64_t)st.st_size;
                }
            } else {
                if (is_dir(input)) list_files(input, o.recursive, s.files); else s.files.push_back(input);
                s.n = s.files.size();
                if (o.shard_world > 1) {                                // sizes only where a cut needs them
                    s.fsize.assign(s.n, 0);
#pragma omp parallel for schedule(dynamic, 256)
                    for (long long i = 0; i < (long long)s.n; i++) { struct stat st; if (stat(s.files[(size_t)i].c_str(), &st) == 0 && S_ISREG(st.st_mode)) s.fsize[(size_t)i] = (uint64_t)st.st_size; }
                    for (uint64_t v : s.fsize) s.bytes += v;
                }
            }
            n_items += s.n; bytes_total += s.bytes;
            srcs.push_back(std::move(sp));
        }
        // cuts: item g belongs to rank r iff cut_r <= g < cut_(r+1), cut_r = the first item with ceil(total x r / world) bytes before it
        // (cut_0 = 0, cut_world = every item); integer arithmetic, so every rank and every implementation agrees
        const int R = o.shard_rank, W = std::max(1, o.shard_world);
        auto target = [&](int r) { return (uint64_t)(((unsigned __int128)bytes_total * (unsigned)r + (unsigned)(W - 1)) / (unsigned)W); };
        auto cut = [&](int r, std::vector<uint64_t>& at) {              // -> per source, the first item at or after the cut
            at.assign(srcs.size(), 0);
            if (r >= W) { for (size_t k = 0; k < srcs.size(); k++) at[k] = srcs[k]->n; return; }
            if (r <= 0) return;
            const uint64_t t = target(r);
            uint64_t before = 0; bool found = false;
            for (size_t k = 0; k < srcs.size(); k++) {
                Src& s = *srcs[k];
                if (found) { at[k] = 0; continue; }
                // the cut falls into this source when some item of it has >= t bytes before it (t - before <= s.bytes counts the
                // position after the last item too: that one belongs to the next source's first item)
                const uint64_t e = t <= before ? 0 : first_at_least(s, t - before);
                if (e < s.n) { at[k] = e; found = true; } else { at[k] = s.n; before += s.bytes; }
            }
        };
        std::vector<uint64_t> a, b;
        cut(R, a); cut(R + 1, b);
        for (size_t k = 0; k < srcs.size(); k++) { srcs[k]->lo = a[k]; srcs[k]->hi = std::max(a[k], b[k]); n_mine += srcs[k]->hi - srcs[k]->lo; }
        for (auto& sp : srcs) if (sp->kind == 4) { sp->lo = 0; sp->hi = UINT64_MAX; }
    }

    // this process's items in order: f(const InputItem&)
    template <class F> void for_each(F&& f) {
        for (size_t k = 0; k < srcs.size(); k++) {
            Src& s = *srcs[k];
            if (s.lo >= s.hi) continue;
            InputItem it; it.src = (int)k;
            if (s.kind == 3) {
                it.kind = 1;
                for (uint64_t i = s.lo; i < s.hi; i++) { it.name = s.members[i].name; it.off = s.members[i].off; it.len = s.members[i].len; f(it); }
            } else if (s.kind == 4) {
                it.kind = 1;
                TarStream ts;
                if (!ts.open_path(s.path)) throw std::runtime_error("open tar " + s.path + " failed.");
                TarMember m;
                while (ts.next(m) == 1) {
                    if (!hold) release();
                    const size_t at = s.arena.size();
                    s.arena.resize(at + m.len);
                    if ((m.len && !ts.read_exact(s.arena.data() + at, m.len)) || !ts.skip_pad(m.len)) { s.arena.resize(at); fprintf(stderr, "[Error] cannot read entry %s\n", m.name.c_str()); break; }
                    it.name = m.name; it.off = s.arena0 + at; it.len = m.len;
                    s.n++; n_items++; n_mine++; s.bytes += m.len; bytes_total += m.len;
                    f(it);
                }
            } else if (s.kind == 0) {
                it.kind = 0;
                for (uint64_t i = s.lo; i < s.hi; i++) { it.name = s.files[i]; it.len = s.fsize.empty() ? UINT64_MAX : s.fsize[i]; f(it); }
            } else if (s.kind == 2) {
                it.kind = 1;
                for (uint64_t i = s.lo; i < s.hi; i++) {
                    const auto& r = s.db->rows[s.ids[i]];
                    it.name = s.id_names.empty() ? s.db->name(s.ids[i]) : s.id_names[i]; it.off = (uint64_t)r.off; it.len = (uint64_t)r.len; f(it);
                }
            } else {
                it.kind = 1;
                const size_t m = (size_t)(s.lo / DbScan::STRIDE);
                LineFile fi, fl;
                if (!fi.open_at(s.path + ".index", s.scan.marks[m].ipos) || !fl.open_at(s.path + ".lookup", s.scan.marks[m].lpos)) throw std::runtime_error("cannot open the index of " + s.path);
                const char *q, *t; size_t qn, tn; const char* w[4]; size_t wl[4];
                for (uint64_t i = (uint64_t)m * DbScan::STRIDE; i < s.hi; i++) {
                    if (!fi.next(q, qn) || !fl.next(t, tn)) throw std::runtime_error(s.path + ".index changed during the run");
                    if (i < s.lo) continue;
                    uint64_t off = 0, len = 0;
                    if (line_words(q, qn, w, wl) != 3 || !all_digits_u64(w[1], wl[1], off) || !all_digits_u64(w[2], wl[2], len)) throw std::runtime_error(s.path + ".index changed during the run");
                    if (line_words(t, tn, w, wl) < 2) throw std::runtime_error(s.path + ".lookup changed during the run");
                    it.name.assign(w[1], wl[1]); it.off = off; it.len = len; f(it);
                }
            }
        }
    }

    // the bytes of a database entry into dst (exactly it.len); false: the entry lies outside the data file
    // does the entry lie inside its data file? (asked BEFORE a buffer is sized from the index's numbers: one corrupt line must not
    // become a bad_alloc that ends the run; off + len cannot wrap this way -- an index number may have 19 digits)
    bool entry_in_range(const InputItem& it) const {
        const Src& s = *srcs[(size_t)it.src];
        if (s.kind == 4) return it.off >= s.arena0 && it.len <= s.arena.size() && it.off - s.arena0 <= s.arena.size() - it.len;
        const uint64_t size = s.kind == 2 ? (uint64_t)s.db->size : s.dsize;
        return it.len <= size && it.off <= size - it.len;
    }
    bool read_entry(const InputItem& it, uint8_t* dst) const {
        const Src& s = *srcs[(size_t)it.src];
        if (!entry_in_range(it)) return false;
        if (s.kind == 2) { memcpy(dst, s.db->data + it.off, it.len); return true; }
        if (s.kind == 4) { memcpy(dst, s.arena.data() + (it.off - s.arena0), it.len); return true; }
        uint64_t got = 0;
        while (got < it.len) { const ssize_t k = pread(s.dfd, dst + got, it.len - got, (off_t)(it.off + got)); if (k <= 0) return false; got += (uint64_t)k; }
        return true;
    }
};

struct Fragment {
    std::string out_name, db_name; AtomTable atoms; std::string title;   // db_name: lookup name = the input file's stem
    size_t file = 0;                           // which input of the run it came from (fragments of one input are written in order, write_fragments_in_order)
    // filled by the parse threads of the compress pipeline (Batch::prepare): what the batch needs besides the atoms, or why not
    bool prepared = false; Batch::Prepared prep; std::string prep_err;
};

// one structure file -> its fragments (src/main.cpp:455-508)
// (out_stem, ext) = getFileParts of the input's base name, or of the OUTPUT path for a single-file run (src/main.cpp:444-457):
// they name the fragments and decide the suffix (isCompressible, :498-502)
void fragments_from_memory(const char* data, size_t size, const std::string& base, const std::string& out_stem, const std::string& ext,
                           bool to_dir_or_file, const Options& o, std::vector<Fragment>& out, bool inflated = false);   // inflated: the bytes of a .gz file after gunzip
void fragments_of(const std::string& path, const std::string& out_stem, const std::string& ext, bool to_dir_or_file, const Options& o,
                  std::vector<Fragment>& out) {
    static thread_local FileImage image;
    read_file_into(path, image);
    fragments_from_memory(image.data(), image.n, base_name(path), out_stem, ext, to_dir_or_file, o, out);
}
// the same from a file image that is already in memory
void fragments_from_memory(const char* data, size_t size, const std::string& base, const std::string& out_stem, const std::string& ext,
                           bool to_dir_or_file, const Options& o, std::vector<Fragment>& out, bool inflated) {
    std::string plain = base, unz;
    if (ends_with(base, ".gz")) {
        plain = base.substr(0, base.size() - 3);
        if (!inflated) { unz = gunzip(std::string(data, size)); data = unz.data(); size = unz.size(); }
    }
    std::string title;
    AtomTable t;
    // PDB or mmCIF by what the bytes say, not by the name (StructureReader::loadFromBuffer); the reference's reader rules for
    // either, then removeAlternativePosition
    { t = parse_structure_gemmi(data, size, title); AtomTable k = remove_alternative_position(t); table_pool().put(std::move(t)); t = std::move(k); }
    if (t.size() == 0) { fprintf(stderr, "[Error] No atoms found in the input file: %s\n", base.c_str()); return; }
    if (title.empty() || title == base) title = out_stem;            // src/main.cpp:465
    const std::vector<Range> chains = identify_chains(t);
    for (const Range& cs : chains) {
        const std::vector<Range> frags = identify_discontinuous(t, cs);
        if (o.skip_discontinuous && frags.size() > 1) { fprintf(stderr, "Skipping discontinuous chain: %s\n", base.c_str()); continue; }
        for (size_t j = 0; j < frags.size(); j++) {
            std::string fname = out_stem;
            if (chains.size() > 1) fname += t.chain_name(cs.a);
            if (frags.size() > 1) fname += "_" + std::to_string(j);
            // (src/main.cpp:498-502: "." + the extension also when there is none -- a file `d1asha_` becomes `d1asha_.`)
            if (to_dir_or_file) fname += is_compressible(out_stem, ext) ? ".fcz" : "." + ext;
            // the usual file is one chain in one piece: its table moves into the fragment instead of being copied
            Fragment f;
            f.out_name = fname; f.db_name = out_stem; f.title = title;
            if (chains.size() == 1 && frags.size() == 1 && frags[j].a == 0 && frags[j].b == t.size()) f.atoms = std::move(t);
            else f.atoms = t.slice(frags[j].a, frags[j].b);
            out.push_back(std::move(f));
        }
    }
}

// fragments of many structure files, parsed on all host threads (the reference: `omp parallel for` over entries,
// src/input_processor.h:85-101), kept in file order; messages are printed in file order too
void fragments_of_files(const std::vector<std::string>& files, size_t a, size_t b, bool single, const std::string& output, bool to_dir_or_file,
                        const Options& o, std::vector<Fragment>& out, bool prepare = false) {
    std::vector<std::vector<Fragment>> per(b - a);
    std::vector<std::string> err(b - a);
#pragma omp parallel for schedule(dynamic, 4)
    for (long long i = (long long)a; i < (long long)b; i++) {
        std::string stem, ext;
        file_parts(base_name(files[i]), stem, ext);
        std::string out_stem = stem;
        if (single) file_parts(base_name(output), out_stem, ext);
        try { fragments_of(files[i], out_stem, ext, to_dir_or_file, o, per[i - a]); }
        catch (const std::exception& e) { err[i - a] = "[Error] " + base_name(files[i]) + ": " + e.what() + "\n"; }
        if (prepare) for (Fragment& f : per[i - a]) {
            try { f.prep = Batch::prepare(f.atoms, o.brk); } catch (const std::exception& e) { f.prep_err = e.what(); }
            f.prepared = true;
        }
    }
    for (size_t i = 0; i < b - a; i++) {
        if (!err[i].empty()) fputs(err[i].c_str(), stderr);
        for (Fragment& f : per[i]) { f.file = a + i; out.push_back(std::move(f)); }
    }

}

// peak resident set of this process: VmHWM (ru_maxrss starts from the PARENT's peak after fork + exec, so a small child of a large
// parent would report the parent)
long max_rss_kb() {
    if (FILE* f = fopen("/proc/self/status", "r")) {
        char line[256]; long kb = -1;
        while (fgets(line, sizeof line, f)) if (sscanf(line, "VmHWM: %ld", &kb) == 1) break;
        fclose(f);
        if (kb >= 0) return kb;
    }
    struct rusage u; return getrusage(RUSAGE_SELF, &u) == 0 ? u.ru_maxrss : -1;
}

// ---- compress ----
// ---- compress: a pipeline over every GPU of the node -------------------------------------------------------------------
// The reference's driver is one `omp parallel for` over the entries with a critical section around the writer
// (src/input_processor.h:237-257, src/main.cpp:510-530). Structures are independent, so here:
//   producer   the calling thread walks the inputs in order; chunks of files are parsed on all host threads (OpenMP) into
//              fragments, and every BATCH_CHAINS fragments become one job;
//   workers    gpus x workers_per_gpu host threads, each with its own fcz_ctx (own stream) on its device and its own batch and
//              blob buffers in page-locked memory: while one worker of a GPU runs the kernels of its job, the other one's
//              host-to-device / device-to-host copies are in flight on the DMA engines. Jobs go to whichever worker is free;
//   sequencer  record sizes are known on the host before the GPU is called (Foldcomp::getSize), so a job's slice of the
//              database file is assigned in job order as soon as every earlier job has announced its size; workers pwrite
//              their blob at that offset: no ordering between the GPUs' writes, no copy through one writer thread;
//   index      rows (job, position, offset, length, name) are merged at the end, keys are numbered in input order over the
//              records that compressed (the reference's key++ under `omp critical` is schedule dependent; per-record
//              bytes and a stable order are what is kept), and .index / .lookup / .dbtype are written as free_writer does.
struct CompressJob { size_t index = 0; std::vector<Fragment> frags; };

template <class T> struct JobQueue {           // bounded hand-over between the producer and the workers
    std::mutex m; std::condition_variable cv_put, cv_get; std::deque<T> q; size_t cap; bool closed = false;
    explicit JobQueue(size_t c) : cap(c) {}
    void put(T&& v) { std::unique_lock<std::mutex> l(m); cv_put.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cv_get.notify_one(); }
    bool get(T& v) {
        std::unique_lock<std::mutex> l(m); cv_get.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front()); q.pop_front(); cv_put.notify_one(); return true;
    }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; cv_get.notify_all(); }
};

// Jobs take their turn in job order: a job's byte range in the data file = the total size of all earlier jobs, its keys = the
// records of all earlier jobs, and -- because the turns come in order -- its .index / .lookup lines are appended right there
// (free_writer's formats, src/database_writer.cpp:59-73). Nothing per record is kept in memory: an index of 214 M entries is
// streamed, not collected and sorted at the end.
struct Sequencer {
    std::mutex m; std::condition_variable cv; size_t next = 0; uint64_t pos = 0; long long key = 0;
    FILE* fi = nullptr; FILE* fl = nullptr;
    bool open_index(const std::string& db, const std::string& tag = "") {      // tag ".R": rank R > 0 of a placed run writes its own line files
        fi = fopen((db + ".index" + tag).c_str(), "w"); fl = fopen((db + ".lookup" + tag).c_str(), "w");
        return fi && fl;
    }
    // lens / names: the job's records in the order they lie in its byte range (both null: a job without records)
    uint64_t claim(size_t job, uint64_t bytes, const std::vector<uint64_t>* lens = nullptr, const std::vector<std::string>* names = nullptr) {
        std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return next == job; });
        const uint64_t at = pos;
        if (lens && names && fi && fl) {
            uint64_t o = at;
            for (size_t q = 0; q < lens->size(); q++) {
                fprintf(fi, "%lld\t%llu\t%llu\n", key, (unsigned long long)o, (unsigned long long)(*lens)[q]);
                fprintf(fl, "%lld\t%s\t0\n", key, (*names)[q].c_str());
                key++; o += (*lens)[q];
            }
        }
        pos += bytes; next++; cv.notify_all(); return at;
    }
    // closes the index files; `keep` false (a failed run): they and the data file are removed -- a database without its index
    // is unusable and one with a partial index looks complete
    void finish(const std::string& db, bool keep) {
        if (fi) fclose(fi);
        if (fl) fclose(fl);
        fi = fl = nullptr;
        if (!keep) { unlink(db.c_str()); unlink((db + ".index").c_str()); unlink((db + ".lookup").c_str()); unlink((db + ".dbtype").c_str()); return; }
        std::ofstream t(db + ".dbtype", std::ios::binary);
        const int32_t twelve = 12; t.write((const char*)&twelve, 4);
    }
};

// many pieces, one after the other, from file position `off` on (pwritev in runs of at most 512 pieces; partial writes resumed)
void pwritev_all(int fd, std::vector<iovec>& iov, uint64_t off) {
    size_t i = 0;
    while (i < iov.size()) {
        if (iov[i].iov_len == 0) { i++; continue; }
        const int cnt = (int)std::min<size_t>(iov.size() - i, 512);
        ssize_t w = pwritev(fd, iov.data() + i, cnt, (off_t)off);
        if (w <= 0) throw std::runtime_error("pwritev failed");
        off += (uint64_t)w;
        while (w > 0 && i < iov.size()) {
            if ((size_t)w >= iov[i].iov_len) { w -= (ssize_t)iov[i].iov_len; i++; }
            else { iov[i].iov_base = (char*)iov[i].iov_base + w; iov[i].iov_len -= (size_t)w; w = 0; }
        }
    }
}
// The fragments of the inputs as files of a directory, in the order the reference's lambda meets them (src/main.cpp:466-531): it walks
// a file's fragments in order and RETURNS at the first output name that already exists -- a file of an earlier run, or an earlier
// fragment of the same file under the same name (chain A, chain B, chain A again) -- unless -y, where a later fragment replaces the
// earlier one. A fragment this host REFUSES (p == nullptr; the reference writes its shifted record) still takes
// its place in that order: what stands under a name here is what the reference writes under it, or nothing -- never another fragment.
struct FragOut { size_t file; uint32_t sub; std::string name; const uint8_t* p; uint64_t len; };
template <class PathOf>
void write_fragments_in_order(std::vector<FragOut>& ev, PathOf path_of, bool overwrite) {
    std::stable_sort(ev.begin(), ev.end(), [](const FragOut& a, const FragOut& b) { return a.file != b.file ? a.file < b.file : a.sub < b.sub; });
    std::unordered_set<std::string> seen; bool stopped = false; size_t cur = SIZE_MAX;
    for (const FragOut& f : ev) {
        if (f.file != cur) { cur = f.file; seen.clear(); stopped = false; }
        if (stopped) continue;
        const std::string path = path_of(f);
        if (!overwrite) {
            if (seen.count(f.name) || exists(path)) { fprintf(stderr, "[Error] Output file already exists: %s\n", base_name(path).c_str()); stopped = true; continue; }
            seen.insert(f.name);
            if (f.p) write_out(path, (const char*)f.p, f.len, true);
        } else {
            if (f.p) write_out(path, (const char*)f.p, f.len, true);
            else if (seen.count(f.name)) unlink(path.c_str());
            seen.insert(f.name);
        }
    }
}
void pwrite_all(int fd, const uint8_t* p, uint64_t n, uint64_t off) {
    while (n) {
        const ssize_t w = pwrite(fd, p, (size_t)std::min<uint64_t>(n, 1u << 30), (off_t)off);
        if (w <= 0) throw std::runtime_error("pwrite failed");
        p += w; n -= (uint64_t)w; off += (uint64_t)w;
    }
}

int run_compress_device(const Options& o, InputPlan& plan, const std::string& output);
int run_compress(const Options& o) {
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    const bool single = o.single;
    const std::string output = o.output;             // main() resolved the reference's defaults (src/main.cpp:356-369)
    const int n_dev = fcz_device_count();
    if (n_dev <= 0) { fprintf(stderr, "[Error] %s\n", fcz_status_string(FCZ_E_NO_DEVICE)); return 1; }
    const int gpus = o.gpus <= 0 ? n_dev - o.device : o.gpus;
    if (gpus < 1 || o.device + gpus > n_dev) { fprintf(stderr, "[Error] --gpus %d from device %d but only %d device(s) are visible\n", gpus, o.device, n_dev); return 1; }
    std::vector<std::string> files;
    if (single) files.push_back(o.inputs[0]);
    else {
        // directories of files and databases of file images: this process's range of the listing (all of it unless --shard)
        InputPlan plan;
        try { plan.build(o); } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); return 1; }
        // the structure ingest runs on the device (the host threads only read)
        if (!o.host_parse) return run_compress_device(o, plan, output);
        bool db_items = false;
        plan.for_each([&](const InputItem& it) { if (it.kind == 0) files.push_back(it.name); else db_items = true; });
        if (db_items) { fprintf(stderr, "[Error] --host-parse reads structure files, not database entries\n"); return 1; }
    }
    const int n_workers = gpus * std::max(1, o.workers_per_gpu);
    pinned_enabled() = true;
    int db_fd = -1;
    const bool placed = o.db || o.tar;          // one output file: a job's records are placed in it by the Sequencer
    if (placed) {
        db_fd = open(output.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
        if (db_fd < 0) { fprintf(stderr, "[Error] cannot write %s\n", output.c_str()); return 1; }
    } else if (!single) make_dir(output);

    JobQueue<CompressJob> queue((size_t)n_workers + 2);
    Sequencer seq;
    if (o.db && !seq.open_index(output)) { fprintf(stderr, "[Error] cannot write %s.index\n", output.c_str()); return 1; }
    std::atomic<bool> hard_fail{false};
    std::atomic<uint64_t> n_res{0}, n_frag_ok{0}, n_bytes{0}, n_atoms{0};
    std::vector<double> gpu_busy(n_workers, 0.0), ctx_ready(n_workers, 0.0);   // ctx_ready: seconds after the start until the worker had its ctx
    std::vector<std::thread> workers;
    for (int w = 0; w < n_workers; w++) workers.emplace_back([&, w]() {
        fcz_ctx* ctx = nullptr;
        if (fcz_ctx_create(o.device + w % gpus, &ctx) != FCZ_OK) { hard_fail = true; fprintf(stderr, "[Error] no ctx on device %d\n", o.device + w % gpus); }
        ctx_ready[w] = std::chrono::duration<double>(clk::now() - t_start).count();
        Batch b;                         // reused: its page-locked buffers grow to the largest job and stay
        pvec<uint8_t> blob;
        CompressJob job;
        while (queue.get(job)) {
            std::vector<size_t> kept;
            b.clear();
            {   // one growth step of the page-locked buffers per job at most (pinning memory is expensive), none once they are large
                // enough; the per-residue arrays are sized from the residue boundaries the parse threads already found
                size_t na = 0, nr = 0;
                for (const Fragment& f : job.frags) { na += f.atoms.size(); nr += f.prepared && f.prep_err.empty() ? f.prep.ro.size() - 1 : f.atoms.size() / 3 + 1; }
                b.x.reserve(na); b.y.reserve(na); b.z.reserve(na); b.atom_code.reserve(na);
                b.atom_off.reserve(nr + 1); b.res_code.reserve(nr); b.bfac_ca.reserve(nr);
            }
            for (size_t i = 0; i < job.frags.size(); i++) {
                // a fragment the codec cannot take is reported and left out (Batch::add throws before it changes the batch)
                Fragment& f = job.frags[i];
                try {
                    if (!f.prepared) b.add(f.atoms, f.title, o.brk);
                    else if (!f.prep_err.empty()) throw std::runtime_error(f.prep_err);
                    else b.append(f.atoms, f.prep, f.title);
                    kept.push_back(i);
                }
                catch (const std::exception& e) { fprintf(stderr, "[Error] compressing %s: %s\n", f.out_name.c_str(), e.what()); }
                table_pool().put(std::move(job.frags[i].atoms));   // its atoms are in the batch (or refused): the table goes round
            }
            fcz_chain_batch v = b.view(o.brk);
            std::vector<uint64_t> off(v.n_chains + 1, 0);
            if (v.n_chains) fcz_compress_sizes(&v, off.data());
            // a job claims its byte range of the database AFTER its GPU call, with the records that compressed packed back to
            // back (the reference's writer appends only what compressed); every job claims, also an empty or failed one
            if (kept.empty() || !ctx) { if (placed) seq.claim(job.index, 0); continue; }
            blob.resize(off.back());
            const int32_t UNSET = INT32_MIN;                              // a status the library never writes
            std::vector<int32_t> status(v.n_chains, UNSET);
            const auto t0 = clk::now();
            const int rc = fcz_compress_batch(ctx, &v, off.data(), blob.data(), status.data());
            gpu_busy[w] += std::chrono::duration<double>(clk::now() - t0).count();
            // rc is the worst per-chain status or a failure of the call itself; FCZ_E_INVALID_ARG is both (a refused chain, or
            // refused arguments -- then no per-chain status was written). Nothing may be emitted unless every chain has one.
            bool call_failed = rc != FCZ_OK && rc != FCZ_E_RESIDUE && rc != FCZ_E_TOO_SHORT && rc != FCZ_E_NONFINITE && rc != FCZ_E_INVALID_ARG;
            for (uint32_t q = 0; q < v.n_chains && !call_failed; q++) if (status[q] == UNSET) call_failed = true;
            if (call_failed) {
                fprintf(stderr, "[Error] %s: %zu chains not compressed\n", fcz_status_string(rc), kept.size());
                hard_fail = true; if (placed) seq.claim(job.index, 0); continue;
            }
            try {
                if (o.tar) {
                    // the records that compressed as members of the archive, named as a directory's files would be (src/main.cpp:519-523)
                    std::vector<size_t> okq;
                    for (size_t q = 0; q < kept.size(); q++) if (status[q] == FCZ_OK) okq.push_back(q);
                    std::vector<uint8_t> members;
                    const uint64_t bytes = tar_pack_members(members, okq.size(), [&](size_t k) { return base_name(job.frags[kept[okq[k]]].out_name); },
                                                            [&](size_t k) { return blob.data() + off[okq[k]]; }, [&](size_t k) { return off[okq[k] + 1] - off[okq[k]]; });
                    const uint64_t at = seq.claim(job.index, bytes);
                    if (bytes) pwrite_all(db_fd, members.data(), bytes, at);
                }
                uint64_t packed = 0;                                      // bytes of the records that compressed
                std::vector<uint64_t> lens; std::vector<std::string> names;
                for (size_t q = 0; q < kept.size(); q++) {
                    if (status[q] != FCZ_OK) continue;
                    const uint64_t len = off[q + 1] - off[q];
                    if (o.db && packed != off[q]) memmove(blob.data() + packed, blob.data() + off[q], len);
                    if (o.db) { lens.push_back(len); names.push_back(job.frags[kept[q]].db_name); }
                    packed += len;
                }
                const uint64_t at = o.db ? seq.claim(job.index, packed, &lens, &names) : 0;
                if (o.db && packed) pwrite_all(db_fd, blob.data(), packed, at);
                std::vector<FragOut> ev;                                  // directory / single-file output: every fragment of the job in input order
                if (!placed) for (size_t i = 0; i < job.frags.size(); i++) ev.push_back({job.frags[i].file, (uint32_t)i, job.frags[i].out_name, nullptr, 0});
                for (size_t q = 0; q < kept.size(); q++) {
                    const Fragment& f = job.frags[kept[q]];
                    if (status[q] != FCZ_OK) { fprintf(stderr, "[Error] compressing %s: %s\n", f.out_name.c_str(), fcz_status_string(status[q])); continue; }
                    n_frag_ok++; n_res += v.res_off[q + 1] - v.res_off[q]; n_bytes += off[q + 1] - off[q];
                    n_atoms += v.atom_off[v.res_off[q + 1]] - v.atom_off[v.res_off[q]];
                    if (placed) continue;
                    ev[kept[q]].p = blob.data() + off[q]; ev[kept[q]].len = off[q + 1] - off[q];
                }
                if (!placed) write_fragments_in_order(ev, [&](const FragOut& f) { return single ? output : output + "/" + f.name; }, o.overwrite);
            } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
        }
        if (ctx) fcz_ctx_destroy(ctx);
    });

    // ---- producer ----
    double t_parse = 0.0;
    uint64_t in_bytes = 0;
    {
        // files parsed side by side before the next job is cut, and fragments per job: small enough that every worker gets
        // several jobs (parse, staging, GPU and writes of different jobs overlap), large enough to fill a GPU launch
        // (bounded: n_workers + 2 jobs wait in the queue with their atom tables, ~100 KB per 350-residue chain)
        const size_t JOB = std::max<size_t>(256, std::min<size_t>(JOB_CHAINS_MAX, files.size() / (4 * (size_t)n_workers) + 1));
        const size_t FILE_CHUNK = std::max<size_t>(JOB, 512);
        std::vector<Fragment> pending;
        size_t job_index = 0;
        auto cut = [&](bool all) {
            while (pending.size() >= JOB || (all && !pending.empty())) {
                CompressJob j; j.index = job_index++;
                const size_t n = std::min(pending.size(), JOB);
                j.frags.assign(std::make_move_iterator(pending.begin()), std::make_move_iterator(pending.begin() + n));
                pending.erase(pending.begin(), pending.begin() + n);
                queue.put(std::move(j));
            }
        };
        for (size_t f0 = 0; f0 < files.size(); f0 += FILE_CHUNK) {
            const auto t0 = clk::now();
            const size_t f1 = std::min(files.size(), f0 + FILE_CHUNK);
            // (a single file into a database or a tar is named by the INPUT, src/main.cpp:447-456)
            fragments_of_files(files, f0, f1, single && !placed, output, !o.db, o, pending, true);
            t_parse += std::chrono::duration<double>(clk::now() - t0).count();
            in_bytes = g_bytes_read.load();
            cut(false);
        }
        cut(true);
        queue.close();
    }
    const double t_parsed = std::chrono::duration<double>(clk::now() - t_start).count();   // every job is in the queue
    for (std::thread& t : workers) t.join();
    const double t_joined = std::chrono::duration<double>(clk::now() - t_start).count();

    // ---- index: rows of all workers in job order, keys numbered over the records that made it ----
    if (o.db) {
        close(db_fd);
        seq.finish(output, !hard_fail);
        if (hard_fail) fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str());
    } else if (o.tar) {
        if (!hard_fail && !tar_finalize(db_fd, seq.pos)) hard_fail = true;
        close(db_fd);
        if (hard_fail) { unlink(output.c_str()); fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str()); }
    }
    if (o.json_stats) {
        const double wall = std::chrono::duration<double>(clk::now() - t_start).count();
        double busy = 0.0; for (double g : gpu_busy) busy += g;
        printf("{\"mode\": \"compress\", \"ingest\": \"host\", \"gpus\": %d, \"workers\": %d, \"host_threads\": %d, \"files\": %zu, \"input_bytes\": %llu, "
               "\"records\": %llu, \"residues\": %llu, \"atoms\": %llu, \"fcz_bytes\": %llu, \"wall_s\": %.4f, \"parse_s\": %.4f, "
               "\"codec_call_s_sum\": %.4f, \"ctx_ready_s\": %.4f, \"all_parsed_s\": %.4f, \"workers_done_s\": %.4f, \"residues_per_s\": %.1f, "
               "\"input_MB_per_s\": %.1f, \"pinned_blocks\": %llu}\n",
               gpus, n_workers, omp_get_max_threads(), files.size(), (unsigned long long)in_bytes, (unsigned long long)n_frag_ok.load(),
               (unsigned long long)n_res.load(), (unsigned long long)n_atoms.load(), (unsigned long long)n_bytes.load(), wall, t_parse, busy,
               *std::max_element(ctx_ready.begin(), ctx_ready.end()), t_parsed, t_joined,
               wall > 0 ? n_res.load() / wall : 0.0, wall > 0 ? in_bytes / wall / 1e6 : 0.0, (unsigned long long)pinned_blocks().load());
    }
    return hard_fail ? 1 : 0;
}


// ---- compress with the structure ingest on the device ---------------------------------------------------------------------
// The host-parse pipeline above spends its wall time in the parse threads (240 KB of text per 20 us of GPU work). Here the
// host threads only READ: the files of a job land back to back in a page-locked buffer, one DMA takes them to the GPU, and
// fcz_compress_pdb_begin / _fetch (parse -> fragments -> batch -> FCZ, all in HBM) return the records. What the device hands
// back (file_status FCZ_INGEST_HOST_*: a field outside the fixed-column layout) and what it does not read (mmCIF, gzip) goes
// through the host parser of this file and fcz_compress_batch; the records of a job are merged in file order either way.
struct TextJob {
    size_t index = 0;
    std::vector<std::string> paths;            // the job's files, in input order
    std::vector<int> slot;                     // position among the text files of the job, -1: a host-parsed file
    pvec<uint8_t>* text = nullptr;             // page-locked: the text files back to back
    std::vector<uint64_t> file_off{0};
    std::string names; std::vector<uint32_t> name_off{0}, stem_len;
    std::vector<uint8_t> is_gz;                // per text slot: the bytes are a gzip member the device inflates (fcz_compress_gz_begin)
    bool any_gz = false;
    std::vector<std::vector<Fragment>> host_frags;   // per file of the job: fragments of the host-parsed ones
};

struct TextPool {                              // the page-locked text buffers go round (pinning memory is expensive)
    std::mutex m; std::condition_variable cv; std::vector<pvec<uint8_t>*> free_;
    pvec<uint8_t>* get() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !free_.empty(); }); pvec<uint8_t>* b = free_.back(); free_.pop_back(); return b; }
    void put(pvec<uint8_t>* b) { { std::lock_guard<std::mutex> l(m); free_.push_back(b); } cv.notify_one(); }
};

// structure text the device ingest reads: PDB records (k_ingest_parse) and mmCIF (k_ingest_parse_cif: the files k_ingest_parse leaves
// because they open with data_); what either kernel cannot promise to read as the reference's reader would comes back for the host reader
bool is_plain_pdb(const std::string& path) { return ends_with(path, ".pdb") || ends_with(path, ".ent") || ends_with(path, ".cif"); }
bool is_gz_pdb(const std::string& path) { return ends_with(path, ".pdb.gz") || ends_with(path, ".ent.gz") || ends_with(path, ".cif.gz"); }
// PDB text the device ingest takes: plain files are read straight into the page-locked buffer, gzipped ones are inflated by
// the reader threads first (their parse still happens on the device)
[[maybe_unused]] bool is_device_text(const std::string& path) { return is_plain_pdb(path) || is_gz_pdb(path); }

int run_compress_device(const Options& o, InputPlan& plan, const std::string& output) {
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    const int n_dev = fcz_device_count();
    const int gpus = o.gpus <= 0 ? n_dev - o.device : o.gpus;
    const int n_workers = gpus * std::max(1, o.workers_per_gpu);
    pinned_enabled() = true;
    int db_fd = -1;
    const bool placed = o.db || o.tar;          // one output file: a job's records are placed in it by the Sequencer
    if (placed) {
        db_fd = open(output.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
        if (db_fd < 0) { fprintf(stderr, "[Error] cannot write %s\n", output.c_str()); return 1; }
    } else make_dir(output);

    JobQueue<TextJob> queue((size_t)n_workers + 1);
    Sequencer seq;
    if (o.db && !seq.open_index(output)) { fprintf(stderr, "[Error] cannot write %s.index\n", output.c_str()); return 1; }
    std::atomic<bool> hard_fail{false};
    std::atomic<uint64_t> n_res{0}, n_frag_ok{0}, n_bytes{0}, n_atoms{0}, n_host_files{0}, n_host_gz{0}, n_dev_gz{0};
    std::vector<double> gpu_busy(n_workers, 0.0), ctx_ready(n_workers, 0.0);
    std::vector<std::unique_ptr<pvec<uint8_t>>> text_bufs;
    TextPool pool;
    for (int i = 0; i < n_workers + 2; i++) { text_bufs.emplace_back(new pvec<uint8_t>()); pool.put(text_bufs.back().get()); }

    std::vector<std::thread> workers;
    for (int w = 0; w < n_workers; w++) workers.emplace_back([&, w]() {
        fcz_ctx* ctx = nullptr;
        if (fcz_ctx_create(o.device + w % gpus, &ctx) != FCZ_OK) { hard_fail = true; fprintf(stderr, "[Error] no ctx on device %d\n", o.device + w % gpus); }
        ctx_ready[w] = std::chrono::duration<double>(clk::now() - t_start).count();
        pvec<uint8_t> blob, blob_host, packed;
        Batch hb;
        TextJob job;
        while (queue.get(job)) {
            struct Rec { size_t file; uint32_t sub; const uint8_t* p; uint64_t len; std::string out_name, db_name; uint32_t nres, natoms; };
            std::vector<Rec> recs;
            bool failed = !ctx;
            const uint32_t n_text = (uint32_t)job.file_off.size() - 1;
            std::vector<uint64_t> off; std::vector<int32_t> status, file_status; std::vector<uint32_t> chain_file, chain_meta, refused, counts(5, 0);
            std::vector<size_t> text_file(n_text);                      // text slot -> file of the job
            for (size_t i = 0; i < job.paths.size(); i++) if (job.slot[i] >= 0) text_file[(size_t)job.slot[i]] = i;
            auto stem_of = [&](size_t i) { std::string stem, ext; file_parts(base_name(job.paths[i]), stem, ext); return stem; };
            // the suffix of a fragment's file (src/main.cpp:498-502): ".fcz" for what isCompressible knows, else the input's own
            auto suffix_of = [&](size_t i) { std::string stem, ext; file_parts(base_name(job.paths[i]), stem, ext); return is_compressible(stem, ext) ? std::string(".fcz") : "." + ext; };
            if (!failed && n_text) {
                uint64_t fcz_bytes = 0;
                const auto t0 = clk::now();
                // (gzip members among the job's files are inflated on the device in front of the parser: a fifth of the text's bytes cross the link)
                int rc = job.any_gz
                    ? fcz_compress_gz_begin(ctx, job.text->data(), job.file_off.data(), n_text, job.is_gz.data(), job.names.data(), job.name_off.data(), job.stem_len.data(),
                                            o.brk, o.skip_discontinuous ? FCZ_INGEST_SKIP_DISCONTINUOUS : 0, counts.data(), &fcz_bytes)
                    : fcz_compress_pdb_begin(ctx, job.text->data(), job.file_off.data(), n_text, job.names.data(), job.name_off.data(), job.stem_len.data(),
                                             o.brk, o.skip_discontinuous ? FCZ_INGEST_SKIP_DISCONTINUOUS : 0, counts.data(), &fcz_bytes);
                if (rc == FCZ_OK) {
                    const uint32_t C = counts[0];
                    off.assign((size_t)C + 1, 0); status.assign(C, 0); chain_file.assign(C, 0); chain_meta.assign(C, 0);
                    file_status.assign(n_text, 0); refused.assign(2 * (size_t)counts[4], 0);
                    blob.resize(fcz_bytes);
                    rc = fcz_compress_pdb_fetch(ctx, off.data(), status.data(), chain_file.data(), chain_meta.data(), file_status.data(), refused.data(), blob.data());
                }
                gpu_busy[w] += std::chrono::duration<double>(clk::now() - t0).count();
                if (rc != FCZ_OK) { fprintf(stderr, "[Error] %s: job of %u files not compressed\n", fcz_status_string(rc), n_text); failed = true; }
            }
            if (!failed && n_text) {
                // the chains' names (mmCIF: up to four characters; chain_meta keeps the first)
                std::vector<uint32_t> name4(counts[0], 0);
                if (counts[0] && fcz_ingest_chain_names_fetch(ctx, name4.data()) != FCZ_OK) std::fill(name4.begin(), name4.end(), 0u);
                auto frag_name = [&](size_t file, uint32_t meta, uint32_t n4 = 0) {
                    std::string nm = stem_of(file);
                    // (a blank chain id names nothing: gemmi's read_string trims it to "", AtomTable::chain_name likewise)
                    if ((meta & FCZ_INGEST_MULTI_CHAIN) && (char)(meta & 0xffu) != ' ') {
                        if (n4 == 0) n4 = meta & 0xffu;
                        for (int b = 0; b < 4 && ((n4 >> (8 * b)) & 0xffu); b++) nm.push_back((char)((n4 >> (8 * b)) & 0xffu));
                    }
                    if (meta & FCZ_INGEST_MULTI_FRAG) nm += "_" + std::to_string((meta >> 8) & 0xffu);
                    return nm;
                };
                // A file whose fragments do not all have names of their own (chain A, chain B, chain A again) written into a DIRECTORY: the
                // reference's lambda stops at the second use of a name (write_fragments_in_order). The device reports a file's fragments
                // without their order among the refused ones, so such a file -- rare -- goes through the host reader, whose fragments are
                // in order; nothing of it is taken from the device's batch.
                std::vector<char> collide(job.paths.size(), 0);
                if (!placed) {
                    std::vector<std::pair<size_t, std::string>> nm_of;
                    for (uint32_t c = 0; c < counts[0]; c++) nm_of.push_back({text_file[chain_file[c]], frag_name(text_file[chain_file[c]], chain_meta[c], name4[c])});
                    for (size_t k = 0; k + 1 < refused.size(); k += 2) nm_of.push_back({text_file[refused[k]], frag_name(text_file[refused[k]], refused[k + 1])});
                    std::sort(nm_of.begin(), nm_of.end());
                    for (size_t k = 1; k < nm_of.size(); k++) if (nm_of[k] == nm_of[k - 1]) collide[nm_of[k].first] = 1;
                }
                for (uint32_t c = 0; c < counts[0]; c++) {
                    const size_t file = text_file[chain_file[c]];
                    if (collide[file]) continue;
                    const std::string nm = frag_name(file, chain_meta[c], name4[c]);
                    if (status[c] != FCZ_OK) { fprintf(stderr, "[Error] compressing %s.fcz: %s\n", nm.c_str(), fcz_status_string(status[c])); continue; }
                    recs.push_back({file, (chain_meta[c] >> 8) & 0xffu, blob.data() + off[c], off[c + 1] - off[c], nm + suffix_of(file), stem_of(file), 0, 0});
                    // (sub: the order of a file's records is the order the device emitted them; see the stable sort below)
                    recs.back().sub = c;
                }
                for (size_t k = 0; k + 1 < refused.size(); k += 2) {
                    static const char* why[] = {"", "residue name is not supported by the codec", "residue without N, CA, C backbone atoms in order",
                                                "chain does not fit the FCZ header (65535 residues, 255 anchors)", "discontinuous chain skipped",
                                                "residue with a second N, CA or C atom", "the chain's last atom carries another residue name than its residue"};
                    const uint32_t reason = refused[k + 1] >> 24;
                    if (collide[text_file[refused[k]]]) continue;
                    fprintf(stderr, "[Error] compressing %s.fcz: %s\n", frag_name(text_file[refused[k]], refused[k + 1]).c_str(), why[reason < 7 ? reason : 0]);
                }
                // what the device handed back: parsed here from the text that is already in memory -- on the host threads, not one file
                // after the other on this worker's (a twentieth of an archive-style mmCIF directory comes back: round 6)
                std::vector<uint32_t> back;
                for (uint32_t t = 0; t < n_text; t++) {
                    if (file_status[t] == FCZ_INGEST_NO_ATOMS) { fprintf(stderr, "[Error] No atoms found in the input file: %s\n", base_name(job.paths[text_file[t]]).c_str()); continue; }
                    if (file_status[t] != FCZ_OK || collide[text_file[t]]) back.push_back(t);
                }
#pragma omp parallel for schedule(dynamic, 1) if (back.size() > 1)
                for (long long bi = 0; bi < (long long)back.size(); bi++) {
                    const uint32_t t = back[(size_t)bi];
                    if (file_status[t] != FCZ_INGEST_HOST_GZIP) n_host_files++;      // (a member zlib has to take is counted on its own below)
                    const size_t file = text_file[t];
                    std::string stem, ext; file_parts(base_name(job.paths[file]), stem, ext);
                    // (a member the device did not inflate, FCZ_INGEST_HOST_GZIP, or one whose text it handed back: zlib here, then the reader)
                    if (file_status[t] == FCZ_INGEST_HOST_GZIP) n_host_gz++;
                    // (the job's buffer holds what was READ: a gzip member the device inflated elsewhere is still a gzip member here)
                    try { fragments_from_memory((const char*)job.text->data() + job.file_off[t], job.file_off[t + 1] - job.file_off[t], base_name(job.paths[file]), stem, ext, !o.db, o, job.host_frags[file], /*inflated=*/!(job.any_gz && job.is_gz[t])); }
                    catch (const std::exception& e) { fprintf(stderr, "[Error] %s: %s\n", base_name(job.paths[file]).c_str(), e.what()); }
                }
            }
            // host-parsed fragments (mmCIF, gzip, and what the device handed back): one fcz_compress_batch for the job
            std::vector<uint64_t> hoff; std::vector<int32_t> hstatus;
            std::vector<FragOut> host_refused;                        // fragments of host-parsed files this host refuses: their place in the file's order of names
            if (!failed) {
                hb.clear();
                std::vector<std::pair<size_t, size_t>> kept;          // (file, fragment)
                for (size_t i = 0; i < job.host_frags.size(); i++) for (size_t j = 0; j < job.host_frags[i].size(); j++) {
                    Fragment& f = job.host_frags[i][j];
                    try { hb.add(f.atoms, f.title, o.brk); kept.push_back({i, j}); }
                    catch (const std::exception& e) { fprintf(stderr, "[Error] compressing %s: %s\n", f.out_name.c_str(), e.what()); host_refused.push_back({i, (uint32_t)j, f.out_name, nullptr, 0}); }
                }
                if (!kept.empty()) {
                    fcz_chain_batch v = hb.view(o.brk);
                    hoff.assign(v.n_chains + 1, 0);
                    fcz_compress_sizes(&v, hoff.data());
                    blob_host.resize(hoff.back());
                    const int32_t UNSET = INT32_MIN;
                    hstatus.assign(v.n_chains, UNSET);
                    const auto t0 = clk::now();
                    const int rc = fcz_compress_batch(ctx, &v, hoff.data(), blob_host.data(), hstatus.data());
                    gpu_busy[w] += std::chrono::duration<double>(clk::now() - t0).count();
                    bool call_failed = rc != FCZ_OK && rc != FCZ_E_RESIDUE && rc != FCZ_E_TOO_SHORT && rc != FCZ_E_NONFINITE && rc != FCZ_E_INVALID_ARG;
                    for (uint32_t q = 0; q < v.n_chains && !call_failed; q++) if (hstatus[q] == UNSET) call_failed = true;
                    if (call_failed) { fprintf(stderr, "[Error] %s: %zu chains not compressed\n", fcz_status_string(rc), kept.size()); failed = true; }
                    for (size_t q = 0; q < kept.size() && !failed; q++) {
                        const Fragment& f = job.host_frags[kept[q].first][kept[q].second];
                        if (hstatus[q] != FCZ_OK) { fprintf(stderr, "[Error] compressing %s: %s\n", f.out_name.c_str(), fcz_status_string(hstatus[q])); host_refused.push_back({kept[q].first, (uint32_t)kept[q].second, f.out_name, nullptr, 0}); continue; }
                        recs.push_back({kept[q].first, (uint32_t)kept[q].second, blob_host.data() + hoff[q], hoff[q + 1] - hoff[q], f.out_name, f.db_name, 0, 0});
                    }
                }
            }
            if (failed) { hard_fail = true; if (placed) seq.claim(job.index, 0); pool.put(job.text); continue; }
            try {
                // records of the job in file order (a file's own records keep the order they were emitted in)
                std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.file != b.file ? a.file < b.file : a.sub < b.sub; });
                uint64_t total = 0; for (const Rec& r : recs) total += r.len;
                if (o.db) {
                    packed.resize(total);
                    uint64_t pos = 0;
                    for (const Rec& r : recs) { memcpy(packed.data() + pos, r.p, r.len); pos += r.len; }
                    std::vector<uint64_t> lens(recs.size()); std::vector<std::string> names(recs.size());
                    for (size_t q = 0; q < recs.size(); q++) { lens[q] = recs[q].len; names[q] = recs[q].db_name; }
                    const uint64_t at = seq.claim(job.index, total, &lens, &names);
                    if (total) pwrite_all(db_fd, packed.data(), total, at);
                } else if (o.tar) {
                    // members named as the files of a directory output would be (writeTar(tar_out, baseName(filename), ...), src/main.cpp:519-523)
                    const uint64_t bytes = tar_pack_members(packed, recs.size(), [&](size_t q) { return base_name(recs[q].out_name); }, [&](size_t q) { return recs[q].p; }, [&](size_t q) { return recs[q].len; });
                    const uint64_t at = seq.claim(job.index, bytes);
                    if (bytes) pwrite_all(db_fd, packed.data(), bytes, at);
                } else {
                    std::vector<FragOut> ev = host_refused;
                    for (const Rec& r : recs) ev.push_back({r.file, r.sub, r.out_name, r.p, r.len});
                    write_fragments_in_order(ev, [&](const FragOut& f) { return output + "/" + f.name; }, o.overwrite);
                }
                for (const Rec& r : recs) {
                    n_frag_ok++; n_bytes += r.len;
                    n_res += (uint32_t)r.p[4] | ((uint32_t)r.p[5] << 8); n_atoms += (uint32_t)r.p[6] | ((uint32_t)r.p[7] << 8);   // header.nResidue, nAtom
                }
            } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
            pool.put(job.text);
        }
        if (ctx) fcz_ctx_destroy(ctx);
    });

    // ---- producer: size, then read, the inputs of a job side by side on the host threads. Inputs are files or database entries
    //      (InputPlan: this process's range of the listing); a database entry is a file image under its lookup name
    //      (src/input_processor.h:237-257 hands `name, data, length` to the same lambda as a directory's files) ----
    double t_read = 0.0; uint64_t in_bytes = 0;
    try {
        // files per job: enough wavefronts for the wavefront-per-file kernels (k_inflate, k_ingest_parse*: 256 CUs x 8-11 resident), few
        // enough for several jobs in flight per worker. --job-files N overrides (measurement)
        const size_t JOB = o.job_files > 0 ? (size_t)o.job_files : plan.unsized ? (size_t)2048
                         : std::max<size_t>(64, std::min<size_t>(2048, (size_t)plan.n_mine / (4 * (size_t)n_workers) + 1));
        size_t job_index = 0;
        std::vector<InputItem> cur;
        plan.hold = true;                       // (a gzipped tar's members stay in memory until their job has copied them: release() below)
        auto make_job = [&]() {
            if (cur.empty()) return;
            const auto t0 = clk::now();
            const size_t nf = cur.size();
            TextJob j; j.index = job_index++;
            j.paths.resize(nf);
            for (size_t i = 0; i < nf; i++) j.paths[i] = cur[i].name;
            j.slot.assign(nf, -1); j.host_frags.resize(nf);
            std::vector<uint64_t> size(nf, 0);
            std::vector<int> cls(nf, 0);                              // 0: the host reader takes it, 1: text read straight into the job's buffer, 2: text that waits in unz[], 3: a gzip member read straight into the buffer (inflated on the device)
            std::vector<std::string> unz(nf);                         // inflated text of gzipped inputs; entries the host reader takes
            std::vector<std::string> err(nf);
#pragma omp parallel for schedule(dynamic, 16)
            for (long long i = 0; i < (long long)nf; i++) {
                const InputItem& it = cur[(size_t)i];
                const std::string& nm = j.paths[(size_t)i];
                if (it.kind == 1) {
                    // a database entry: gzipped by its NAME, PDB or mmCIF by its CONTENT (StructureReader::loadFromBuffer)
                    try {
                        // (before anything is sized from the index's numbers: a corrupt line is this entry's error, not a bad_alloc)
                        if (!plan.entry_in_range(it)) throw std::runtime_error("database entry out of range");
                        if (ends_with(nm, ".gz") && !o.host_inflate && it.len >= 18) {
                            // the member goes to the device as it is; PDB / mmCIF by the first bytes of its text (inflated here: microseconds)
                            uint8_t zh[4096]; char head[4096];
                            InputItem pre = it; pre.len = std::min<uint64_t>(it.len, sizeof zh);
                            if (!plan.read_entry(pre, zh)) throw std::runtime_error("database entry out of range");
                            InputItem tail = it; tail.off = it.off + it.len - 4; tail.len = 4; uint8_t t4[4];
                            if (!plan.read_entry(tail, t4)) throw std::runtime_error("database entry out of range");
                            const uint64_t isize = (uint64_t)t4[0] | ((uint64_t)t4[1] << 8) | ((uint64_t)t4[2] << 16) | ((uint64_t)t4[3] << 24);
                            const long got = gunzip_head(zh, (size_t)pre.len, head, sizeof head);
                            const int fz = got < 0 ? 0 : coor_format_from_prefix(head, (size_t)got, (size_t)std::max<uint64_t>(isize, (uint64_t)got));
                            if (fz == 1 || fz == 2) { cls[i] = 3; size[i] = it.len; continue; }
                        }
                        if (ends_with(nm, ".gz")) {
                            std::string z(it.len, '\0');
                            if (!plan.read_entry(it, (uint8_t*)&z[0])) throw std::runtime_error("database entry out of range");
                            g_bytes_read += z.size();
                            unz[i] = gunzip(z);
                            { const int fz = coor_format_from_content(unz[i].data(), unz[i].size()); if (fz == 1 || fz == 2) { cls[i] = 2; size[i] = unz[i].size(); } }
                            continue;
                        }
                        char head[4096];
                        InputItem pre = it; pre.len = std::min<uint64_t>(it.len, sizeof head);
                        if (!plan.read_entry(pre, (uint8_t*)head)) throw std::runtime_error("database entry out of range");
                        int fmt = coor_format_from_prefix(head, (size_t)pre.len, (size_t)it.len);
                        if (fmt == 1 || fmt == 2) {
                            // MMseqs-made databases end an entry with a NUL after the text's last line end: not part of the text
                            // (the reader ignores a line of one NUL; the device parser would hand the file back for it)
                            size[i] = it.len; cls[i] = 1;
                            if (it.len >= 2) {
                                InputItem tail = it; tail.off = it.off + it.len - 2; tail.len = 2; char t2[2];
                                if (plan.read_entry(tail, (uint8_t*)t2) && t2[1] == '\0' && t2[0] == '\n') size[i] = it.len - 1;
                            }
                            continue;
                        }
                        unz[i].resize(it.len);
                        if (it.len && !plan.read_entry(it, (uint8_t*)&unz[i][0])) throw std::runtime_error("database entry out of range");
                        g_bytes_read += it.len;
                        if (fmt < 0) { const int fz = coor_format_from_content(unz[i].data(), unz[i].size()); if (fz == 1 || fz == 2) { cls[i] = 2; size[i] = unz[i].size(); } }
                    } catch (const std::exception& e) { err[i] = "[Error] " + base_name(nm) + ": " + e.what() + "\n"; size[i] = UINT64_MAX - 1; cls[i] = 2; }
                    continue;
                }
                if (is_gz_pdb(nm) && !o.host_inflate) {
                    // the gzip member is read straight into the job's buffer and inflated on the device
                    cls[i] = 3;
                    struct stat st;
                    if (stat(nm.c_str(), &st) == 0 && S_ISREG(st.st_mode)) size[i] = (uint64_t)st.st_size; else size[i] = UINT64_MAX;
                    continue;
                }
                if (is_gz_pdb(nm)) {
                    cls[i] = 2;
                    try { const std::string z = read_file(nm); g_bytes_read += z.size(); unz[i] = gunzip(z); size[i] = unz[i].size(); }
                    catch (const std::exception& e) { err[i] = "[Error] " + base_name(nm) + ": " + e.what() + "\n"; size[i] = UINT64_MAX - 1; }
                    continue;
                }
                if (!is_plain_pdb(nm)) continue;
                cls[i] = 1;
                struct stat st;
                if (stat(nm.c_str(), &st) == 0 && S_ISREG(st.st_mode)) size[i] = (uint64_t)st.st_size; else size[i] = UINT64_MAX;
            }
            uint32_t n_text = 0;
            for (size_t i = 0; i < nf; i++) {
                if (cls[i] == 0) continue;
                if (size[i] == UINT64_MAX - 1) continue;               // a gzip stream that does not inflate, an entry outside its file (reported below)
                if (size[i] == UINT64_MAX) { fprintf(stderr, "[Error] cannot open %s\n", j.paths[i].c_str()); continue; }
                j.slot[i] = (int)n_text++;
                j.is_gz.push_back(cls[i] == 3 ? 1 : 0);
                if (cls[i] == 3) { j.any_gz = true; n_dev_gz++; }
                j.file_off.push_back(j.file_off.back() + size[i]);
                const std::string base = base_name(j.paths[i]);
                std::string stem, ext; file_parts(base, stem, ext);
                j.names += base; j.name_off.push_back((uint32_t)j.names.size()); j.stem_len.push_back((uint32_t)stem.size());
            }
            j.text = pool.get();
            if (j.text->size() < j.file_off.back() + 64) j.text->resize(j.file_off.back() + j.file_off.back() / 8 + 64);   // grows, never shrinks: a resize touches (zero-fills) what it adds
#pragma omp parallel for schedule(dynamic, 8)
            for (long long i = 0; i < (long long)nf; i++) {
                const InputItem& it = cur[(size_t)i];
                if (j.slot[i] >= 0 && cls[i] == 2) {
                    memcpy(j.text->data() + j.file_off[(size_t)j.slot[i]], unz[i].data(), unz[i].size());
                    std::string().swap(unz[i]);
                } else if (j.slot[i] >= 0 && it.kind == 1) {
                    InputItem body = it; body.len = size[i];
                    if (!plan.read_entry(body, j.text->data() + j.file_off[(size_t)j.slot[i]])) memset(j.text->data() + j.file_off[(size_t)j.slot[i]], ' ', size[i]);
                    g_bytes_read += size[i];
                } else if (j.slot[i] >= 0) {
                    // straight into the page-locked buffer; a file that shrank since stat() leaves spaces (an empty line), one that
                    // grew is cut at its stat size
                    const uint64_t at = j.file_off[(size_t)j.slot[i]], want = j.file_off[(size_t)j.slot[i] + 1] - at;
                    uint64_t got = 0;
                    const int fd = open(j.paths[i].c_str(), O_RDONLY);
                    if (fd >= 0) {
                        while (got < want) { const ssize_t k = read(fd, j.text->data() + at + got, want - got); if (k <= 0) break; got += (uint64_t)k; }
                        close(fd);
                    }
                    if (got < want) memset(j.text->data() + at + got, ' ', want - got);
                    g_bytes_read += got;
                } else if (cls[i] == 0) {
                    std::string stem, ext; file_parts(base_name(j.paths[i]), stem, ext);
                    try {
                        if (it.kind == 1) fragments_from_memory(unz[i].data(), unz[i].size(), base_name(j.paths[i]), stem, ext, !o.db, o, j.host_frags[i], /*inflated=*/ends_with(j.paths[i], ".gz"));
                        else fragments_of(j.paths[i], stem, ext, !o.db, o, j.host_frags[i]);
                    }
                    catch (const std::exception& e) { err[i] = "[Error] " + base_name(j.paths[i]) + ": " + e.what() + "\n"; }
                    std::string().swap(unz[i]);
                }
            }
            for (const std::string& e : err) if (!e.empty()) fputs(e.c_str(), stderr);
            t_read += std::chrono::duration<double>(clk::now() - t0).count();
            in_bytes = g_bytes_read.load();
            queue.put(std::move(j));
            cur.clear();
            plan.release();
        };
        plan.for_each([&](const InputItem& it) { cur.push_back(it); if (cur.size() >= JOB) make_job(); });
        make_job();
    } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
    queue.close();
    const double t_queued = std::chrono::duration<double>(clk::now() - t_start).count();
    for (std::thread& t : workers) t.join();
    const double t_joined = std::chrono::duration<double>(clk::now() - t_start).count();
    if (o.db) {
        close(db_fd);
        seq.finish(output, !hard_fail);
        if (hard_fail) fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str());
    } else if (o.tar) {
        if (!hard_fail && !tar_finalize(db_fd, seq.pos)) hard_fail = true;
        close(db_fd);
        if (hard_fail) { unlink(output.c_str()); fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str()); }
    }
    if (o.json_stats) {
        const double wall = std::chrono::duration<double>(clk::now() - t_start).count();
        double busy = 0.0; for (double g : gpu_busy) busy += g;
        const double ready = *std::max_element(ctx_ready.begin(), ctx_ready.end());
        printf("{\"mode\": \"compress\", \"ingest\": \"device\", \"gpus\": %d, \"workers\": %d, \"host_threads\": %d, \"files\": %zu, \"input_bytes\": %llu, "
               "\"records\": %llu, \"residues\": %llu, \"atoms\": %llu, \"fcz_bytes\": %llu, \"host_parsed_files\": %llu, \"device_inflated_files\": %llu, \"host_inflated_after_device_refusal\": %llu, \"wall_s\": %.4f, \"parse_s\": %.4f, "
               "\"codec_call_s_sum\": %.4f, \"ctx_ready_s\": %.4f, \"all_parsed_s\": %.4f, \"workers_done_s\": %.4f, \"residues_per_s\": %.1f, "
               "\"input_MB_per_s\": %.1f, \"pinned_blocks\": %llu, \"shard\": \"%d/%d\", \"items_total\": %llu, \"data_bytes\": %llu, \"streamed_inputs\": %s, \"max_rss_kb\": %ld}\n",
               gpus, n_workers, omp_get_max_threads(), (size_t)plan.n_mine, (unsigned long long)in_bytes, (unsigned long long)n_frag_ok.load(),
               (unsigned long long)n_res.load(), (unsigned long long)n_atoms.load(), (unsigned long long)n_bytes.load(), (unsigned long long)n_host_files.load(),
               (unsigned long long)(n_dev_gz.load() - n_host_gz.load()), (unsigned long long)n_host_gz.load(), wall, t_read, busy,
               ready, t_queued, t_joined, wall > 0 ? n_res.load() / wall : 0.0, wall > 0 ? in_bytes / wall / 1e6 : 0.0, (unsigned long long)pinned_blocks().load(),
               o.shard_rank, o.shard_world, (unsigned long long)plan.n_items, (unsigned long long)seq.pos, plan.streamed_all ? "true" : "false", max_rss_kb());
    }
    return hard_fail ? 1 : 0;
}

// ---- FCZ inputs ----
// page-locked like pvec, but resize() leaves what it adds as it is: the gather below overwrites every byte it asked for (a value-
// initialising resize is a second pass over ~100 MB per batch)
template <class T> struct RawPinnedAlloc : PinnedAlloc<T> {
    using value_type = T;
    template <class U> struct rebind { using other = RawPinnedAlloc<U>; };
    RawPinnedAlloc() = default;
    template <class U> RawPinnedAlloc(const RawPinnedAlloc<U>&) {}
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using rvec = std::vector<T, RawPinnedAlloc<T>>;

template <class Blob> struct EntriesT {
    std::vector<std::string> names;
    Blob blob;
    std::vector<uint64_t> off{0};
    void add(const std::string& name, const std::string& data) {
        names.push_back(name); blob.insert(blob.end(), data.begin(), data.end()); off.push_back(blob.size());
    }
    uint32_t n() const { return (uint32_t)names.size(); }
    void clear() { names.clear(); blob.clear(); off.assign(1, 0); }
};
using Entries = EntriesT<std::vector<uint8_t>>;
using RawEntries = EntriesT<rvec<uint8_t>>;        // extract: the batch goes to the device by DMA from where the gather put it

// every FCZ entry of the inputs (files, directories, databases -- optionally only the ids of --id-list), in batches. The entries of a
// database and the files of a directory are gathered a batch at a time on all host threads (the reference: `omp parallel for` over
// the entries, src/input_processor.h:85-101, :237-257); names, order and messages are those of the one-by-one walk.
template <class E, class F>
void for_each_entry(const Options& o, E& ents, F&& flush, size_t batch = BATCH_CHAINS) {
    for (const std::string& input : o.inputs) {
        if (!is_dir(input) && is_tar_name(input)) {
            // the members of a tar archive, plain or gzipped, in archive order (TarProcessor, src/input_processor.h:109-198)
            TarStream ts;
            if (!ts.open_path(input)) { fprintf(stderr, "[Error] open tar %s failed.\n", input.c_str()); continue; }
            TarMember m;
            while (ts.next(m) == 1) {
                std::string d((size_t)m.len, '\0');
                if ((m.len && !ts.read_exact(&d[0], m.len)) || !ts.skip_pad(m.len)) { fprintf(stderr, "[Error] cannot read entry %s\n", m.name.c_str()); break; }
                ents.add(m.name, d);
                if (ents.n() >= batch) flush();
            }
        } else if (is_db(input)) {
            DbReader r(input);
            std::vector<size_t> ids; std::vector<std::string> id_names;   // (an entry of an id list goes by the list's own line, src/input_processor.h:262-278)
            if (!o.id_list.empty()) {
                std::ifstream f(o.id_list);
                if (!f) fprintf(stderr, "[Error] user id '%s' does not exist.\n", o.id_list.c_str());
                std::string line;
                while (f && std::getline(f, line)) {
                    line = strip(line);
                    if (line.empty()) continue;
                    const long long id = o.id_mode == 0 ? r.id_of_key(atoll(line.c_str())) : r.id_of_name(line);
                    if (id < 0) { fprintf(stderr, "[Warning] %s not found in database.\n", line.c_str()); continue; }
                    ids.push_back((size_t)id); id_names.push_back(line);
                }
            } else { ids.resize(r.n()); for (size_t i = 0; i < r.n(); i++) ids[i] = i; }
            std::vector<size_t> pick;
            for (size_t q = 0; q < ids.size();) {
                const size_t q1 = std::min(ids.size(), q + (batch - std::min<size_t>(batch - 1, ents.n())));
                // in order: the entries that lie inside the data file get their place in the batch ...
                const size_t n0 = ents.n();
                uint64_t at = ents.blob.size();
                pick.clear();
                for (size_t k = q; k < q1; k++) {
                    const long long eo = r.rows[ids[k]].off, el = r.rows[ids[k]].len;
                    if (eo < 0 || el < 0 || (size_t)eo > r.size || (size_t)el > r.size - (size_t)eo) { fprintf(stderr, "[Error] database entry out of range\n"); continue; }
                    pick.push_back(k); at += (uint64_t)el; ents.off.push_back(at);
                }
                ents.blob.resize(at); ents.names.resize(n0 + pick.size());
                // ... and every thread copies its share of them out of the page cache
#pragma omp parallel for schedule(static)
                for (long long j = 0; j < (long long)pick.size(); j++) {
                    const size_t k = pick[(size_t)j];
                    const DbReader::Row& row = r.rows[ids[k]];
                    memcpy(ents.blob.data() + ents.off[n0 + (size_t)j], r.data + row.off, (size_t)row.len);
                    ents.names[n0 + (size_t)j] = id_names.empty() ? r.name(ids[k]) : id_names[k];
                }
                q = q1;
                if (ents.n() >= batch) flush();
            }
        } else {
            std::vector<std::string> files;
            if (is_dir(input)) list_files(input, o.recursive, files); else files.push_back(input);
            std::vector<std::string> bytes, err; std::vector<size_t> took;
            for (size_t q = 0; q < files.size();) {
                const size_t q1 = std::min(files.size(), q + (batch - std::min<size_t>(batch - 1, ents.n())));
                bytes.assign(q1 - q, std::string()); err.assign(q1 - q, std::string());
#pragma omp parallel for schedule(dynamic, 16) if (q1 - q > 32)
                for (long long k = (long long)q; k < (long long)q1; k++) {
                    try { bytes[(size_t)k - q] = read_file(files[(size_t)k]); } catch (const std::exception& e) { err[(size_t)k - q] = std::string("[Error] ") + e.what() + "\n"; }
                }
                const size_t n0 = ents.n();
                uint64_t at = ents.blob.size();
                took.clear();
                for (size_t k = q; k < q1; k++) {
                    if (!err[k - q].empty()) { fputs(err[k - q].c_str(), stderr); continue; }
                    took.push_back(k); at += bytes[k - q].size(); ents.off.push_back(at); ents.names.push_back(files[k]);
                }
                ents.blob.resize(at);
#pragma omp parallel for schedule(static) if (took.size() > 32)
                for (long long j = 0; j < (long long)took.size(); j++) {
                    const std::string& d = bytes[took[(size_t)j] - q];
                    if (!d.empty()) memcpy(ents.blob.data() + ents.off[n0 + (size_t)j], d.data(), d.size());
                }
                q = q1;
                if (ents.n() >= batch) flush();
            }
        }
    }
    flush();
}

// decompress: the same pipeline as compress (producer -> job queue -> gpus x workers_per_gpu workers with their own ctx ->
// sequenced pwrite / per-file writes -> merged index). The text of a job is formatted on the device (fcz_decompress_pdb_begin /
// _fetch); its size is known after `begin`, which is when the job claims its byte range of the output database.
struct DecompressJob { size_t index = 0; Entries ents; };

int run_decompress(const Options& o) {
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    const bool single = o.single;
    const std::string output = o.output;
    const int n_dev = fcz_device_count();
    if (n_dev <= 0) { fprintf(stderr, "[Error] %s\n", fcz_status_string(FCZ_E_NO_DEVICE)); return 1; }
    const int gpus = o.gpus <= 0 ? n_dev - o.device : o.gpus;
    if (gpus < 1 || o.device + gpus > n_dev) { fprintf(stderr, "[Error] --gpus %d from device %d but only %d device(s) are visible\n", gpus, o.device, n_dev); return 1; }
    const int n_workers = single ? 1 : gpus * std::max(1, o.workers_per_gpu);
    // A rank of a sharded run (--place, database output): the record and byte counts of every rank must be known before any rank
    // writes, so that every record is appended ONCE, at its final offset of the final data file (the reference appends every record
    // once: writer_append, src/database_writer.cpp:36-58, from src/main.cpp:656-664). The text of a record is ~40x its FCZ bytes and
    // its exact size needs the decoded numbers (printf widens a column that overflows), so the range is walked twice:
    //   sizes pass   the same producer and workers, fcz_decompress_pdb_sizes instead of _begin / _fetch: entries read (16.5 B per
    //                residue), decoded and measured on the device, nothing formatted, copied back or written; --check and entries
    //                that do not decode are decided here exactly as the real pass decides them, so the count is final;
    //   exchange     {"phase": "sizes", records, data_bytes} on stdout; the caller (foldcomp_amd/sharded_cli.py) does the run's one
    //                all_gather and answers `key0 off0 total` on stdin (or `abort`);
    //   real pass    the sequencer starts at key0 / off0: every job pwrites into <output> itself, index / lookup lines carry final
    //                keys and offsets (rank 0: <output>.index, rank R: <output>.index.R -- rank 0 only concatenates line files).
    // (--place is the sharded driver's handshake: without a database output there is nothing to place, and the driver would wait for a
    //  sizes line that never comes)
    if (o.place && (!o.db || single)) { fprintf(stderr, "[Error] --place needs `decompress -d` (--db) of a directory or database input\n"); return 1; }
    const bool place = o.place;
    pinned_enabled() = true;
    int db_fd = -1;
    if (!o.db && !o.tar && !single) make_dir(output);

    InputPlan plan;
    Sequencer seq;
    std::atomic<bool> hard_fail{false};
    std::atomic<uint64_t> n_ok{0}, n_text{0}, n_fcz{0}, n_res{0};
    std::vector<double> gpu_busy(n_workers, 0.0), ctx_ready(n_workers, 0.0), wait_s(n_workers, 0.0), write_s(n_workers, 0.0), alloc_s(n_workers, 0.0);
    std::vector<fcz_ctx*> ctxs(n_workers, nullptr);
    std::vector<char> ctx_tried(n_workers, 0);
    // what the sizes pass found per job (records that decode, bytes incl. their terminators): the real pass must find the same
    std::mutex planned_m; std::vector<std::pair<uint64_t, uint64_t>> planned;
    std::atomic<uint64_t> planned_records{0}, planned_bytes{0};
    bool plan_built = false;
    double t_queued = 0.0;

    auto run_pass = [&](const bool sizes_only) {
        JobQueue<DecompressJob> queue((size_t)n_workers + 2);
        std::vector<std::thread> workers;
        for (int w = 0; w < n_workers; w++) workers.emplace_back([&, w]() {
            if (!ctx_tried[w]) {
                ctx_tried[w] = 1;
                if (fcz_ctx_create(o.device + w % gpus, &ctxs[w]) != FCZ_OK) { ctxs[w] = nullptr; hard_fail = true; fprintf(stderr, "[Error] no ctx on device %d\n", o.device + w % gpus); }
                ctx_ready[w] = std::chrono::duration<double>(clk::now() - t_start).count();
            }
            fcz_ctx* ctx = ctxs[w];
            pvec<uint8_t> text, packed;
            DecompressJob job;
            for (;;) {
                const auto tw = clk::now();
                const bool more = queue.get(job);
                if (!sizes_only) wait_s[w] += std::chrono::duration<double>(clk::now() - tw).count();
                if (!more) break;
                const uint32_t n = job.ents.n();
                std::vector<uint64_t> text_off(n + 1, 0);
                std::vector<int32_t> status(n, 0);
                const auto t0 = clk::now();
                // a database record is the text + the MMseqs terminator (src/main.cpp:659): the device leaves the NULs in place, so a job's
                // records are one contiguous range of the data file
                const int flags = (o.alt ? FCZ_PDB_ALT_ORDER : 0) | (o.db ? FCZ_PDB_NUL_TERMINATED : 0);
                int rc = !ctx ? FCZ_E_NO_DEVICE
                       : sizes_only ? fcz_decompress_pdb_sizes(ctx, job.ents.blob.data(), job.ents.off.data(), n, flags, text_off.data(), status.data())
                                    : fcz_decompress_pdb_begin(ctx, job.ents.blob.data(), job.ents.off.data(), n, flags, text_off.data(), status.data());
                if (!sizes_only) gpu_busy[w] += std::chrono::duration<double>(clk::now() - t0).count();
                uint32_t n_good = 0;
                if (rc == FCZ_OK) for (uint32_t i = 0; i < n; i++) n_good += status[i] == FCZ_OK ? 1u : 0u;
                const uint64_t bytes = rc == FCZ_OK ? text_off[n] : 0;
                if (sizes_only) {
                    if (rc != FCZ_OK) { fprintf(stderr, "[Error] %s\n", fcz_status_string(rc)); hard_fail = true; continue; }
                    std::lock_guard<std::mutex> l(planned_m);
                    if (planned.size() <= job.index) planned.resize(job.index + 1, {0, 0});
                    planned[job.index] = {n_good, bytes};
                    planned_records += n_good; planned_bytes += bytes;
                    continue;
                }
                if (place) {
                    // the placement rests on the sizes pass: a job that measures differently now (the input changed under the run) would
                    // write into another job's -- or another rank's -- range
                    std::pair<uint64_t, uint64_t> want{0, 0};
                    { std::lock_guard<std::mutex> l(planned_m); if (job.index < planned.size()) want = planned[job.index]; }
                    if (rc == FCZ_OK && (want.first != n_good || want.second != bytes)) {
                        fprintf(stderr, "[Error] job %zu measures %llu bytes now, %llu in the sizes pass: the input changed during the run\n", job.index, (unsigned long long)bytes, (unsigned long long)want.second);
                        rc = FCZ_E_INVALID_ARG;
                    }
                }
                std::vector<uint64_t> lens; std::vector<std::string> dbnames;
                if (o.db && rc == FCZ_OK) for (uint32_t i = 0; i < n; i++) {
                    if (status[i] != FCZ_OK) continue;
                    std::string stem, ext; file_parts(base_name(job.ents.names[i]), stem, ext);
                    lens.push_back(text_off[i + 1] - text_off[i]); dbnames.push_back(stem);
                }
                // -z: an entry's text becomes the member <stem>.pdb (src/main.cpp:646-647, :666-674)
                std::vector<std::string> members; uint64_t tar_bytes = 0;
                if (o.tar && rc == FCZ_OK) for (uint32_t i = 0; i < n; i++) {
                    if (status[i] != FCZ_OK) continue;
                    std::string stem, ext; file_parts(base_name(job.ents.names[i]), stem, ext);
                    members.push_back(stem + ".pdb"); tar_bytes += tar_record_bytes(text_off[i + 1] - text_off[i]);
                }
                const uint64_t at = o.db ? seq.claim(job.index, rc == FCZ_OK ? bytes : 0, &lens, &dbnames)     // every job claims, also a failed one
                                  : o.tar ? seq.claim(job.index, tar_bytes) : 0;
                if (rc == FCZ_OK) {
                    const auto ta = clk::now();
                    if (text.size() < text_off[n]) text.resize(text_off[n] + text_off[n] / 8);     // grows, never shrinks (a resize zero-fills what it adds)
                    alloc_s[w] += std::chrono::duration<double>(clk::now() - ta).count();
                    const auto t1 = clk::now();
                    rc = fcz_decompress_pdb_fetch(ctx, text.data());
                    gpu_busy[w] += std::chrono::duration<double>(clk::now() - t1).count();
                    for (uint32_t i = 0; i < n; i++) if (status[i] == FCZ_OK) { const uint8_t* e = job.ents.blob.data() + job.ents.off[i]; n_res += (uint32_t)e[4] | ((uint32_t)e[5] << 8); }
                }
                if (rc != FCZ_OK) { fprintf(stderr, "[Error] %s\n", fcz_status_string(rc)); hard_fail = true; continue; }
                const auto t_write = clk::now();
                try {
                    if (o.db) {
                        // the job's records lie in the fetched buffer exactly as in the data file: one run of large writes. (The copy into
                        // the page cache is the cost of this direction -- the text is 40x the FCZ bytes -- and writes to ONE file
                        // serialise on its inode lock: large writes from one thread reach what the file system gives,
                        // tools/dbg/write_bench.cpp; more writer threads only add contention.)
                        for (uint32_t i = 0; i < n; i++) if (status[i] != FCZ_OK) fprintf(stderr, "[Error] decompressing %s\n", job.ents.names[i].c_str());
                        pwrite_all(db_fd, text.data(), bytes, at);
                    } else if (o.tar) {
                        // header, text, padding per member, gathered by the kernel from where they lie (no second copy of the text)
                        static const uint8_t zeros[512] = {0};
                        std::vector<uint8_t> heads(512 * members.size());
                        std::vector<iovec> iov; iov.reserve(3 * members.size());
                        size_t k = 0;
                        for (uint32_t i = 0; i < n; i++) {
                            if (status[i] != FCZ_OK) { fprintf(stderr, "[Error] decompressing %s\n", job.ents.names[i].c_str()); continue; }
                            const uint64_t len = text_off[i + 1] - text_off[i];
                            tar_header(heads.data() + 512 * k, members[k], len);
                            iov.push_back({heads.data() + 512 * k, 512});
                            iov.push_back({text.data() + text_off[i], (size_t)len});
                            iov.push_back({(void*)zeros, (size_t)TarStream::pad(len)});
                            k++;
                        }
                        pwritev_all(db_fd, iov, at);
                    } else {
                        // one file per entry, written by several threads (open / write / close per file is what takes the time)
                        const int pieces = (int)std::min<size_t>(std::max<size_t>(n / 64, 1), (size_t)std::max(1, o.write_threads));
                        std::vector<std::thread> wt;
                        for (int pc = 0; pc < pieces; pc++) wt.emplace_back([&, pc]() {
                            for (uint32_t i = (uint32_t)((uint64_t)n * pc / pieces); i < (uint32_t)((uint64_t)n * (pc + 1) / pieces); i++) {
                                if (status[i] != FCZ_OK) { fprintf(stderr, "[Error] decompressing %s\n", job.ents.names[i].c_str()); continue; }
                                std::string stem, ext;
                                file_parts(base_name(job.ents.names[i]), stem, ext);
                                const std::string fname = stem + ".pdb";      // getFileParts(baseName(name)).first + ".pdb" (src/main.cpp:646-653)
                                write_out(single ? output : output + "/" + fname, (const char*)text.data() + text_off[i], text_off[i + 1] - text_off[i], o.overwrite);
                            }
                        });
                        for (std::thread& t : wt) t.join();
                    }
                    n_ok += n_good; n_text += text_off[n] - (o.db ? n_good : 0); n_fcz += job.ents.off.back();
                } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
                write_s[w] += std::chrono::duration<double>(clk::now() - t_write).count();
            }
        });
        {
            const size_t JOB = 2048;            // ~0.5 GB of text per job: page-locked buffers of that size, several jobs in flight
            size_t job_index = 0;
            Entries ents;
            auto flush = [&]() {
                if (!ents.n()) return;
                if (o.check) {
                    // --check: entries that fail Foldcomp::checkValidity are reported and left out (src/main.cpp:629-636)
                    Entries ok;
                    for (uint32_t i = 0; i < ents.n(); i++) {
                        const uint64_t len = ents.off[i + 1] - ents.off[i];
                        const int rc = len ? fcz_check(ents.blob.data() + ents.off[i], len) : FCZ_E_TRUNCATED;
                        if (rc != 0) {
                            // the reference prints printValidityError's line with the record's TITLE (src/main.cpp:630-635)
                            static const char* what[] = {"", "Number of backbone angles does not match header", "Number of sidechain angles does not match header",
                                                         "Number of temperature factors does not match header", "All backbone angles are empty",
                                                         "All sidechain angles are empty", "All temperature factors are empty"};
                            if (!sizes_only) {
                                if (rc >= 1 && rc <= 6 && len >= 76) {
                                    const uint8_t* e = ents.blob.data() + ents.off[i];
                                    const uint64_t o_title = 76 + 4 * (uint64_t)e[12], tl = (uint64_t)e[24] | ((uint64_t)e[25] << 8) | ((uint64_t)e[26] << 16) | ((uint64_t)e[27] << 24);
                                    const std::string title = o_title + tl <= len ? std::string((const char*)e + o_title, (size_t)tl) : std::string();
                                    fprintf(stderr, "[Error] %s: %s\n", what[rc], title.c_str());
                                } else fprintf(stderr, "[Error] invalid FCZ entry skipped: %s\n", ents.names[i].c_str());
                            }
                            continue;
                        }
                        ok.add(ents.names[i], std::string((const char*)ents.blob.data() + ents.off[i], len));
                    }
                    ents = std::move(ok);
                    if (!ents.n()) { ents = Entries(); return; }
                }
                DecompressJob j; j.index = job_index++; j.ents = std::move(ents);
                ents = Entries();
                queue.put(std::move(j));
            };
            if (single) for_each_entry(o, ents, flush, JOB);
            else try {
                // this process's range of the listing (all of it unless --shard), database entries streamed from their files
                if (!plan_built) { plan.build(o); plan_built = true; }
                plan.for_each([&](const InputItem& it) {
                    if (it.kind == 0) { try { ents.add(it.name, read_file(it.name)); } catch (const std::exception& e) { if (!sizes_only) fprintf(stderr, "[Error] %s\n", e.what()); } }
                    else if (!plan.entry_in_range(it)) { if (!sizes_only) fprintf(stderr, "[Error] database entry out of range: %s\n", it.name.c_str()); }   // (checked before anything is sized from it)
                    else {
                        const size_t at = ents.blob.size();
                        ents.blob.resize(at + it.len);
                        if (!plan.read_entry(it, ents.blob.data() + at)) { ents.blob.resize(at); if (!sizes_only) fprintf(stderr, "[Error] database entry out of range: %s\n", it.name.c_str()); }
                        else { ents.names.push_back(it.name); ents.off.push_back(ents.blob.size()); }
                    }
                    if (ents.n() >= JOB) flush();
                });
                flush();
            } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
            queue.close();
        }
        if (!sizes_only) t_queued = std::chrono::duration<double>(clk::now() - t_start).count();
        for (std::thread& t : workers) t.join();
    };

    double sizes_pass_s = 0.0, placed_wait_s = 0.0;
    unsigned long long total_bytes = 0;
    if (place) {
        run_pass(true);
        sizes_pass_s = std::chrono::duration<double>(clk::now() - t_start).count();
        printf("{\"phase\": \"sizes\", \"records\": %llu, \"data_bytes\": %llu, \"failed\": %s, \"sizes_pass_s\": %.4f}\n", (unsigned long long)planned_records.load(),
               (unsigned long long)planned_bytes.load(), hard_fail ? "true" : "false", sizes_pass_s);
        fflush(stdout);
        const auto tp = clk::now();
        char line[256]; long long key0 = 0; unsigned long long off0 = 0;
        const bool got = fgets(line, sizeof line, stdin) != nullptr && sscanf(line, "%lld %llu %llu", &key0, &off0, &total_bytes) == 3;
        placed_wait_s = std::chrono::duration<double>(clk::now() - tp).count();
        if (!got || hard_fail) {
            if (hard_fail) fprintf(stderr, "[Error] the sizes pass failed: nothing was written\n");
            for (fcz_ctx* c : ctxs) if (c) fcz_ctx_destroy(c);
            return 1;                                                    // (`abort`, or the caller went away: nothing was written)
        }
        seq.key = key0; seq.pos = off0;
    }
    if (o.db) {
        // a placed rank writes into the final file beside the other ranks: created, never truncated (the caller removed what was there)
        db_fd = open(output.c_str(), place ? (O_CREAT | O_WRONLY) : (O_CREAT | O_TRUNC | O_WRONLY), 0666);
        if (db_fd < 0) { fprintf(stderr, "[Error] cannot write %s\n", output.c_str()); return 1; }
        if (!seq.open_index(output, place && o.shard_rank > 0 ? "." + std::to_string(o.shard_rank) : "")) { fprintf(stderr, "[Error] cannot write %s.index\n", output.c_str()); return 1; }
    } else if (o.tar) {
        db_fd = open(output.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
        if (db_fd < 0) { fprintf(stderr, "[Error] cannot write %s\n", output.c_str()); return 1; }
    }
    const uint64_t pos0 = seq.pos;
    run_pass(false);
    for (fcz_ctx* c : ctxs) if (c) fcz_ctx_destroy(c);
    if (o.db) {
        if (place && !hard_fail && seq.pos - pos0 != planned_bytes.load()) { fprintf(stderr, "[Error] wrote %llu bytes, planned %llu\n", (unsigned long long)(seq.pos - pos0), (unsigned long long)planned_bytes.load()); hard_fail = true; }
        // (a pre-existing longer file cannot be: the caller removed it; rank 0 still pins the size, which also covers ranks without records)
        if (place && !hard_fail && o.shard_rank == 0 && ftruncate(db_fd, (off_t)total_bytes) != 0) hard_fail = true;
        close(db_fd);
        if (place) {
            // the data file and the dbtype belong to the whole run: on failure this rank removes only its own line files, the caller
            // removes the rest once every rank has reported
            if (seq.fi) fclose(seq.fi);
            if (seq.fl) fclose(seq.fl);
            seq.fi = seq.fl = nullptr;
            const std::string tag = o.shard_rank > 0 ? "." + std::to_string(o.shard_rank) : "";
            if (hard_fail) { unlink((output + ".index" + tag).c_str()); unlink((output + ".lookup" + tag).c_str()); }
            else if (o.shard_rank == 0) { std::ofstream t(output + ".dbtype", std::ios::binary); const int32_t twelve = 12; t.write((const char*)&twelve, 4); }
        } else seq.finish(output, !hard_fail);
        if (hard_fail) fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str());
    } else if (o.tar) {
        if (!hard_fail && !tar_finalize(db_fd, seq.pos)) hard_fail = true;
        close(db_fd);
        if (hard_fail) { unlink(output.c_str()); fprintf(stderr, "[Error] the run failed: %s was not written\n", output.c_str()); }
    }
    if (o.json_stats) {
        const double wall = std::chrono::duration<double>(clk::now() - t_start).count();
        double busy = 0.0, waited = 0.0, wrote = 0.0, alloc = 0.0;
        for (double g : alloc_s) alloc += g;
        for (double g : gpu_busy) busy += g;
        for (double g : wait_s) waited += g;
        for (double g : write_s) wrote += g;
        printf("{\"mode\": \"decompress\", \"gpus\": %d, \"workers\": %d, \"records\": %llu, \"residues\": %llu, \"fcz_bytes\": %llu, \"text_bytes\": %llu, "
               "\"wall_s\": %.4f, \"codec_call_s_sum\": %.4f, \"ctx_ready_s\": %.4f, \"all_queued_s\": %.4f, \"residues_per_s\": %.1f, \"text_MB_per_s\": %.1f, "
               "\"pinned_blocks\": %llu, \"queue_wait_s_sum\": %.4f, \"write_s_sum\": %.4f, \"buffer_alloc_s_sum\": %.4f, \"host_threads\": %d, "
               "\"shard\": \"%d/%d\", \"items\": %llu, \"items_total\": %llu, \"data_bytes\": %llu, \"streamed_inputs\": %s, \"max_rss_kb\": %ld, "
               "\"placed\": %s, \"sizes_pass_s\": %.4f, \"placement_wait_s\": %.4f}\n", gpus, n_workers, (unsigned long long)n_ok.load(), (unsigned long long)n_res.load(), (unsigned long long)n_fcz.load(),
               (unsigned long long)n_text.load(), wall, busy, *std::max_element(ctx_ready.begin(), ctx_ready.end()), t_queued,
               wall > 0 ? n_res.load() / wall : 0.0, wall > 0 ? n_text.load() / wall / 1e6 : 0.0, (unsigned long long)pinned_blocks().load(), waited, wrote, alloc, o.write_threads * n_workers,
               o.shard_rank, o.shard_world, (unsigned long long)plan.n_mine, (unsigned long long)plan.n_items, (unsigned long long)(seq.pos - pos0), plan.streamed_all ? "true" : "false", max_rss_kb(),
               place ? "true" : "false", sizes_pass_s, placed_wait_s);
    }
    return hard_fail ? 1 : 0;
}

// title and residue count straight from the FCZ header (src/foldcomp.h:118-136)
bool fcz_header(const uint8_t* e, uint64_t len, std::string& title, uint32_t& n_res) {
    if (len < 76 || memcmp(e, "FCMP", 4) != 0) return false;
    n_res = (uint32_t)e[4] | ((uint32_t)e[5] << 8);
    uint32_t tl; memcpy(&tl, e + 24, 4);
    const uint64_t o_title = 76 + 4ull * e[12];
    if (o_title + tl > len) return false;
    title.assign((const char*)e + o_title, tl);
    return true;
}

std::string extract_suffix(const Options& o) {
    return o.ext_mode == 1 ? "fasta" : (std::min(std::max(o.digits, 1), 4) == 1 ? "plddt" : "plddt.tsv");
}
// extract: the same three stages as the other directions, sized for a direction whose INPUT is the large side (5.7 KB of record per
// 375 bytes of answer): the calling thread gathers batches of entries on all host threads into page-locked buffers (for_each_entry),
// the device thread -- which creates the ctx while the first batch is being read -- sizes and extracts them (fcz_extract: one DMA
// in, one kernel, one DMA out), the output thread lays the answers out as the reference's writers do (writeFASTALike / writeTSV,
// src/main.cpp:738-741, :790-850) on all host threads and writes them as they come: nothing of a run stays in memory but the
// buffers in flight. Entries keep the order of the inputs (the reference's order under `omp critical` is schedule dependent).
struct ExtractBuf {
    size_t index = 0; RawEntries ents; std::vector<uint64_t> data_off; rvec<uint8_t> data; int rc = FCZ_OK;
};
int run_extract(const Options& o) {
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    auto since = [&](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
    const bool single = o.single;
    const int digits = std::min(std::max(o.digits, 1), 4);
    const int mode = o.ext_mode == 1 ? 1 : 0;
    const std::string suffix = extract_suffix(o);
    const std::string output = o.output;
    // where an entry's text goes (src/main.cpp:738-741, :790-850): a tar member <stem>.<suffix> (-z), a database record under <stem>
    // with the MMseqs terminator (-d), the one output file of a single input, one merged file, or -- --no-merge -- <output>/<stem>.<suffix>
    const bool per_entry = !o.tar && !o.db && !single && !o.merge;
    if (fcz_device_count() <= 0) { fprintf(stderr, "[Error] %s\n", fcz_status_string(FCZ_E_NO_DEVICE)); return 1; }
    if (per_entry) make_dir(output);
    pinned_enabled() = true;
    std::unique_ptr<DbWriter> dbw;
    if (o.db) { try { dbw.reset(new DbWriter(output)); } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); return 1; } }

    constexpr int N_BUF = 3;
    ExtractBuf bufs[N_BUF];
    JobQueue<ExtractBuf*> free_q(N_BUF + 1), dev_q(N_BUF + 1), out_q(N_BUF + 1);
    for (ExtractBuf& b : bufs) free_q.put(&b);
    std::atomic<bool> hard_fail{false};
    double ctx_ready_s = 0.0, device_s = 0.0, format_s = 0.0, write_s = 0.0, read_s = 0.0;
    uint64_t n_entries = 0, n_in_bytes = 0, n_out_bytes = 0;

    std::thread device([&]() {
        fcz_ctx* ctx = nullptr;
        const int crc = fcz_ctx_create(o.device, &ctx);
        if (crc != FCZ_OK) { fprintf(stderr, "[Error] %s\n", fcz_status_string(crc)); hard_fail = true; ctx = nullptr; }
        ctx_ready_s = since(t_start);
        ExtractBuf* b;
        while (dev_q.get(b)) {
            const auto t0 = clk::now();
            const uint32_t n = b->ents.n();
            b->data_off.assign((size_t)n + 1, 0);
            b->rc = ctx ? fcz_extract_sizes(b->ents.blob.data(), b->ents.off.data(), n, mode, digits, b->data_off.data()) : FCZ_E_NO_DEVICE;
            if (b->rc == FCZ_OK) {
                b->data.resize(b->data_off[n]);
                b->rc = fcz_extract(ctx, b->ents.blob.data(), b->ents.off.data(), n, mode, digits, b->data_off.data(), b->data.data());
            }
            device_s += since(t0);
            out_q.put(std::move(b));
        }
        out_q.close();
        if (ctx) fcz_ctx_destroy(ctx);
    });

    std::thread writer([&]() {
        int out_fd = -1; uint64_t out_pos = 0; long long key = 0;
        auto open_out = [&]() {
            if (out_fd >= 0 || per_entry || o.db) return true;
            out_fd = open(output.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
            if (out_fd < 0) { fprintf(stderr, "[Error] cannot write %s\n", output.c_str()); hard_fail = true; return false; }
            return true;
        };
        std::vector<uint64_t> text_off; std::vector<uint32_t> n_res; std::vector<std::string> titles, stems; std::vector<uint8_t> text, packed;
        ExtractBuf* b;
        while (out_q.get(b)) {
            const uint32_t n = b->ents.n();
            if (b->rc != FCZ_OK) { if (b->rc != FCZ_E_NO_DEVICE) fprintf(stderr, "[Error] %s\n", fcz_status_string(b->rc)); b->ents.clear(); free_q.put(std::move(b)); continue; }
            const auto t0 = clk::now();
            // in order: which entries have an answer, under which title, where its text goes
            text_off.assign((size_t)n + 1, 0); n_res.assign(n, 0); titles.resize(n); if (!(single || (!o.tar && !o.db && o.merge))) stems.resize(n);
            const bool tsv = mode == 0 && digits > 1;
#pragma omp parallel for schedule(static)
            for (long long q = 0; q < (long long)n; q++) {
                const uint32_t i = (uint32_t)q;
                const uint64_t len = b->ents.off[i + 1] - b->ents.off[i], dl = b->data_off[i + 1] - b->data_off[i];
                if (!fcz_header(b->ents.blob.data() + b->ents.off[i], len, titles[i], n_res[i]) || (dl == 0 && n_res[i])) { text_off[i + 1] = UINT64_MAX; continue; }
                if (!o.use_title) titles[i] = b->ents.names[i];   // the entry's name as the run met it -- a directory's file with its path (src/main.cpp:780-781)
                char num[16];
                text_off[i + 1] = tsv ? titles[i].size() + 1 + (size_t)snprintf(num, sizeof num, "%u", n_res[i]) + 1 + dl + 1      // writeTSV: title \t n \t values \n
                                      : 1 + titles[i].size() + 1 + dl + 1;                                                           // writeFASTALike: >title \n values \n
                if (!stems.empty()) { std::string ext; file_parts(base_name(b->ents.names[i]), stems[i], ext); }
            }
            std::vector<char> good(n, 1);
            for (uint32_t i = 0; i < n; i++) {
                if (text_off[i + 1] == UINT64_MAX) { fprintf(stderr, "[Error] reading %s\n", b->ents.names[i].c_str()); good[i] = 0; text_off[i + 1] = 0; }
                text_off[i + 1] += text_off[i];
            }
            text.resize(text_off[n]);
#pragma omp parallel for schedule(static)
            for (long long q = 0; q < (long long)n; q++) {
                const uint32_t i = (uint32_t)q;
                if (!good[i]) continue;
                uint8_t* w = text.data() + text_off[i];
                const uint64_t dl = b->data_off[i + 1] - b->data_off[i];
                if (tsv) {
                    memcpy(w, titles[i].data(), titles[i].size()); w += titles[i].size(); *w++ = '\t';
                    w += sprintf((char*)w, "%u", n_res[i]); *w++ = '\t';      // (the terminator lands on the byte the values overwrite next)
                } else { *w++ = '>'; memcpy(w, titles[i].data(), titles[i].size()); w += titles[i].size(); *w++ = '\n'; }
                if (dl) memcpy(w, b->data.data() + b->data_off[i], dl);
                w[dl] = '\n';
            }
            format_s += since(t0);
            const auto t1 = clk::now();
            if (o.tar) {
                uint64_t total = 0;
                for (uint32_t i = 0; i < n; i++) if (good[i]) total += tar_record_bytes(text_off[i + 1] - text_off[i]);
                packed.assign(total, 0);
                uint64_t at = 0;
                for (uint32_t i = 0; i < n; i++) {
                    if (!good[i]) continue;
                    const uint64_t len = text_off[i + 1] - text_off[i];
                    tar_header(packed.data() + at, stems[i] + "." + suffix, len);
                    memcpy(packed.data() + at + 512, text.data() + text_off[i], len);
                    at += tar_record_bytes(len);
                }
                if (open_out()) { try { pwrite_all(out_fd, packed.data(), total, out_pos); out_pos += total; } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; } }
            } else if (o.db) {
                for (uint32_t i = 0; i < n; i++) if (good[i]) dbw->append((const char*)text.data() + text_off[i], text_off[i + 1] - text_off[i], key++, stems[i], true);
            } else if (per_entry) {
#pragma omp parallel for schedule(dynamic, 64)
                for (long long q = 0; q < (long long)n; q++) {
                    const uint32_t i = (uint32_t)q;
                    if (good[i]) write_out(output + "/" + stems[i] + "." + suffix, (const char*)text.data() + text_off[i], text_off[i + 1] - text_off[i], true);
                }
            } else if (open_out()) {
                try { pwrite_all(out_fd, text.data(), text_off[n], out_pos); out_pos += text_off[n]; } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
            }
            for (uint32_t i = 0; i < n; i++) n_entries += good[i] ? 1u : 0u;
            n_in_bytes += b->ents.off[n]; n_out_bytes += text_off[n];
            write_s += since(t1);
            b->ents.clear();
            free_q.put(std::move(b));
        }
        // a run without entries still leaves its (empty) output, an archive its two closing records (mtar_finalize)
        if (!hard_fail && open_out() && o.tar) { static const uint8_t zeros[1024] = {0}; try { pwrite_all(out_fd, zeros, 1024, out_pos); } catch (const std::exception&) { hard_fail = true; } }
        if (out_fd >= 0) close(out_fd);
    });

    {
        RawEntries ents;
        ExtractBuf* cur = nullptr;
        size_t job = 0;
        auto t_read = clk::now();
        auto flush = [&]() {
            if (!ents.n()) return;
            read_s += since(t_read);
            ExtractBuf* b = nullptr;
            free_q.get(b);
            std::swap(b->ents, ents);             // the gather's buffers travel with the job, a finished job's come back (capacity kept)
            ents.clear();
            b->index = job++;
            dev_q.put(std::move(b));
            t_read = clk::now();
        };
        (void)cur;
        try { for_each_entry(o, ents, flush); } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); hard_fail = true; }
        dev_q.close();
    }
    device.join();
    writer.join();
    if (o.db) dbw->close();
    if (o.json_stats) {
        const double wall = since(t_start);
        printf("{\"mode\": \"extract\", \"entries\": %llu, \"fcz_bytes\": %llu, \"text_bytes\": %llu, \"wall_s\": %.4f, \"ctx_ready_s\": %.4f, \"read_s_sum\": %.4f, "
               "\"device_call_s_sum\": %.4f, \"format_s_sum\": %.4f, \"write_s_sum\": %.4f, \"entries_per_s\": %.1f, \"host_threads\": %d, \"pinned_blocks\": %llu, \"max_rss_kb\": %ld}\n",
               (unsigned long long)n_entries, (unsigned long long)n_in_bytes, (unsigned long long)n_out_bytes, wall, ctx_ready_s, read_s, device_s, format_s, write_s,
               wall > 0 ? n_entries / wall : 0.0, omp_get_max_threads(), (unsigned long long)pinned_blocks().load(), max_rss_kb());
    }
    return hard_fail ? 1 : 0;
}

int run_check(const Options& o) {
    // the lines of printValidityError (src/foldcomp.cpp:1534-1560), word for word: "[Error] <what>: <name>"
    static const char* msgs[] = {"", "Number of backbone angles does not match header", "Number of sidechain angles does not match header",
                                 "Number of temperature factors does not match header", "All backbone angles are empty",
                                 "All sidechain angles are empty", "All temperature factors are empty"};
    // checkValidity of every entry on all host threads (the reference: process_entry_func under `omp parallel for`,
    // src/main.cpp:911-930); the lines of a batch are printed in input order, one write per stream
    Entries ents;
    std::vector<int> rcs; std::string out, err;
    auto flush = [&]() {
        const uint32_t n = ents.n();
        rcs.assign(n, 0);
#pragma omp parallel for schedule(static)
        for (long long q = 0; q < (long long)n; q++) {
            const uint64_t len = ents.off[(size_t)q + 1] - ents.off[(size_t)q];
            rcs[(size_t)q] = len ? fcz_check(ents.blob.data() + ents.off[(size_t)q], len) : -5;
        }
        out.clear(); err.clear();
        for (uint32_t i = 0; i < n; i++) {
            const int rc = rcs[i];
            if (rc == 0) { out += "[Info] "; out += ents.names[i]; out += " is valid.\n"; }
            else if (rc >= 1 && rc <= 6) { err += "[Error] "; err += msgs[rc]; err += ": "; err += ents.names[i]; err += "\n"; }
            else { err += "[Error] "; err += ents.names[i]; err += ": not a valid FCZ entry\n"; }      // (the reference: "[Error] File is not a valid fcz file", without the name)
        }
        if (!out.empty()) fwrite(out.data(), 1, out.size(), stdout);
        if (!err.empty()) fwrite(err.data(), 1, err.size(), stderr);
        ents.clear();
    };
    for_each_entry(o, ents, flush);
    return 0;
}

// pure host: files of a directory -> database / database -> files (no GPU)
int run_db_pack(const Options& o) {
    if (o.output.empty()) { fprintf(stderr, "[Error] db-pack needs an output database.\n"); return 1; }
    std::vector<std::string> files;
    list_files(o.input, o.recursive, files);
    DbWriter w(o.output);
    long long key = 0;
    for (const std::string& path : files) {
        std::string stem, ext;
        file_parts(base_name(path), stem, ext);
        const std::string d = read_file(path);
        w.append(d.data(), d.size(), key++, stem, false);
    }
    w.close();
    return 0;
}
// files of a directory -> tar archive with the reference writer's headers (no GPU; what the -z outputs are made of)
int run_tar_pack(const Options& o) {
    if (o.output.empty()) { fprintf(stderr, "[Error] tar-pack needs an output archive.\n"); return 1; }
    std::vector<std::string> files;
    list_files(o.input, o.recursive, files);
    std::vector<std::string> data(files.size());
    for (size_t i = 0; i < files.size(); i++) data[i] = read_file(files[i]);
    std::vector<uint8_t> archive;
    tar_pack_members(archive, files.size(), [&](size_t q) { return base_name(files[q]); }, [&](size_t q) { return data[q].data(); }, [&](size_t q) { return (uint64_t)data[q].size(); });
    archive.resize(archive.size() + 1024, 0);
    return write_out(o.output, (const char*)archive.data(), archive.size(), true) ? 0 : 1;
}
int run_db_unpack(const Options& o) {
    if (o.output.empty()) { fprintf(stderr, "[Error] db-unpack needs an output directory.\n"); return 1; }
    DbReader r(o.input);
    make_dir(o.output);
    for (size_t i = 0; i < r.n(); i++) {
        const std::string d = r.entry(i);
        // names come from the (untrusted) .lookup file: only their last path component is used
        write_out(o.output + "/" + base_name(r.name(i)), d.data(), d.size(), true);
    }
    return 0;
}

// ---- db-splice: the exchange step of a sharded database run, file side (free_writer's layout,
//      src/database_writer.cpp:59-73). Every rank's engine has written a complete partial database with keys and offsets counted
//      from 0: rank 0 straight into <final>, rank r > 0 into its own <part>. After the ranks have exchanged {records, bytes} (the
//      only collective of the run) rank r knows key0 = records of the ranks before it, off0 = their bytes, and runs
//          db-splice --shard r/N --key0 K --off0 B <part> <final>      r > 0: the data of <part> goes to <final> at byte B (in-kernel
//                       copy, no user-space buffer beyond 8 MB), its index / lookup lines are rewritten with K and B added into
//                       <final>.index.r / <final>.lookup.r, <part>* are removed;
//          db-splice --shard 0/N <final> <final>                        rank 0, after the others: appends <final>.index.1 .. N-1 and
//                       <final>.lookup.1 .. N-1 to its own index and lookup (already in key order: ranks own contiguous key ranges)
//                       and removes them.
//      Nothing per record stays in memory on any rank.
long long g_key0 = 0; unsigned long long g_off0 = 0;
static void copy_range(int in, int out, uint64_t n, uint64_t off_out) {
    off_t oi = 0, oo = (off_t)off_out;
    bool kernel_copy = true;
    std::vector<char> buf;
    while (n) {
        if (kernel_copy) {
            const ssize_t k = copy_file_range(in, &oi, out, &oo, (size_t)std::min<uint64_t>(n, 1u << 30), 0);
            if (k > 0) { n -= (uint64_t)k; continue; }
            if (k == 0) throw std::runtime_error("partial database is shorter than its index says");
            kernel_copy = false;                                      // EXDEV / ENOSYS / EINVAL: plain reads and writes
        }
        if (buf.empty()) buf.resize(8u << 20);
        const ssize_t r = pread(in, buf.data(), (size_t)std::min<uint64_t>(n, buf.size()), oi);
        if (r <= 0) throw std::runtime_error("cannot read the partial database");
        pwrite_all(out, (const uint8_t*)buf.data(), (uint64_t)r, (uint64_t)oo);
        oi += r; oo += r; n -= (uint64_t)r;
    }
}
static void append_file(const std::string& from, FILE* to) {
    const int fd = open(from.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + from);
    std::vector<char> buf(8u << 20);
    for (;;) { const ssize_t k = read(fd, buf.data(), buf.size()); if (k < 0) { close(fd); throw std::runtime_error("cannot read " + from); } if (k == 0) break; if (fwrite(buf.data(), 1, (size_t)k, to) != (size_t)k) { close(fd); throw std::runtime_error("cannot write the index"); } }
    close(fd);
}
int run_db_splice(const Options& o) {
    const std::string part = o.input, fin = o.output;
    try {
        if (o.shard_rank == 0) {
            FILE* fi = fopen((fin + ".index").c_str(), "a"); FILE* fl = fopen((fin + ".lookup").c_str(), "a");
            if (!fi || !fl) throw std::runtime_error("cannot append to " + fin + ".index");
            for (int r = 1; r < o.shard_world; r++) {
                append_file(fin + ".index." + std::to_string(r), fi); append_file(fin + ".lookup." + std::to_string(r), fl);
            }
            if (fclose(fi) != 0 || fclose(fl) != 0) throw std::runtime_error("cannot write " + fin + ".index");
            for (int r = 1; r < o.shard_world; r++) { unlink((fin + ".index." + std::to_string(r)).c_str()); unlink((fin + ".lookup." + std::to_string(r)).c_str()); }
            return 0;
        }
        const int in = open(part.c_str(), O_RDONLY);
        if (in < 0) throw std::runtime_error("cannot open " + part);
        struct stat st; fstat(in, &st);
        const int out = open(fin.c_str(), O_WRONLY | O_CREAT, 0666);
        if (out < 0) { close(in); throw std::runtime_error("cannot write " + fin); }
        copy_range(in, out, (uint64_t)st.st_size, g_off0);
        close(in);
        if (close(out) != 0) throw std::runtime_error("cannot write " + fin);
        const std::string tag = "." + std::to_string(o.shard_rank);
        LineFile li, ll;
        if (!li.open_at(part + ".index") || !ll.open_at(part + ".lookup")) throw std::runtime_error("cannot open " + part + ".index");
        FILE* fi = fopen((fin + ".index" + tag).c_str(), "w"); FILE* fl = fopen((fin + ".lookup" + tag).c_str(), "w");
        if (!fi || !fl) throw std::runtime_error("cannot write " + fin + ".index" + tag);
        std::vector<char> b1(4u << 20), b2(4u << 20);
        setvbuf(fi, b1.data(), _IOFBF, b1.size()); setvbuf(fl, b2.data(), _IOFBF, b2.size());
        const char* q; size_t qn; const char* w[4]; size_t wl[4];
        while (li.next(q, qn)) {
            uint64_t k, off, len;
            if (line_words(q, qn, w, wl) != 3 || !all_digits_u64(w[0], wl[0], k) || !all_digits_u64(w[1], wl[1], off) || !all_digits_u64(w[2], wl[2], len)) throw std::runtime_error("unexpected line in " + part + ".index");
            fprintf(fi, "%llu\t%llu\t%llu\n", (unsigned long long)k + (unsigned long long)g_key0, (unsigned long long)off + g_off0, (unsigned long long)len);
        }
        while (ll.next(q, qn)) {
            uint64_t k;
            if (line_words(q, qn, w, wl) < 2 || !all_digits_u64(w[0], wl[0], k)) throw std::runtime_error("unexpected line in " + part + ".lookup");
            fprintf(fl, "%llu\t", (unsigned long long)k + (unsigned long long)g_key0);
            fwrite(w[1], 1, (size_t)(q + qn - w[1]), fl); fputc('\n', fl);     // the name and what follows it, as written
        }
        if (fclose(fi) != 0 || fclose(fl) != 0) throw std::runtime_error("cannot write " + fin + ".index" + tag);
        for (const char* ext : {"", ".index", ".lookup", ".dbtype"}) unlink((part + ext).c_str());
    } catch (const std::exception& e) { fprintf(stderr, "[Error] db-splice: %s\n", e.what()); return 1; }
    return 0;
}

// the host-side batch of one file as text (no GPU): what fcz_compress_batch would be handed
int run_dump_batch(const Options& o) {
    std::vector<std::string> files;
    if (is_dir(o.input)) list_files(o.input, o.recursive, files); else files.push_back(o.input);
    std::vector<Fragment> frags;
    fragments_of_files(files, 0, files.size(), false, "", true, o, frags);
    Batch b;
    for (const Fragment& f : frags) {
        try { b.add(f.atoms, f.title); printf("fragment %s\n", f.out_name.c_str()); }
        catch (const std::exception& e) { printf("rejected %s: %s\n", f.out_name.c_str(), e.what()); }
    }
    fcz_chain_batch v = b.view(o.brk);
    printf("n_chains %u n_residues %u n_atoms %u anchor %d\n", v.n_chains, v.n_residues, v.n_atoms, v.anchor_threshold);
    auto dump_u32 = [](const char* name, const uint32_t* p, size_t n) { printf("%s", name); for (size_t i = 0; i < n; i++) printf(" %u", p[i]); printf("\n"); };
    dump_u32("res_off", v.res_off, v.n_chains + 1);
    dump_u32("atom_off", v.atom_off, v.n_residues + 1);
    dump_u32("title_off", v.title_off, v.n_chains + 1);
    printf("res_code"); for (uint32_t i = 0; i < v.n_residues; i++) printf(" %u", v.res_code[i]); printf("\n");
    printf("atom_code"); for (uint32_t i = 0; i < v.n_atoms; i++) printf(" %u", v.atom_code[i]); printf("\n");
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    printf("x"); for (uint32_t i = 0; i < v.n_atoms; i++) printf(" %08x", bits(v.x[i])); printf("\n");
    printf("y"); for (uint32_t i = 0; i < v.n_atoms; i++) printf(" %08x", bits(v.y[i])); printf("\n");
    printf("z"); for (uint32_t i = 0; i < v.n_atoms; i++) printf(" %08x", bits(v.z[i])); printf("\n");
    printf("bfac_ca"); for (uint32_t i = 0; i < v.n_residues; i++) printf(" %08x", bits(v.bfac_ca[i])); printf("\n");
    printf("first_res"); for (uint32_t i = 0; i < v.n_chains; i++) printf(" %d", v.first_res_index[i]); printf("\n");
    printf("first_atom"); for (uint32_t i = 0; i < v.n_chains; i++) printf(" %d", v.first_atom_index[i]); printf("\n");
    printf("chain_id %s\n", std::string(v.chain_id, v.n_chains).c_str());
    printf("titles %s\n", std::string(v.titles, v.title_off[v.n_chains]).c_str());
    std::vector<uint64_t> off(v.n_chains + 1);
    fcz_compress_sizes(&v, off.data());
    printf("fcz_size"); for (uint32_t i = 0; i < v.n_chains; i++) printf(" %llu", (unsigned long long)(off[i + 1] - off[i])); printf("\n");
    return 0;
}

// `foldcomp rmsd a.pdb b.pdb` (reference src/main.cpp:1016-1060, RMSD at src/atom_coordinate.cpp:424-434: float accumulation,
// no superposition): prints "file1 file2 n_residues n_atoms backbone_rmsd all_atom_rmsd"
AtomTable load_table(const std::string& path) {
    const std::string base = base_name(path);
    std::string raw = read_file(path), plain = base, title;
    if (ends_with(base, ".gz")) { raw = gunzip(raw); plain = base.substr(0, base.size() - 3); }
    // rmsd reads through StructureReader::load (src/main.cpp:108-111): the format by the name's extension, anything unknown as PDB
    if (ends_with(plain, ".cif") || ends_with(plain, ".mmcif")) return parse_cif_gemmi(raw.data(), raw.size(), title);
    return parse_pdb_gemmi(raw.data(), raw.size(), title);
}
int run_rmsd(const Options& o) {
    if (o.output.empty()) { fprintf(stderr, "[Error] rmsd needs two structure files.\n"); return 1; }
    const AtomTable a = load_table(o.input), b = load_table(o.output);
    if (a.size() != b.size()) { fprintf(stderr, "[Error] The number of atoms in the two structures differ.\n"); return 1; }
    float sum_bb = 0.0f, sum_all = 0.0f; size_t n_bb = 0;
    for (size_t i = 0; i < a.size(); i++) {
        const float dx = a.x[i] - b.x[i], dy = a.y[i] - b.y[i], dz = a.z[i] - b.z[i];
        const float d2 = dx * dx + dy * dy + dz * dz;
        sum_all += d2;
        if (a.atom[i] == PK_N || a.atom[i] == PK_CA || a.atom[i] == PK_C) { sum_bb += d2; n_bb++; }
    }
    size_t n_res = 0;
    for (size_t i = 0; i < a.size(); i++) if (i == 0 || a.res_index[i] != a.res_index[i - 1] || a.chain[i] != a.chain[i - 1]) n_res++;
    const double bb = std::sqrt((double)(sum_bb / (float)std::max<size_t>(n_bb, 1))), all = std::sqrt((double)(sum_all / (float)std::max<size_t>(a.size(), 1)));
    printf("%s\t%s\t%zu\t%zu\t%g\t%g\n", o.input.c_str(), o.output.c_str(), n_res, a.size(), bb, all);
    return 0;
}

void usage() {
    fprintf(stderr,
            "usage: foldcomp-hip compress   [-t threads] [--gpus N] [-b N] [-y] [-r] [-d|-z] [--skip-discontinuous] [--json-stats] <pdb|cif file|dir|tar(.gz)|db> [<fcz file|dir|tar|db>]\n"
            "       foldcomp-hip decompress [--gpus N] [-a] [-y] [-r] [-d|-z] [--check] [-l ids [-m 0|1]] [--json-stats] <fcz file|dir|tar(.gz)|db> [<pdb file|dir|tar|db>]\n"
            "       foldcomp-hip extract    [--plddt|--fasta|--amino-acid] [-p digits] [--no-merge] [--use-title] [-d|-z] [-r] [-l ids [-m 0|1]] <fcz file|dir|tar(.gz)|db> [<out>]\n"
            "       foldcomp-hip check      [-r] [-l ids [-m 0|1]] <fcz file|dir|tar(.gz)|db>\n"
            "       -f / --file: <input> is a text file listing the inputs, one per line (any mode)\n"
            "       -d / --db: the outputs become one database; -z / --tar (or an output named *.tar): one tar archive\n"
            "       foldcomp-hip rmsd       <pdb|cif> <pdb|cif>\n");
}

}  // namespace

// CPUs the process may use, the default for -t: the smallest of the OpenMP default, the scheduler affinity mask and the
// cgroup's CFS quota (v2: cpu.max "quota period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us) -- the same rule as bench.py's
// effective_cores(). More parse threads than the quota only thrash (measured on a 256-thread host under a 16-CPU quota:
// 0.33 s at 16 threads, 1.6 s at 256).
int default_threads() {
    int n = omp_get_max_threads();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && c < n) n = c; }
    auto cut = [&](long long quota, long long period) { if (quota > 0 && period > 0) { const long long c = (quota + period / 2) / period; if (c >= 1 && c < n) n = (int)c; } };
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0}; long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) cut(atoll(q), period);
        fclose(f);
    } else {
        long long quota = -1, period = 0;
        if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
        cut(quota, period);
    }
    return n;
}

int main(int argc, char** argv) {
    Options o;
    std::vector<std::string> pos;
    bool device_mod = false;
    if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(default_threads());
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next_int = [&](int& dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (a == "-a" || a == "--alt") o.alt = true;
        else if (a == "-y" || a == "--overwrite") o.overwrite = true;
        else if (a == "-r" || a == "--recursive") o.recursive = true;
        else if (a == "-b" || a == "--break") next_int(o.brk);
        else if (a == "-p" || a == "--plddt-digits") next_int(o.digits);
        else if (a == "-t" || a == "--threads") { int t = 0; next_int(t); if (t > 0) { omp_set_num_threads(t); o.threads_said = t; } }   // host parse threads
        else if (a == "--gpus") next_int(o.gpus);
        else if (a == "--workers-per-gpu") next_int(o.workers_per_gpu);
        else if (a == "--json-stats") o.json_stats = true;
        else if (a == "--device") next_int(o.device);
        else if (a == "--place") o.place = true;               // decompress -d of a sharded run: sizes pass, placement on stdin, one write (run_decompress)
        else if (a == "--device-mod") device_mod = true;     // the device number wraps at the device count (ranks of a test run that share GPUs)
        else if (a == "--key0") { if (i + 1 < argc) g_key0 = atoll(argv[++i]); }
        else if (a == "--off0") { if (i + 1 < argc) g_off0 = strtoull(argv[++i], nullptr, 10); }
        else if (a == "--shard") {   // R/N: rank R of N of a sharded run
            if (i + 1 >= argc || sscanf(argv[++i], "%d/%d", &o.shard_rank, &o.shard_world) != 2 || o.shard_world < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_world) {
                fprintf(stderr, "[Error] --shard takes R/N with 0 <= R < N\n"); return 1; }
        }
        else if (a == "--plddt") o.ext_mode = 0;
        else if (a == "--fasta" || a == "--amino-acid") o.ext_mode = 1;
        else if (a == "--use-title") o.use_title = true;
        else if (a == "--skip-discontinuous") o.skip_discontinuous = true;
        else if (a == "-d" || a == "--db") o.db = true;
        else if (a == "-z" || a == "--tar") o.tar = true;
        else if (a == "--check") o.check = true;
        else if (a == "--host-parse") o.host_parse = true;
        else if (a == "--host-inflate") o.host_inflate = true;
        else if (a == "--job-files") next_int(o.job_files);
        else if (a == "--no-merge") o.merge = false;
        else if (a == "-f" || a == "--file") o.file_input = true;
        else if (a == "-l" || a == "--id-list") { if (i + 1 < argc) o.id_list = argv[++i]; }
        else if (a == "-m" || a == "--id-mode") { next_int(o.id_mode); if (o.id_mode != 0 && o.id_mode != 1) { fprintf(stderr, "[Error] Invalid id mode. Please use 0 or 1.\n"); usage(); return 1; } }
        else if (a == "-v" || a == "--version") { printf("foldcomp (MI355X / libfcz_hip) 1.0\n"); return 0; }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        // --time (per-entry timers of the reference's one-entry-at-a-time loop, src/main.cpp:439) and --use-cache (a cached sorted index
        // for --id-list runs, src/input_processor.h:209-216) have nothing to switch here -- entries go through the device in jobs, the
        // index is streamed -- and are accepted so that a command line written for the reference runs unchanged
        else if (a == "--time" || a == "--use-cache") {}
        else if (a.size() > 1 && a[0] == '-' && pos.size() < 3 && !(a[1] >= '0' && a[1] <= '9')) { fprintf(stderr, "[Error] unknown option %s\n", a.c_str()); usage(); return 1; }
        else pos.push_back(a);
    }
    if (pos.size() < 2) { usage(); return 0; }
    o.mode = pos[0]; o.input = pos[1];
    while (o.input.size() > 1 && o.input.back() == '/') o.input.pop_back();
    if (pos.size() > 2) { o.output = pos[2]; while (o.output.size() > 1 && o.output.back() == '/') o.output.pop_back(); }
    if (!exists(o.input)) { fprintf(stderr, "[Error] %s does not exist.\n", o.input.c_str()); return 1; }
    if (o.brk <= 0) { fprintf(stderr, "[Error] -b needs a positive value.\n"); return 1; }
    // -f: the input is a list; lines that name a structure / FCZ file are single files (processed last, as one directory-like
    // set), everything else is a directory or database (src/main.cpp:304-325)
    if (o.file_input) {
        std::ifstream f(o.input);
        if (!f) { fprintf(stderr, "[Error] Could not open file %s\n", o.input.c_str()); return 1; }
        std::vector<std::string> singles; std::string line;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (ends_with(line, ".pdb") || ends_with(line, ".pdb.gz") || ends_with(line, ".cif") || ends_with(line, ".cif.gz") || ends_with(line, ".fcz")) singles.push_back(line);
            else o.inputs.push_back(line);
        }
        o.inputs.insert(o.inputs.end(), singles.begin(), singles.end());
    } else o.inputs.push_back(o.input);
    // a tar archive is a container by its NAME (src/main.cpp:341-342), an output that ends in .tar asks for one (:333-335)
    { struct stat st; o.single = !o.file_input && stat(o.input.c_str(), &st) == 0 && S_ISREG(st.st_mode) && !is_db(o.input) && !is_tar_name(o.input); }
    const bool writes_members = o.mode == "compress" || o.mode == "decompress" || o.mode == "extract";
    if (writes_members && ends_with(o.output, ".tar")) o.tar = true;
    if (o.tar && o.db) { fprintf(stderr, "[Error] -z and -d name two different outputs: one of them.\n"); return 1; }
    if (o.tar && (o.shard_world > 1 || o.place)) { fprintf(stderr, "[Error] a sharded run writes a database (-d), not a tar archive.\n"); return 1; }
    if (o.output.empty() && writes_members) {   // src/main.cpp:356-369
        const std::string suffix = o.mode == "compress" ? "fcz" : (o.mode == "decompress" ? "pdb" : extract_suffix(o));
        if (o.db) o.output = o.input + "_db";
        else if (o.tar) o.output = o.input + "." + suffix + ".tar";
        else if (o.single) { const size_t i = o.input.rfind('.'); o.output = (i == std::string::npos ? o.input : o.input.substr(0, i)) + "." + suffix; }
        else o.output = o.input + "_" + suffix;
    }
    if (device_mod && (o.mode == "compress" || o.mode == "decompress")) { const int nd = fcz_device_count(); if (nd > 0) o.device %= nd; }
    // the reference's announcement of what it is about to do (src/main.cpp:392-403, :564-575, :717-734, :870-875), word for word --
    // not from an engine of a sharded run or with --json-stats, whose stdout is read by a program
    if (!o.json_stats && o.shard_world <= 1 && !o.place && (writes_members || o.mode == "check")) {
        const char* verb = o.mode == "compress" ? "Compressing" : o.mode == "decompress" ? "Decompressing" : o.mode == "extract" ? "Extracting" : "Checking";
        if (o.mode == "check") {
            if (o.inputs.size() == 1) printf("Checking %s\n", o.input.c_str());
            else printf("Checking files in %s using %d threads\n", o.input.c_str(), o.threads_said);
        } else if (o.single) printf("%s %s to %s\n", verb, o.input.c_str(), o.output.c_str());
        else {
            printf("%s files in %s using %d threads\n", verb, o.input.c_str(), o.threads_said);
            if (o.db) printf("Output database: %s\n", o.output.c_str());
            else if (o.tar) printf("Output tar file: %s\n", o.output.c_str());
            else if (o.mode != "extract" || !o.merge) printf("Output directory: %s\n", o.output.c_str());
            else printf("Output: %s\n", o.output.c_str());
        }
        fflush(stdout);
    }
    if (o.mode == "compress") return run_compress(o);
    o.write_threads = std::max(1, omp_get_max_threads() / std::max(1, (o.gpus <= 0 ? 1 : o.gpus) * std::max(1, o.workers_per_gpu)));
    if (o.mode == "decompress") return run_decompress(o);
    if (o.mode == "extract") return run_extract(o);
    if (o.mode == "check") return run_check(o);
    if (o.mode == "dump-batch") return run_dump_batch(o);
    if (o.mode == "parse-bench") {   // no GPU: parse every file of a directory on the host threads and report the rate
        std::vector<std::string> files; list_files(o.input, o.recursive, files);
        std::vector<Fragment> frags;
        const auto t0 = std::chrono::steady_clock::now();
        fragments_of_files(files, 0, files.size(), false, "", true, o, frags);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        size_t atoms = 0; for (const Fragment& f : frags) atoms += f.atoms.size();
        printf("files %zu fragments %zu atoms %zu seconds %.4f threads %d\n", files.size(), frags.size(), atoms, dt, omp_get_max_threads());
        return 0;
    }
    if (o.mode == "rmsd") return run_rmsd(o);
    if (o.mode == "plan-dump") {   // no GPU: this process's range of the inputs (--shard R/N), one line per item, then a summary line
        try {
            InputPlan plan; plan.build(o);
            uint64_t bytes = 0;
            plan.for_each([&](const InputItem& it) {
                if (!o.json_stats) printf("%d\t%s\t%llu\t%llu\n", it.kind, it.name.c_str(), (unsigned long long)it.off, (unsigned long long)(it.len == UINT64_MAX ? 0 : it.len));
                bytes += it.len == UINT64_MAX ? 0 : it.len;
            });
            printf("{\"shard\": \"%d/%d\", \"items\": %llu, \"items_total\": %llu, \"bytes\": %llu, \"bytes_total\": %llu, \"streamed_inputs\": %s, \"max_rss_kb\": %ld}\n", o.shard_rank, o.shard_world,
                   (unsigned long long)plan.n_mine, (unsigned long long)plan.n_items, (unsigned long long)bytes, (unsigned long long)plan.bytes_total, plan.streamed_all ? "true" : "false", max_rss_kb());
        } catch (const std::exception& e) { fprintf(stderr, "[Error] %s\n", e.what()); return 1; }
        return 0;
    }
    if (o.mode == "db-splice") return run_db_splice(o);
    if (o.mode == "db-pack") return run_db_pack(o);
    if (o.mode == "tar-pack") return run_tar_pack(o);
    if (o.mode == "db-unpack") return run_db_unpack(o);
    usage();
    return 1;
}

// ---- the readers as a library (host/libfcz_host.so): what the Python command line calls instead of its own restatement of the
//      same rules (foldcomp_amd/structure.py; the two are held equal on mutated files in tests/test_ingest_vs_reference.py) ----
extern "C" {
// data: the bytes of a structure file; gz != 0: a gzip stream to inflate first. PDB or mmCIF by content (loadFromBuffer).
int fcz_host_read_structure(const uint8_t* data, uint64_t len, int gz, fcz_host_atoms* out) {
    memset(out, 0, sizeof *out);
    try {
        std::string unz;
        const char* d = (const char*)data; size_t n = (size_t)len;
        if (gz) { unz = gunzip(std::string(d, n)); d = unz.data(); n = unz.size(); }
        std::string title;
        AtomTable t = parse_structure_gemmi(d, n, title);
        const size_t na = t.size();
        std::string a, r, c;
        for (size_t i = 0; i < na; i++) { a += t.name(t.atom[i]); a.push_back('\0'); r += t.name(t.residue[i]); r.push_back('\0'); c += t.chain_name(i); c.push_back('\0'); }
        auto dup = [](const void* p, size_t bytes) { void* q = malloc(bytes ? bytes : 1); if (bytes) memcpy(q, p, bytes); return q; };
        out->n = na;
        out->atom = (char*)dup(a.data(), a.size()); out->atom_bytes = a.size();
        out->residue = (char*)dup(r.data(), r.size()); out->residue_bytes = r.size();
        out->chain = (char*)dup(c.data(), c.size()); out->chain_bytes = c.size();
        out->atom_index = (int32_t*)dup(t.atom_index.data(), 4 * na); out->res_index = (int32_t*)dup(t.res_index.data(), 4 * na);
        out->x = (float*)dup(t.x.data(), 4 * na); out->y = (float*)dup(t.y.data(), 4 * na); out->z = (float*)dup(t.z.data(), 4 * na);
        out->bfac = (float*)dup(t.bfac.data(), 4 * na);
        out->title = (char*)dup(title.data(), title.size()); out->title_len = title.size();
        table_pool().put(std::move(t));
        return 0;
    } catch (const std::exception& e) {
        snprintf(out->error, sizeof out->error, "%s", e.what());
        return 1;
    }
}
void fcz_host_free(fcz_host_atoms* o) {
    free(o->atom); free(o->residue); free(o->chain); free(o->atom_index); free(o->res_index); free(o->x); free(o->y); free(o->z); free(o->bfac); free(o->title);
    memset(o, 0, sizeof *o);
}
}
