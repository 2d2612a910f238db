// fcz_abi.hip -- C-ABI (include/fcz_hip.h) over the gfx950 kernels. No torch, no C++ types cross the ABI.
// There is deliberately NO CPU fallback in this library: without a HIP device every compute entry
// point returns FCZ_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "fcz_kernels.h"
#include "fcz_compress.h"
// persistent grids: blocks per resident slot. Measured (1 M chains): 1 block per slot is 5 % slower than 4, 16 is 3-5 % faster
// than 4 (k_compress_angles_w 17.2 -> 16.4 ms, k_sidechain 16.0 -> 15.5 ms), 64 no better: with more, shorter blocks the
// hardware dispatcher evens out what the CUs finish at different times, and the table prologue is still paid once per ~30+ tiles
#ifndef FCZ_CW_GRID_FACTOR
#define FCZ_CW_GRID_FACTOR 16u
#endif
#ifndef FCZ_SC_GRID_FACTOR
#define FCZ_SC_GRID_FACTOR 16u
#endif
#include "fcz_sidechain.h"
#include "fcz_backbone_fast.h"
#include "fcz_pdb.h"
#include "fcz_extract.h"
#include "fcz_ingest.h"
#include "fcz_ingest_cif.h"
#include "fcz_inflate.h"

// second, host-side instance of the generated tables (integer metadata for sizes/validation)
namespace host_tab {
#undef FCZ_TABLE_QUAL
#undef FCZ_T
#define FCZ_TABLE_QUAL static const
#define FCZ_T(name) h_##name
#include "aa_tables.inc"
}  // namespace host_tab

using namespace fcz;

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            fprintf(stderr, "fcz_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return FCZ_E_HIP;                                                             \
        }                                                                                 \
    } while (0)

namespace {

struct dev_buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return FCZ_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return FCZ_E_NOMEM; }
        cap = want;
        return FCZ_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

struct timed_span { std::string name; hipEvent_t a, b; };

}  // namespace

struct fcz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // scratch
    dev_buf ang;        // compress: 6 x R floats
    dev_buf sizes;      // compress: C x u64
    dev_buf scan_tmp;   // block partials of the device scans
    dev_buf codes;      // decompress: residue codes, one byte per residue at (record offset >> 3) + k (k_entry_sizes -> k_res_index)
    dev_buf res_sc_addr; // compress: residue -> output byte offset of its side-chain torsion bytes
    dev_buf tile_work;   // compress: per 256-residue tile flag + list + count of the tiles left to the block-tile kernel
    // decompress: the totals and the length order computed by fcz_decompress_sizes_dev are reused by the
    // fcz_decompress_batch_dev call that follows on the same entries
    const void* sized_blob = nullptr; const void* sized_off = nullptr; uint32_t sized_n = 0, sized_R = 0, sized_maxseg = 0, sized_maxnseg = 0, sized_nlong = 0;
    bool sizes_fresh = false;
    dev_buf cnt;        // decompress: 2 x n u32 counts (residues, atoms) + n i32 status + n u32 segment info
    dev_buf fwd;        // decompress: per-group ring of forward atoms (one segment deep)
    dev_buf wring;      // decompress: per-group ring of cos/sin of the segment's torsions
    dev_buf fwd_long, wring_long;   // decompress: the same per (group, segment) for the long chains' split form
    dev_buf bb;         // decompress: blended backbone
    dev_buf len_perm;   // decompress: entries ordered by residue count (n u32) + bucket counters (2 x LEN_BUCKETS + 1)
    dev_buf res_aoff;   // decompress: residue -> first output atom
    dev_buf res_rc;     // decompress: residue -> residue code
    dev_buf res_sc;     // decompress: residue -> its side-chain torsion bytes, 3 x R dwords
    // staging for the host-pointer entry points
    dev_buf stage[20];
    dev_buf pdb_size, pdb_off, pdb_text;   // PDB text: per-entry sizes, offsets (n+1 u64), the text of the last begin() call
    uint64_t pdb_bytes = 0;
    uint32_t* pinned = nullptr;  // 16 words
    hipStream_t stream2 = nullptr;   // long chains of a decompress batch run beside the rest
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int n_cu = 256;
    bool timing = false;
    bool keep_first_angle = false;
    int numerics = FCZ_NUMERICS_EXACT;
    // profiling aid (tools/hbm_busy_probe.py): FCZ_PROFILE_STAGES=<mask> in the environment (read at every decompress batch call)
    // leaves out stages of the call so that ONE kernel fills the device for seconds (1 backbone, 2 residue index, 4 side chains;
    // unset = all). The outputs are then stale by construction: never set outside a profiling run.
    unsigned profile_stages = 7u;
    dev_buf fast_scratch;   // decompress, FCZ_NUMERICS_FAST: forward atoms of segments longer than one chunk
    // structure ingest: scratch atom table, per-file lists, the resident batch of the last ingest call
    dev_buf ig[40];
    fcz_ingest_result ig_res{};
    uint32_t ig_counts[5] = {0, 0, 0, 0, 0};
    uint64_t ig_fcz_bytes = 0;
    // inflate in front of the ingest: the files' bytes as they came over the link, their offsets / kinds / text offsets / statuses
    dev_buf gz_raw, gz_off, gz_kind, gz_toff, gz_status;
    std::vector<timed_span> spans;
    std::map<std::string, std::pair<double, uint64_t>> acc;
};

namespace {

struct span_guard {
    fcz_ctx* ctx; hipEvent_t a = nullptr, b = nullptr; const char* name;
    span_guard(fcz_ctx* c, const char* n) : ctx(c), name(n) {
        if (ctx->timing) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, ctx->stream); }
    }
    ~span_guard() {
        if (ctx->timing) { (void)hipEventRecord(b, ctx->stream); ctx->spans.push_back({name, a, b}); }
    }
};

void drain_spans(fcz_ctx* ctx) {
    if (ctx->spans.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { auto& e = ctx->acc[s.name]; e.first += ms; e.second += 1; }
        (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
    }
    ctx->spans.clear();
}

inline unsigned grid_for(uint32_t items, unsigned per_block) { return (items + per_block - 1) / per_block; }

// exclusive scan of n elements into out[n+1] on the ctx stream (three launches, any n)
// overflow (may be null): set to 1 on the device when the total does not fit T
template <class T>
int device_scan(fcz_ctx* ctx, const T* in, T* out, uint32_t n, uint32_t* overflow = nullptr) {
    if (n == 0) { if (hipMemsetAsync(out, 0, sizeof(T), ctx->stream) != hipSuccess) return FCZ_E_HIP; return FCZ_OK; }
    const unsigned nb = grid_for(n, SCAN_CHUNK);
    int rc = ctx->scan_tmp.ensure(sizeof(unsigned long long) * 2 * ((size_t)nb + 1));
    if (rc) return rc;
    unsigned long long* part = ctx->scan_tmp.as<unsigned long long>();
    unsigned long long* part_ex = part + nb + 1;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_reduce<T>), dim3(nb), dim3(1024), 0, ctx->stream, n, in, part);
    hipLaunchKernelGGL(k_scan_u64, dim3(1), dim3(1024), 0, ctx->stream, nb, (const uint64_t*)part, (uint64_t*)part_ex);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<T>), dim3(nb), dim3(1024), 0, ctx->stream, n, in, part_ex, out, overflow);
    return FCZ_OK;
}

}  // namespace

extern "C" {

const char* fcz_status_string(int s) {
    switch (s) {
        case FCZ_OK: return "ok";
        case FCZ_E_INVALID_ARG: return "invalid argument";
        case FCZ_E_NO_DEVICE: return "no HIP device (libfcz_hip has no CPU fallback)";
        case FCZ_E_HIP: return "HIP runtime error";
        case FCZ_E_BAD_MAGIC: return "not an FCZ entry (bad magic)";
        case FCZ_E_TRUNCATED: return "truncated or inconsistent FCZ entry";
        case FCZ_E_RESIDUE: return "residue code not supported by the codec";
        case FCZ_E_TOO_SHORT: return "chain shorter than 2 residues";
        case FCZ_E_NOMEM: return "out of device memory";
        case FCZ_E_NONFINITE: return "a coordinate or B-factor of the chain is not a finite number";
        default: return "unknown status";
    }
}

const char* fcz_atom_code_name(int code) {
    if (code < 0 || code >= FCZ_N_ATOM_CODES) return nullptr;
    return host_tab::h_atom_name[code];
}
int fcz_atom_code_from_name(const char* name) {
    for (int i = 0; i < FCZ_N_ATOM_CODES; i++) if (strcmp(host_tab::h_atom_name[i], name) == 0) return i;
    return FCZ_ATOM_CODE_OTHER;
}
int fcz_res_code_from_name(const char* n3) {
    for (int i = 0; i < FCZ_N_RES_CODES; i++)
        if (strcmp(host_tab::h_res3[i], n3) == 0) return (i < 20 || i == 23) ? i : -1;
    return -1;
}
const char* fcz_res_code_name(int rc) { return (rc >= 0 && rc < FCZ_N_RES_CODES) ? host_tab::h_res3[rc] : "UNK"; }
int fcz_res_code_natoms(int rc) { return (rc >= 0 && rc < FCZ_N_RES_CODES) ? host_tab::h_res_natoms[rc] : 3; }
int fcz_res_code_atom(int rc, int j, int alt) {
    if (rc < 0 || rc >= FCZ_N_RES_CODES) rc = 23;
    if (j < 0 || j >= host_tab::h_res_natoms[rc]) return FCZ_ATOM_CODE_OTHER;
    int slot = alt ? host_tab::h_res_alt_slot[rc][j] : j;
    return host_tab::h_res_atom[rc][slot];
}

int fcz_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
void* fcz_pinned_alloc(size_t bytes) { void* p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void fcz_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

void fcz_ctx_destroy(fcz_ctx* c);
int fcz_ctx_create(int device, fcz_ctx** out) {
    if (!out) return FCZ_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FCZ_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FCZ_E_NO_DEVICE;
    fcz_ctx* c = new fcz_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    c->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return FCZ_E_HIP; }
    if (hipHostMalloc((void**)&c->pinned, 128, hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return FCZ_E_HIP; }
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { fcz_ctx_destroy(c); return FCZ_E_HIP; }
    *out = c;
    return FCZ_OK;
}

void fcz_ctx_destroy(fcz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drain_spans(c);
    (void)hipStreamSynchronize(c->stream);
    c->ang.release(); c->res_sc_addr.release(); c->tile_work.release(); c->sizes.release(); c->scan_tmp.release(); c->codes.release(); c->cnt.release(); c->fwd.release(); c->bb.release(); c->wring.release(); c->fwd_long.release(); c->wring_long.release(); c->res_aoff.release(); c->len_perm.release(); c->pdb_size.release(); c->pdb_off.release(); c->pdb_text.release(); c->res_rc.release(); c->res_sc.release(); c->fast_scratch.release();
    for (auto& b : c->stage) b.release();
    for (auto& b : c->ig) b.release();
    c->gz_raw.release(); c->gz_off.release(); c->gz_kind.release(); c->gz_toff.release(); c->gz_status.release();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

void* fcz_ctx_stream(fcz_ctx* c) { return c ? (void*)c->stream : nullptr; }
int fcz_ctx_set_numerics(fcz_ctx* c, int mode) {
    if (!c || (mode != FCZ_NUMERICS_EXACT && mode != FCZ_NUMERICS_FAST)) return FCZ_E_INVALID_ARG;
    c->numerics = mode;
    return FCZ_OK;
}
int fcz_ctx_get_numerics(fcz_ctx* c) { return c ? c->numerics : FCZ_E_INVALID_ARG; }
int fcz_ctx_synchronize(fcz_ctx* c) {
    if (!c) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FCZ_OK;
}
// ------------------------------------------------------------------------------------------------
// PDB text of decompressed chains (writeAtomCoordinatesToPDB, reference src/atom_coordinate.cpp:220-291)
// ------------------------------------------------------------------------------------------------
static int pdb_sizes_impl(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, const uint32_t* res_off_dev,
                          const uint32_t* atom_off_dev, const fcz_atoms_out* atoms_dev, uint32_t pad, uint64_t* text_off_dev) {
    if (!ctx || !blob_dev || !off_dev || !res_off_dev || !atom_off_dev || !atoms_dev || !text_off_dev) return FCZ_E_INVALID_ARG;
    if (!atoms_dev->x || !atoms_dev->y || !atoms_dev->z || !atoms_dev->bfac_res || !atoms_dev->res_code) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ctx->pdb_size.ensure(sizeof(uint64_t) * (size_t)std::max<uint32_t>(n, 1)); if (rc) return rc;
    span_guard g(ctx, "pdb_sizes");
    if (n) hipLaunchKernelGGL(k_pdb_sizes, dim3(grid_for(n, WAVES_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n,
                              res_off_dev, atom_off_dev, *atoms_dev, pad, ctx->pdb_size.as<uint64_t>());
    if ((rc = device_scan<uint64_t>(ctx, ctx->pdb_size.as<uint64_t>(), text_off_dev, n))) return rc;
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}
int fcz_pdb_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, const uint32_t* res_off_dev,
                      const uint32_t* atom_off_dev, const fcz_atoms_out* atoms_dev, uint64_t* text_off_dev) {
    return pdb_sizes_impl(ctx, blob_dev, off_dev, n, res_off_dev, atom_off_dev, atoms_dev, 0, text_off_dev);
}

int fcz_pdb_format_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, const uint32_t* res_off_dev,
                       const uint32_t* atom_off_dev, const fcz_atoms_out* atoms_dev, int alt_order, const uint64_t* text_off_dev,
                       uint8_t* text_dev) {
    if (!ctx || !blob_dev || !off_dev || !res_off_dev || !atom_off_dev || !atoms_dev || !text_off_dev || !text_dev) return FCZ_E_INVALID_ARG;
    if (!atoms_dev->x || !atoms_dev->y || !atoms_dev->z || !atoms_dev->bfac_res || !atoms_dev->res_code) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return FCZ_OK;
    span_guard g(ctx, "pdb_format");
    hipLaunchKernelGGL(k_pdb_format, dim3(grid_for(n, WAVES_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n, res_off_dev,
                       atom_off_dev, *atoms_dev, alt_order, text_off_dev, text_dev);
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

// Host-pointer convenience: FCZ entries in, PDB text out, everything in between on the device. begin() leaves the text in
// the ctx and reports the per-entry text offsets; fetch() copies it out.
static int pdb_begin_impl(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int alt_order, uint64_t* text_off,
                          int32_t* status, bool format) {
    if (!ctx || !blob || !off || !text_off) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t pad = (alt_order & FCZ_PDB_NUL_TERMINATED) ? 1u : 0u;   // every entry followed by one NUL (a database record)
    alt_order &= FCZ_PDB_ALT_ORDER;
    ctx->pdb_bytes = 0;
    ctx->sizes_fresh = false;   // the staging buffers the cache is keyed on are about to be rewritten
    if (n == 0) { text_off[0] = 0; return FCZ_OK; }
    const uint64_t blob_bytes = off[n];
    int rc;
    if ((rc = ctx->stage[0].ensure(std::max<uint64_t>(blob_bytes, 16)))) return rc;
    if ((rc = ctx->stage[1].ensure(sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->stage[2].ensure(sizeof(uint32_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->stage[3].ensure(sizeof(uint32_t) * ((size_t)n + 1)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->stage[0].p, blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[1].p, off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    uint32_t R = 0, M = 0;
    rc = fcz_decompress_sizes_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, ctx->stage[2].as<uint32_t>(),
                                  ctx->stage[3].as<uint32_t>(), &R, &M);
    if (rc) return rc;
    if (status) {   // per-entry status of the sizes pass (cnt layout: 2 x n counts, then n status words)
        HIP_TRY(hipMemcpyAsync(status, ctx->cnt.as<uint32_t>() + 2 * (size_t)n, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    for (int i = 4; i < 7; i++) if ((rc = ctx->stage[i].ensure(std::max<size_t>(sizeof(float) * (size_t)M, 16)))) return rc;
    if ((rc = ctx->stage[7].ensure(std::max<size_t>(sizeof(float) * (size_t)R, 16)))) return rc;
    if ((rc = ctx->stage[8].ensure(std::max<size_t>((size_t)R, 16)))) return rc;
    fcz_atoms_out dv;
    dv.x = ctx->stage[4].as<float>(); dv.y = ctx->stage[5].as<float>(); dv.z = ctx->stage[6].as<float>();
    dv.bfac_res = ctx->stage[7].as<float>(); dv.res_code = ctx->stage[8].as<uint8_t>(); dv.atom_code = nullptr;
    if ((rc = ctx->pdb_off.ensure(sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if (R) {
        rc = fcz_decompress_batch_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, ctx->stage[2].as<uint32_t>(),
                                      ctx->stage[3].as<uint32_t>(), alt_order, &dv);
        if (rc) return rc;
    }
    ctx->sizes_fresh = false;
    rc = pdb_sizes_impl(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, ctx->stage[2].as<uint32_t>(),
                        ctx->stage[3].as<uint32_t>(), &dv, pad, ctx->pdb_off.as<uint64_t>());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(text_off, ctx->pdb_off.p, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!format) return FCZ_OK;           // sizes only: nothing is kept for a fetch
    ctx->pdb_bytes = text_off[n];
    if ((rc = ctx->pdb_text.ensure(std::max<uint64_t>(ctx->pdb_bytes, 16)))) return rc;
    if (pad && ctx->pdb_bytes) HIP_TRY(hipMemsetAsync(ctx->pdb_text.p, 0, ctx->pdb_bytes, ctx->stream));   // the terminators: the format pass writes the text around them
    return fcz_pdb_format_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, ctx->stage[2].as<uint32_t>(),
                              ctx->stage[3].as<uint32_t>(), &dv, alt_order, ctx->pdb_off.as<uint64_t>(), ctx->pdb_text.as<uint8_t>());
}

int fcz_decompress_pdb_begin(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int alt_order, uint64_t* text_off,
                             int32_t* status) {
    return pdb_begin_impl(ctx, blob, off, n, alt_order, text_off, status, true);
}

int fcz_decompress_pdb_sizes(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int alt_order, uint64_t* text_off,
                             int32_t* status) {
    return pdb_begin_impl(ctx, blob, off, n, alt_order, text_off, status, false);
}

int fcz_decompress_pdb_fetch(fcz_ctx* ctx, uint8_t* text_out) {
    if (!ctx || (!text_out && ctx->pdb_bytes)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->pdb_bytes) HIP_TRY(hipMemcpyAsync(text_out, ctx->pdb_text.p, ctx->pdb_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

// ------------------------------------------------------------------------------------------------
// extract (Foldcomp::extract, reference src/foldcomp.cpp:1260-1336)
// ------------------------------------------------------------------------------------------------
static inline int extract_args_ok(int mode, int digits) { return (mode == 0 && digits >= 1 && digits <= 4) || mode == 1; }

int fcz_extract_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n, int mode, int digits, uint64_t* data_off) {
    if (!blob || !off || !data_off || !extract_args_ok(mode, digits)) return FCZ_E_INVALID_ARG;
    data_off[0] = 0;
    for (uint32_t i = 0; i < n; i++)
        data_off[i + 1] = data_off[i] + extract_bytes(extract_entry_residues(blob + off[i], off[i + 1] - off[i]), mode, digits);
    return FCZ_OK;
}

int fcz_extract_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, int mode, int digits,
                          uint64_t* data_off_dev) {
    if (!ctx || !blob_dev || !off_dev || !data_off_dev || !extract_args_ok(mode, digits)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ctx->pdb_size.ensure(sizeof(uint64_t) * (size_t)std::max<uint32_t>(n, 1)); if (rc) return rc;
    span_guard g(ctx, "extract_sizes");
    if (n) hipLaunchKernelGGL(k_extract_sizes, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n, mode, digits,
                              ctx->pdb_size.as<uint64_t>());
    if ((rc = device_scan<uint64_t>(ctx, ctx->pdb_size.as<uint64_t>(), data_off_dev, n))) return rc;
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

int fcz_extract_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, int mode, int digits,
                    const uint64_t* data_off_dev, uint8_t* data_dev) {
    if (!ctx || !blob_dev || !off_dev || !data_off_dev || !data_dev || !extract_args_ok(mode, digits)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return FCZ_OK;
    span_guard g(ctx, "extract");
    hipLaunchKernelGGL(k_extract, dim3(grid_for(n, WAVES_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n, mode, digits,
                       data_off_dev, data_dev);
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

int fcz_extract(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int mode, int digits, const uint64_t* data_off,
                uint8_t* data_out) {
    if (!ctx || !blob || !off || !data_off || !extract_args_ok(mode, digits)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0 || data_off[n] == 0) return FCZ_OK;
    if (!data_out) return FCZ_E_INVALID_ARG;
    ctx->sizes_fresh = false;   // staging buffers are rewritten
    const uint64_t blob_bytes = off[n], data_bytes = data_off[n];
    int rc;
    if ((rc = ctx->stage[0].ensure(std::max<uint64_t>(blob_bytes, 16)))) return rc;
    if ((rc = ctx->stage[1].ensure(sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->pdb_off.ensure(sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->pdb_text.ensure(std::max<uint64_t>(data_bytes, 16)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->stage[0].p, blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[1].p, off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->pdb_off.p, data_off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    rc = fcz_extract_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, mode, digits, ctx->pdb_off.as<uint64_t>(),
                         ctx->pdb_text.as<uint8_t>());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(data_out, ctx->pdb_text.p, data_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

int fcz_ctx_enable_timing(fcz_ctx* c, int enable) { if (!c) return FCZ_E_INVALID_ARG; drain_spans(c); c->timing = enable != 0; return FCZ_OK; }
void fcz_ctx_reset_timing(fcz_ctx* c) { if (!c) return; drain_spans(c); c->acc.clear(); }
int fcz_ctx_kernel_time(fcz_ctx* c, const char* name, double* ms, uint64_t* launches) {
    if (!c || !name) return FCZ_E_INVALID_ARG;
    drain_spans(c);
    auto it = c->acc.find(name);
    if (ms) *ms = it == c->acc.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == c->acc.end() ? 0 : it->second.second;
    return FCZ_OK;
}

// ------------------------------------------------------------------------------------------------
// compress
// ------------------------------------------------------------------------------------------------
int fcz_compress_sizes(const fcz_chain_batch* in, uint64_t* out_off) {
    if (!in || !out_off || in->anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    uint64_t o = 0;
    for (uint32_t c = 0; c < in->n_chains; c++) {
        out_off[c] = o;
        const uint32_t r0 = in->res_off[c], n = in->res_off[c + 1] - r0;
        uint32_t nsc = 0;
        for (uint32_t k = 0; k < n; k++) { uint32_t rc = in->res_code[r0 + k]; nsc += host_tab::h_res_natoms[rc < 24 ? rc : 23] - 3; }
        o += make_layout(n, n / (uint32_t)in->anchor_threshold + 2, in->title_off[c + 1] - in->title_off[c], nsc).size;
    }
    out_off[in->n_chains] = o;
    return FCZ_OK;
}

int fcz_compress_sizes_dev(fcz_ctx* ctx, const fcz_chain_batch* in, uint64_t* out_off_dev) {
    if (!ctx || !in || !out_off_dev || in->anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (in->n_chains == 0) { HIP_TRY(hipMemsetAsync(out_off_dev, 0, sizeof(uint64_t), ctx->stream)); return FCZ_OK; }
    int rc = ctx->sizes.ensure(sizeof(uint64_t) * (size_t)in->n_chains);
    if (rc) return rc;
    span_guard g(ctx, "compress_sizes");
    hipLaunchKernelGGL(k_compress_sizes, dim3(grid_for(in->n_chains, GROUPS_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, *in, ctx->sizes.as<uint64_t>());
    rc = device_scan<uint64_t>(ctx, ctx->sizes.as<uint64_t>(), out_off_dev, in->n_chains);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

int fcz_compress_batch_dev(fcz_ctx* ctx, const fcz_chain_batch* in, const uint64_t* out_off_dev, uint8_t* out_dev,
                           int32_t* status_dev) {
    if (!ctx || !in || !out_off_dev || !out_dev) return FCZ_E_INVALID_ARG;
    if (in->anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (in->n_chains == 0) return FCZ_OK;
    int rc = ctx->ang.ensure(sizeof(float) * 6 * (size_t)std::max<uint32_t>(in->n_residues, 1));
    if (rc) return rc;
    rc = ctx->res_sc_addr.ensure(sizeof(uint64_t) * (size_t)std::max<uint32_t>(in->n_residues, 1));
    if (rc) return rc;
    const dim3 per_chain(grid_for(in->n_chains, WAVES_PER_BLOCK));
    const size_t nf_words = ((size_t)in->n_chains + 31) / 32;
    uint32_t* nonfinite = nullptr;
    {
        span_guard g(ctx, "compress_index");
        hipLaunchKernelGGL(k_compress_index, dim3(grid_for(in->n_chains, GROUPS_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, ctx->res_sc_addr.as<uint64_t>());
    }
    if (in->n_residues) {
        // wavefront-private tiles first; what does not fit them (atom-rich stretches, the tail of the arrays) is listed per
        // 256-residue tile and taken by the block-tile kernel, which handles every special case
        span_guard g(ctx, "compress_angles");
        const uint32_t n_tiles = grid_for(in->n_residues, CK_TILE);
        const uint32_t n_wtiles = grid_for(in->n_residues, CW_RES);
        rc = ctx->tile_work.ensure(sizeof(uint32_t) * (2 * (size_t)n_tiles + 4 + nf_words)); if (rc) return rc;
        uint32_t* flags = ctx->tile_work.as<uint32_t>(); uint32_t* list = flags + n_tiles; uint32_t* count = list + n_tiles;
        nonfinite = count + 4;             // one bit per chain: a named atom with a NaN / infinite coordinate (set by the angle kernels)
        HIP_TRY(hipMemsetAsync(flags, 0, sizeof(uint32_t) * (2 * (size_t)n_tiles + 4 + nf_words), ctx->stream));
        const uint32_t blocks_w = std::min<uint32_t>(grid_for(n_wtiles, WAVES_PER_BLOCK), (uint32_t)ctx->n_cu * 3u * FCZ_CW_GRID_FACTOR);
        hipLaunchKernelGGL(k_compress_angles_w, dim3(blocks_w), dim3(BLOCK), 0, ctx->stream, *in, n_wtiles, ctx->res_sc_addr.as<uint64_t>(), out_dev,
                           ctx->ang.as<float>(), flags, list, count, nonfinite);
        const uint32_t blocks = std::min<uint32_t>(n_tiles, (uint32_t)ctx->n_cu * FCZ_COMPRESS_MIN_BLOCKS);
        hipLaunchKernelGGL(k_compress_angles, dim3(blocks), dim3(BLOCK), 0, ctx->stream, *in, n_tiles, (const uint32_t*)list, (const uint32_t*)count,
                           ctx->res_sc_addr.as<uint64_t>(), out_dev, ctx->ang.as<float>(), nonfinite);
    } else {
        rc = ctx->tile_work.ensure(sizeof(uint32_t) * (4 + nf_words)); if (rc) return rc;
        nonfinite = ctx->tile_work.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(nonfinite, 0, sizeof(uint32_t) * nf_words, ctx->stream));
    }
    {
        span_guard g(ctx, "compress_pack");
        hipLaunchKernelGGL(k_compress_pack, per_chain, dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, out_dev, status_dev,
                           ctx->ang.as<float>(), ctx->keep_first_angle ? 1 : 0, (const uint32_t*)nonfinite);
        // chains of 2 .. 128 residues (k_compress_pack leaves them): four to a wavefront, a persistent grid over chunks of 16 chains
        // (one launch per length class -- 2..16, 17..32, 33..64, 65..128 residues in 1, 2, 4, 8 rounds of 16 -- each a scan of the chunks' lengths)
        const uint32_t rows_blocks = std::min<uint32_t>(grid_for(grid_for(in->n_chains, CP_CHUNK), WAVES_PER_BLOCK), (uint32_t)ctx->n_cu * 4u);
        // (every class launch is tied to the bound k_compress_pack skips by: a build with fewer rounds must not run a class twice)
        static_assert(FCZ_PACK_ROWS_MAX_ROUNDS == 1 || FCZ_PACK_ROWS_MAX_ROUNDS == 2 || FCZ_PACK_ROWS_MAX_ROUNDS == 4 || FCZ_PACK_ROWS_MAX_ROUNDS == 8,
                      "k_compress_pack_rows has the classes of 1, 2, 4 and 8 rounds");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compress_pack_rows<1>), dim3(rows_blocks), dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, out_dev, status_dev,
                           ctx->ang.as<float>(), ctx->keep_first_angle ? 1 : 0, (const uint32_t*)nonfinite);
#if FCZ_PACK_ROWS_MAX_ROUNDS >= 2
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compress_pack_rows<2>), dim3(rows_blocks), dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, out_dev, status_dev,
                           ctx->ang.as<float>(), ctx->keep_first_angle ? 1 : 0, (const uint32_t*)nonfinite);
#endif
#if FCZ_PACK_ROWS_MAX_ROUNDS >= 4
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compress_pack_rows<4>), dim3(rows_blocks), dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, out_dev, status_dev,
                           ctx->ang.as<float>(), ctx->keep_first_angle ? 1 : 0, (const uint32_t*)nonfinite);
#endif
#if FCZ_PACK_ROWS_MAX_ROUNDS >= 8
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compress_pack_rows<8>), dim3(rows_blocks), dim3(BLOCK), 0, ctx->stream, *in, out_off_dev, out_dev, status_dev,
                           ctx->ang.as<float>(), ctx->keep_first_angle ? 1 : 0, (const uint32_t*)nonfinite);
#endif
    }
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

int fcz_compress_batch(fcz_ctx* ctx, const fcz_chain_batch* in, const uint64_t* out_off, uint8_t* out, int32_t* status) {
    if (!ctx || !in || !out_off || !out) return FCZ_E_INVALID_ARG;
    if (in->anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t C = in->n_chains, R = in->n_residues, M = in->n_atoms;
    if (C == 0) return FCZ_OK;
    const uint32_t TL = in->title_off[C];
    const uint64_t out_bytes = out_off[C];
    // the side-chain byte addresses handed from k_compress_index to the angle kernels keep 39 bits (fcz_compress.h, sc_addr_put):
    // a blob of 512 GB or more is refused, not truncated (no device holds one: a device-resident out_dev cannot reach the limit)
    if (out_bytes >= (1ull << 39)) return FCZ_E_INVALID_ARG;
    struct item { const void* src; size_t bytes; };
    const item items[] = {
        {in->res_off, sizeof(uint32_t) * (C + 1)}, {in->atom_off, sizeof(uint32_t) * (R + 1)},
        {in->x, sizeof(float) * M}, {in->y, sizeof(float) * M}, {in->z, sizeof(float) * M},
        {in->atom_code, (size_t)M}, {in->res_code, (size_t)R}, {in->bfac_ca, sizeof(float) * R},
        {in->first_res_index, sizeof(int32_t) * C}, {in->first_atom_index, sizeof(int32_t) * C},
        {in->chain_id, (size_t)C}, {in->titles, (size_t)TL}, {in->title_off, sizeof(uint32_t) * (C + 1)},
        {out_off, sizeof(uint64_t) * (C + 1)},
    };
    void* d[14];
    for (int i = 0; i < 14; i++) {
        int rc = ctx->stage[i].ensure(std::max<size_t>(items[i].bytes, 16));
        if (rc) return rc;
        d[i] = ctx->stage[i].p;
        if (items[i].bytes) HIP_TRY(hipMemcpyAsync(d[i], items[i].src, items[i].bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = ctx->stage[14].ensure(std::max<uint64_t>(out_bytes, 16)); if (rc) return rc;
    rc = ctx->stage[15].ensure(sizeof(int32_t) * C); if (rc) return rc;
    fcz_chain_batch dv = *in;
    dv.res_off = (const uint32_t*)d[0]; dv.atom_off = (const uint32_t*)d[1];
    dv.x = (const float*)d[2]; dv.y = (const float*)d[3]; dv.z = (const float*)d[4];
    dv.atom_code = (const uint8_t*)d[5]; dv.res_code = (const uint8_t*)d[6]; dv.bfac_ca = (const float*)d[7];
    dv.first_res_index = (const int32_t*)d[8]; dv.first_atom_index = (const int32_t*)d[9];
    dv.chain_id = (const char*)d[10]; dv.titles = (const char*)d[11]; dv.title_off = (const uint32_t*)d[12];
    rc = fcz_compress_batch_dev(ctx, &dv, (const uint64_t*)d[13], ctx->stage[14].as<uint8_t>(), ctx->stage[15].as<int32_t>());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->stage[14].p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> st_host;
    int32_t* st = status;
    if (!st) { st_host.resize(C); st = st_host.data(); }
    HIP_TRY(hipMemcpyAsync(st, ctx->stage[15].p, sizeof(int32_t) * C, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int worst = FCZ_OK;
    for (uint32_t c = 0; c < C; c++) if (st[c] != FCZ_OK) worst = st[c];
    return worst;
}

// Pre-quantisation backbone angles of a host batch (what get_data() of the Python module reports for
// PDB input, foldcomp/foldcomp.cxx:633-662): angles_out is [6][R] floats in the order phi, psi, omega,
// n_ca_c, ca_c_n, c_n_ca; entry r0+k (k < n-1) belongs to packed word k of the chain starting at residue
// r0; n_ca_c[r0+n-1] holds the first residue's N-CA-C angle that the FCZ format drops.
int fcz_compress_angles(fcz_ctx* ctx, const fcz_chain_batch* in, float* angles_out) {
    if (!ctx || !in || !angles_out) return FCZ_E_INVALID_ARG;
    std::vector<uint64_t> off((size_t)in->n_chains + 1);
    int rc = fcz_compress_sizes(in, off.data());
    if (rc) return rc;
    std::vector<uint8_t> out(off[in->n_chains] ? off[in->n_chains] : 1);
    ctx->keep_first_angle = true;
    rc = fcz_compress_batch(ctx, in, off.data(), out.data(), nullptr);
    ctx->keep_first_angle = false;
    if (rc) return rc;
    HIP_TRY(hipMemcpy(angles_out, ctx->ang.p, sizeof(float) * 6 * (size_t)in->n_residues, hipMemcpyDeviceToHost));
    return FCZ_OK;
}


// ------------------------------------------------------------------------------------------------
// structure ingest: PDB text -> fcz_chain_batch on the device (fcz_ingest.h)
// ------------------------------------------------------------------------------------------------
int fcz_ingest_pdb_dev(fcz_ctx* ctx, const uint8_t* text_dev, const uint64_t* file_off_dev, uint32_t n_files, uint64_t text_bytes,
                       const char* names_dev, const uint32_t* name_off_dev, const uint32_t* stem_len_dev, int anchor_threshold, int flags,
                       fcz_ingest_result* out) {
    if (!ctx || !out || anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    if (n_files && (!text_dev || !file_off_dev || !names_dev || !name_off_dev || !stem_len_dev)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    memset(out, 0, sizeof *out);
    memset(&ctx->ig_res, 0, sizeof ctx->ig_res);
    memset(ctx->ig_counts, 0, sizeof ctx->ig_counts);
    out->n_files = n_files;
    if (n_files == 0) return FCZ_OK;
    // every accepted ATOM record has >= 54 characters (+ its line end, except possibly the file's last line)
    const uint64_t cap64 = text_bytes / 54u + (uint64_t)n_files;
    if (cap64 >> 32) return FCZ_E_INVALID_ARG;                 // atom offsets are 32-bit: split the batch
    const size_t cap = (size_t)cap64, F = n_files;
    int rc;
    enum { B_CAP, B_ABASE, B_NAME, B_RESN, B_SERIAL, B_RESSEQ, B_X, B_Y, B_Z, B_B, B_CHAIN, B_ACODE, B_RCODE, B_RFIRST, B_RBFAC, B_RCODE2,
           B_TITLES, B_TLEN, B_NKEPT, B_STATUS, B_FRAGS, B_NFRAGS, B_TOTC, B_TOTR, B_TOTA, B_TOTT, B_USESTEM, B_OFFC, B_OFFR, B_OFFA, B_OFFT,
           B_REFUSED, B_NREF, B_OUT_A, B_OUT_R, B_OUT_C, B_OUT_T, B_CIFROWS };
    auto need = [&](int i, size_t bytes) { return ctx->ig[i].ensure(std::max<size_t>(bytes, 16)); };
    if ((rc = need(B_CAP, 8 * F)) || (rc = need(B_ABASE, 8 * (F + 1))) || (rc = need(B_NAME, 4 * cap)) || (rc = need(B_RESN, 4 * cap)) ||
        (rc = need(B_SERIAL, 4 * cap)) || (rc = need(B_RESSEQ, 4 * cap)) || (rc = need(B_X, 4 * cap)) || (rc = need(B_Y, 4 * cap)) ||
        (rc = need(B_Z, 4 * cap)) || (rc = need(B_B, 4 * cap)) || (rc = need(B_CHAIN, 4 * cap)) || (rc = need(B_ACODE, cap)) || (rc = need(B_RCODE, cap)) ||
        (rc = need(B_RFIRST, 4 * cap)) || (rc = need(B_RBFAC, 4 * cap)) || (rc = need(B_RCODE2, cap)) || (rc = need(B_TITLES, (size_t)IG_TITLE_CAP * F)) ||
        (rc = need(B_TLEN, 4 * F)) || (rc = need(B_NKEPT, 4 * F)) || (rc = need(B_STATUS, 4 * F)) || (rc = need(B_FRAGS, sizeof(ingest_frag) * IG_MAX_FRAGS * F)) ||
        (rc = need(B_NFRAGS, 4 * F)) || (rc = need(B_TOTC, 4 * F)) || (rc = need(B_TOTR, 4 * F)) || (rc = need(B_TOTA, 4 * F)) || (rc = need(B_TOTT, 4 * F)) ||
        (rc = need(B_USESTEM, 4 * F)) || (rc = need(B_OFFC, 4 * (F + 1))) || (rc = need(B_OFFR, 4 * (F + 1))) || (rc = need(B_OFFA, 4 * (F + 1))) ||
        (rc = need(B_OFFT, 4 * (F + 1))) || (rc = need(B_REFUSED, 8 * (size_t)IG_MAX_FRAGS * F)) || (rc = need(B_NREF, 16)) || (rc = need(B_CIFROWS, 4 * F)))
        return rc;
    auto P = [&](int i) { return ctx->ig[i].p; };
    ingest_scratch T;
    T.name = (uint32_t*)P(B_NAME); T.resn = (uint32_t*)P(B_RESN); T.serial = (int32_t*)P(B_SERIAL); T.resseq = (int32_t*)P(B_RESSEQ);
    T.x = (float*)P(B_X); T.y = (float*)P(B_Y); T.z = (float*)P(B_Z); T.b = (float*)P(B_B);
    T.chain = (uint32_t*)P(B_CHAIN); T.acode = (uint8_t*)P(B_ACODE); T.rcode = (int8_t*)P(B_RCODE);
    T.r_first = (uint32_t*)P(B_RFIRST); T.r_bfac = (float*)P(B_RBFAC); T.r_code = (uint8_t*)P(B_RCODE2);
    hipLaunchKernelGGL(k_ingest_caps, dim3(grid_for(n_files, 256)), dim3(256), 0, ctx->stream, file_off_dev, n_files, (uint64_t*)P(B_CAP));
    if ((rc = device_scan<uint64_t>(ctx, (const uint64_t*)P(B_CAP), (uint64_t*)P(B_ABASE), n_files))) return rc;
    {
        span_guard g(ctx, "ingest_parse");
        hipLaunchKernelGGL(k_ingest_parse, dim3(n_files), dim3(WAVE), 0, ctx->stream, text_dev, file_off_dev, n_files, text_bytes,
                           (const uint64_t*)P(B_ABASE), T, (uint8_t*)P(B_TITLES), (uint32_t*)P(B_TLEN), (uint32_t*)P(B_NKEPT), (int32_t*)P(B_STATUS));
    }
    {
        // mmCIF text: the files the PDB kernel left to the host because they open with `data_` (a wavefront of any other file returns
        // at once); what this kernel cannot promise to read as the reference's reader would stays handed back
        span_guard g(ctx, "ingest_parse_cif");
        hipLaunchKernelGGL(k_ingest_parse_cif, dim3(n_files), dim3(WAVE), 0, ctx->stream, text_dev, file_off_dev, n_files,
                           (const uint64_t*)P(B_ABASE), T, (uint8_t*)P(B_TITLES), (uint32_t*)P(B_TLEN), (const int32_t*)P(B_STATUS), (uint32_t*)P(B_CIFROWS));
    }
    {
        // ... and the rows it marked, read by a kernel of their own (fcz_ingest_cif.h: the register file)
        span_guard g(ctx, "ingest_rows_cif");
        hipLaunchKernelGGL(k_ingest_rows_cif, dim3(n_files), dim3(WAVE), 0, ctx->stream, text_dev, file_off_dev, n_files, text_bytes,
                           (const uint64_t*)P(B_ABASE), T, (uint32_t*)P(B_NKEPT), (int32_t*)P(B_STATUS), (const uint32_t*)P(B_CIFROWS));
    }
    ingest_counts tot{(uint32_t*)P(B_TOTC), (uint32_t*)P(B_TOTR), (uint32_t*)P(B_TOTA), (uint32_t*)P(B_TOTT)};
    {
        span_guard g(ctx, "ingest_frags");
        hipLaunchKernelGGL(k_ingest_frags, dim3(n_files), dim3(WAVE), 0, ctx->stream, n_files, (const uint64_t*)P(B_ABASE), T, (const uint32_t*)P(B_NKEPT),
                           (int32_t*)P(B_STATUS), (const uint32_t*)P(B_TLEN), names_dev, name_off_dev, stem_len_dev, (const uint8_t*)P(B_TITLES),
                           anchor_threshold, (flags & FCZ_INGEST_SKIP_DISCONTINUOUS) ? 1 : 0, (ingest_frag*)P(B_FRAGS), (uint32_t*)P(B_NFRAGS), tot,
                           (uint32_t*)P(B_USESTEM));
    }
    uint32_t* ovf = (uint32_t*)P(B_NREF) + 1;
    HIP_TRY(hipMemsetAsync(P(B_NREF), 0, 16, ctx->stream));
    if ((rc = device_scan<uint32_t>(ctx, tot.chains, (uint32_t*)P(B_OFFC), n_files, ovf)) || (rc = device_scan<uint32_t>(ctx, tot.residues, (uint32_t*)P(B_OFFR), n_files, ovf)) ||
        (rc = device_scan<uint32_t>(ctx, tot.atoms, (uint32_t*)P(B_OFFA), n_files, ovf)) || (rc = device_scan<uint32_t>(ctx, tot.title_bytes, (uint32_t*)P(B_OFFT), n_files, ovf)))
        return rc;
    uint32_t* pin = ctx->pinned + 8;    // [8..11] totals, [12] overflow
    HIP_TRY(hipMemcpyAsync(&pin[0], (uint32_t*)P(B_OFFC) + n_files, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&pin[1], (uint32_t*)P(B_OFFR) + n_files, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&pin[2], (uint32_t*)P(B_OFFA) + n_files, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&pin[3], (uint32_t*)P(B_OFFT) + n_files, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&pin[4], ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (pin[4]) return FCZ_E_INVALID_ARG;
    const uint32_t C = pin[0], R = pin[1], M = pin[2], TB = pin[3];
    // the batch arrays: [atoms] x y z code | [residues + 1] atom_off, [residues] code bfac | [chains (+1)] ... | titles
    const size_t oa_x = 0, oa_y = 4 * (size_t)M, oa_z = 8 * (size_t)M, oa_c = 12 * (size_t)M;
    const size_t or_off = 0, or_bf = 4 * ((size_t)R + 1), or_rc = or_bf + 4 * (size_t)R;
    const size_t oc_res = 0, oc_tit = 4 * ((size_t)C + 1), oc_fr = 2 * oc_tit, oc_fa = oc_fr + 4 * (size_t)C, oc_file = oc_fa + 4 * (size_t)C,
                 oc_meta = oc_file + 4 * (size_t)C, oc_name = oc_meta + 4 * (size_t)C, oc_id = oc_name + 4 * (size_t)C;
    if ((rc = need(B_OUT_A, 13 * (size_t)M + 16)) || (rc = need(B_OUT_R, or_rc + R + 16)) || (rc = need(B_OUT_C, oc_id + C + 16)) || (rc = need(B_OUT_T, (size_t)TB + 16))) return rc;
    char* ba = (char*)P(B_OUT_A); char* br = (char*)P(B_OUT_R); char* bc = (char*)P(B_OUT_C);
    ingest_out O;
    O.x = (float*)(ba + oa_x); O.y = (float*)(ba + oa_y); O.z = (float*)(ba + oa_z); O.atom_code = (uint8_t*)(ba + oa_c);
    O.atom_off = (uint32_t*)(br + or_off); O.bfac_ca = (float*)(br + or_bf); O.res_code = (uint8_t*)(br + or_rc);
    O.res_off = (uint32_t*)(bc + oc_res); O.title_off = (uint32_t*)(bc + oc_tit); O.first_res = (int32_t*)(bc + oc_fr); O.first_atom = (int32_t*)(bc + oc_fa);
    O.chain_file = (uint32_t*)(bc + oc_file); O.chain_meta = (uint32_t*)(bc + oc_meta); O.chain_name4 = (uint32_t*)(bc + oc_name); O.chain_id = bc + oc_id;
    O.titles = (char*)P(B_OUT_T);
    {
        span_guard g(ctx, "ingest_fill");
        hipLaunchKernelGGL(k_ingest_fill, dim3(n_files), dim3(WAVE), 0, ctx->stream, n_files, (const uint64_t*)P(B_ABASE), T, (const ingest_frag*)P(B_FRAGS),
                           (const uint32_t*)P(B_NFRAGS), (const uint32_t*)P(B_OFFC), (const uint32_t*)P(B_OFFR), (const uint32_t*)P(B_OFFA), (const uint32_t*)P(B_OFFT),
                           (const uint32_t*)P(B_TLEN), (const uint32_t*)P(B_USESTEM), names_dev, name_off_dev, stem_len_dev, (const uint8_t*)P(B_TITLES), O,
                           (uint32_t*)P(B_REFUSED), (uint32_t*)P(B_NREF));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&pin[5], P(B_NREF), 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    fcz_ingest_result& r = ctx->ig_res;
    r.batch.n_chains = C; r.batch.n_residues = R; r.batch.n_atoms = M; r.batch.anchor_threshold = anchor_threshold;
    r.batch.res_off = O.res_off; r.batch.atom_off = O.atom_off; r.batch.x = O.x; r.batch.y = O.y; r.batch.z = O.z;
    r.batch.atom_code = O.atom_code; r.batch.res_code = O.res_code; r.batch.bfac_ca = O.bfac_ca;
    r.batch.first_res_index = O.first_res; r.batch.first_atom_index = O.first_atom; r.batch.chain_id = O.chain_id;
    r.batch.titles = O.titles; r.batch.title_off = O.title_off;
    r.chain_file = O.chain_file; r.chain_meta = O.chain_meta; r.chain_name4 = O.chain_name4; r.file_status = (const int32_t*)P(B_STATUS); r.refused = (const uint32_t*)P(B_REFUSED);
    r.n_files = n_files; r.n_refused = pin[5];
    ctx->ig_counts[0] = C; ctx->ig_counts[1] = R; ctx->ig_counts[2] = M; ctx->ig_counts[3] = TB; ctx->ig_counts[4] = pin[5];
    *out = r;
    return FCZ_OK;
}

int fcz_ingest_pdb_begin(fcz_ctx* ctx, const uint8_t* text, const uint64_t* file_off, uint32_t n_files, const char* names, const uint32_t* name_off,
                         const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5]) {
    if (!ctx || !counts || anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    if (n_files && (!text || !file_off || !names || !name_off || !stem_len)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->sizes_fresh = false;   // staging buffers are rewritten
    memset(counts, 0, 5 * sizeof(uint32_t));
    if (n_files == 0) { memset(&ctx->ig_res, 0, sizeof ctx->ig_res); memset(ctx->ig_counts, 0, sizeof ctx->ig_counts); return FCZ_OK; }
    const uint64_t text_bytes = file_off[n_files];
    const uint32_t name_bytes = name_off[n_files];
    int rc;
    if ((rc = ctx->stage[0].ensure(std::max<uint64_t>(text_bytes, 16))) || (rc = ctx->stage[1].ensure(8 * ((size_t)n_files + 1))) ||
        (rc = ctx->stage[2].ensure(std::max<size_t>(name_bytes, 16))) || (rc = ctx->stage[3].ensure(4 * ((size_t)n_files + 1))) || (rc = ctx->stage[4].ensure(4 * (size_t)n_files)))
        return rc;
    HIP_TRY(hipMemcpyAsync(ctx->stage[0].p, text, text_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[1].p, file_off, 8 * ((size_t)n_files + 1), hipMemcpyHostToDevice, ctx->stream));
    if (name_bytes) HIP_TRY(hipMemcpyAsync(ctx->stage[2].p, names, name_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[3].p, name_off, 4 * ((size_t)n_files + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[4].p, stem_len, 4 * (size_t)n_files, hipMemcpyHostToDevice, ctx->stream));
    fcz_ingest_result res;
    rc = fcz_ingest_pdb_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n_files, text_bytes, ctx->stage[2].as<char>(),
                            ctx->stage[3].as<uint32_t>(), ctx->stage[4].as<uint32_t>(), anchor_threshold, flags, &res);
    if (rc) return rc;
    memcpy(counts, ctx->ig_counts, sizeof ctx->ig_counts);
    return FCZ_OK;
}

static int ingest_fetch_meta(fcz_ctx* ctx, uint32_t* chain_file, uint32_t* chain_meta, int32_t* file_status, uint32_t* refused) {
    const fcz_ingest_result& r = ctx->ig_res;
    const uint32_t C = r.batch.n_chains;
    if (chain_file && C) HIP_TRY(hipMemcpyAsync(chain_file, r.chain_file, 4 * (size_t)C, hipMemcpyDeviceToHost, ctx->stream));
    if (chain_meta && C) HIP_TRY(hipMemcpyAsync(chain_meta, r.chain_meta, 4 * (size_t)C, hipMemcpyDeviceToHost, ctx->stream));
    if (file_status && r.n_files) HIP_TRY(hipMemcpyAsync(file_status, r.file_status, 4 * (size_t)r.n_files, hipMemcpyDeviceToHost, ctx->stream));
    if (refused && r.n_refused) HIP_TRY(hipMemcpyAsync(refused, r.refused, 8 * (size_t)r.n_refused, hipMemcpyDeviceToHost, ctx->stream));
    return FCZ_OK;
}

int fcz_ingest_chain_names_fetch(fcz_ctx* ctx, uint32_t* chain_name4) {
    if (!ctx) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const fcz_ingest_result& r = ctx->ig_res;
    if (r.batch.n_chains) {
        if (!chain_name4) return FCZ_E_INVALID_ARG;
        HIP_TRY(hipMemcpyAsync(chain_name4, r.chain_name4, 4 * (size_t)r.batch.n_chains, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return FCZ_OK;
}

int fcz_ingest_pdb_fetch(fcz_ctx* ctx, const fcz_chain_batch* hb, uint32_t* chain_file, uint32_t* chain_meta, int32_t* file_status, uint32_t* refused) {
    if (!ctx) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const fcz_ingest_result& r = ctx->ig_res;
    const fcz_chain_batch& d = r.batch;
    const size_t C = d.n_chains, R = d.n_residues, M = d.n_atoms, TB = ctx->ig_counts[3];
    if (hb && r.n_files) {
        auto cp = [&](const void* dst, const void* src, size_t bytes) -> int {
            if (!bytes) return FCZ_OK;
            if (!dst) return FCZ_E_INVALID_ARG;
            HIP_TRY(hipMemcpyAsync(const_cast<void*>(dst), src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            return FCZ_OK;
        };
        int rc;
        if ((rc = cp(hb->res_off, d.res_off, 4 * (C + 1))) || (rc = cp(hb->atom_off, d.atom_off, 4 * (R + 1))) || (rc = cp(hb->x, d.x, 4 * M)) ||
            (rc = cp(hb->y, d.y, 4 * M)) || (rc = cp(hb->z, d.z, 4 * M)) || (rc = cp(hb->atom_code, d.atom_code, M)) || (rc = cp(hb->res_code, d.res_code, R)) ||
            (rc = cp(hb->bfac_ca, d.bfac_ca, 4 * R)) || (rc = cp(hb->first_res_index, d.first_res_index, 4 * C)) ||
            (rc = cp(hb->first_atom_index, d.first_atom_index, 4 * C)) || (rc = cp(hb->chain_id, d.chain_id, C)) || (rc = cp(hb->titles, d.titles, TB)) ||
            (rc = cp(hb->title_off, d.title_off, 4 * (C + 1))))
            return rc;
    }
    int rc = ingest_fetch_meta(ctx, chain_file, chain_meta, file_status, refused);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

// the compress half of fcz_compress_pdb_begin / fcz_compress_gz_begin: sizes, then the codec, on the batch the ingest left in the ctx
static int compress_resident_batch(fcz_ctx* ctx, uint64_t* fcz_bytes);

int fcz_compress_pdb_begin(fcz_ctx* ctx, const uint8_t* text, const uint64_t* file_off, uint32_t n_files, const char* names, const uint32_t* name_off,
                           const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5], uint64_t* fcz_bytes) {
    if (!fcz_bytes) return FCZ_E_INVALID_ARG;
    *fcz_bytes = 0; if (ctx) ctx->ig_fcz_bytes = 0;
    int rc = fcz_ingest_pdb_begin(ctx, text, file_off, n_files, names, name_off, stem_len, anchor_threshold, flags, counts);
    if (rc) return rc;
    return compress_resident_batch(ctx, fcz_bytes);
}

// ------------------------------------------------------------------------------------------------
// inflate: gzip members -> text on the device (fcz_inflate.h)
// ------------------------------------------------------------------------------------------------
int fcz_inflate_sizes(const uint8_t* gz, const uint64_t* gz_off, uint32_t n, const uint8_t* kind, uint64_t* text_off) {
    if (!text_off || (n && (!gz || !gz_off))) return FCZ_E_INVALID_ARG;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        text_off[i] = pos;
        if (gz_off[i + 1] < gz_off[i]) return FCZ_E_INVALID_ARG;
        const uint64_t len = gz_off[i + 1] - gz_off[i];
        if (kind && kind[i] == 0) { pos += len; continue; }
        if (len < 18) continue;
        const uint8_t* t = gz + gz_off[i + 1] - 4;
        const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
        // DEFLATE expands by at most 1032 : 1 (258 bytes per two bits, zlib's documented bound); an ISIZE beyond that is not this
        // member's text size (a corrupt trailer, several members, > 4 GB of text): zlib's to read
        if (isize > (len - 18) * 1032 + 1024 || isize >= (1ull << 31)) continue;
        pos += isize;
    }
    text_off[n] = pos;
    return FCZ_OK;
}

int fcz_inflate_dev(fcz_ctx* ctx, const uint8_t* gz_dev, const uint64_t* gz_off_dev, uint32_t n, const uint8_t* kind_dev,
                    const uint64_t* text_off_dev, uint8_t* text_dev, int32_t* status_dev) {
    if (!ctx) return FCZ_E_INVALID_ARG;
    if (n == 0) return FCZ_OK;
    if (!gz_dev || !gz_off_dev || !text_off_dev || !text_dev || !status_dev) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    {
        span_guard g(ctx, "inflate");
        hipLaunchKernelGGL(inflate::k_inflate, dim3(n), dim3(WAVE), 0, ctx->stream, gz_dev, gz_off_dev, n, kind_dev, text_off_dev, text_dev, status_dev);
    }
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

// the files' bytes -> ctx->gz_raw, their text -> ctx->stage[0] (text offsets in ctx->gz_toff, statuses in ctx->gz_status)
static int inflate_into_stage(fcz_ctx* ctx, const uint8_t* data, const uint64_t* file_off, uint32_t n, const uint8_t* kind, const uint64_t* text_off) {
    const uint64_t raw_bytes = file_off[n], text_bytes = text_off[n];
    int rc;
    if ((rc = ctx->gz_raw.ensure(std::max<uint64_t>(raw_bytes, 16))) || (rc = ctx->gz_off.ensure(8 * ((size_t)n + 1))) || (rc = ctx->gz_kind.ensure(std::max<size_t>(n, 16))) ||
        (rc = ctx->gz_toff.ensure(8 * ((size_t)n + 1))) || (rc = ctx->gz_status.ensure(4 * (size_t)n + 16)) || (rc = ctx->stage[0].ensure(std::max<uint64_t>(text_bytes, 16))))
        return rc;
    if (raw_bytes) HIP_TRY(hipMemcpyAsync(ctx->gz_raw.p, data, raw_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->gz_off.p, file_off, 8 * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->gz_toff.p, text_off, 8 * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (kind) HIP_TRY(hipMemcpyAsync(ctx->gz_kind.p, kind, n, hipMemcpyHostToDevice, ctx->stream));
    return fcz_inflate_dev(ctx, ctx->gz_raw.as<uint8_t>(), ctx->gz_off.as<uint64_t>(), n, kind ? ctx->gz_kind.as<uint8_t>() : nullptr,
                           ctx->gz_toff.as<uint64_t>(), ctx->stage[0].as<uint8_t>(), ctx->gz_status.as<int32_t>());
}

int fcz_inflate(fcz_ctx* ctx, const uint8_t* gz, const uint64_t* gz_off, uint32_t n, const uint8_t* kind, const uint64_t* text_off,
                uint8_t* text, int32_t* status) {
    if (!ctx) return FCZ_E_INVALID_ARG;
    if (n == 0) return FCZ_OK;
    if (!gz || !gz_off || !text_off || !status || (text_off[n] && !text)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->sizes_fresh = false;   // staging buffers are rewritten
    int rc = inflate_into_stage(ctx, gz, gz_off, n, kind, text_off);
    if (rc) return rc;
    if (text_off[n]) HIP_TRY(hipMemcpyAsync(text, ctx->stage[0].p, text_off[n], hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(status, ctx->gz_status.p, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

int fcz_ingest_gz_begin(fcz_ctx* ctx, const uint8_t* data, const uint64_t* file_off, uint32_t n_files, const uint8_t* is_gz, const char* names,
                        const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5]) {
    if (!ctx || !counts || anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    if (n_files && (!data || !file_off || !names || !name_off || !stem_len)) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->sizes_fresh = false;   // staging buffers are rewritten
    memset(counts, 0, 5 * sizeof(uint32_t));
    if (n_files == 0) { memset(&ctx->ig_res, 0, sizeof ctx->ig_res); memset(ctx->ig_counts, 0, sizeof ctx->ig_counts); return FCZ_OK; }
    std::vector<uint64_t> text_off((size_t)n_files + 1);
    int rc = fcz_inflate_sizes(data, file_off, n_files, is_gz, text_off.data());
    if (rc) return rc;
    if ((rc = inflate_into_stage(ctx, data, file_off, n_files, is_gz, text_off.data()))) return rc;
    const uint32_t name_bytes = name_off[n_files];
    if ((rc = ctx->stage[2].ensure(std::max<size_t>(name_bytes, 16))) || (rc = ctx->stage[3].ensure(4 * ((size_t)n_files + 1))) || (rc = ctx->stage[4].ensure(4 * (size_t)n_files)))
        return rc;
    if (name_bytes) HIP_TRY(hipMemcpyAsync(ctx->stage[2].p, names, name_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[3].p, name_off, 4 * ((size_t)n_files + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[4].p, stem_len, 4 * (size_t)n_files, hipMemcpyHostToDevice, ctx->stream));
    fcz_ingest_result res;
    rc = fcz_ingest_pdb_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->gz_toff.as<uint64_t>(), n_files, text_off[n_files], ctx->stage[2].as<char>(),
                            ctx->stage[3].as<uint32_t>(), ctx->stage[4].as<uint32_t>(), anchor_threshold, flags, &res);
    if (rc) return rc;
    // a member the device did not inflate left blanks (no atoms): the FILE goes back to the caller's zlib and reader
    hipLaunchKernelGGL(inflate::k_inflate_merge_status, dim3(grid_for(n_files, 256)), dim3(256), 0, ctx->stream, ctx->gz_status.as<int32_t>(), n_files,
                       (int32_t)FCZ_INGEST_HOST_GZIP, const_cast<int32_t*>(ctx->ig_res.file_status));
    HIP_TRY(hipGetLastError());
    memcpy(counts, ctx->ig_counts, sizeof ctx->ig_counts);
    return FCZ_OK;
}

int fcz_compress_gz_begin(fcz_ctx* ctx, const uint8_t* data, const uint64_t* file_off, uint32_t n_files, const uint8_t* is_gz, const char* names,
                          const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5], uint64_t* fcz_bytes) {
    if (!fcz_bytes) return FCZ_E_INVALID_ARG;
    *fcz_bytes = 0; if (ctx) ctx->ig_fcz_bytes = 0;
    int rc = fcz_ingest_gz_begin(ctx, data, file_off, n_files, is_gz, names, name_off, stem_len, anchor_threshold, flags, counts);
    if (rc) return rc;
    return compress_resident_batch(ctx, fcz_bytes);
}

static int compress_resident_batch(fcz_ctx* ctx, uint64_t* fcz_bytes) {
    int rc;
    const fcz_chain_batch& b = ctx->ig_res.batch;
    const uint32_t C = b.n_chains;
    if (C == 0) return FCZ_OK;
    if ((rc = ctx->stage[13].ensure(8 * ((size_t)C + 1))) || (rc = ctx->stage[15].ensure(4 * (size_t)C))) return rc;
    if ((rc = fcz_compress_sizes_dev(ctx, &b, ctx->stage[13].as<uint64_t>()))) return rc;
    uint64_t* tot = reinterpret_cast<uint64_t*>(ctx->pinned + 14);
    HIP_TRY(hipMemcpyAsync(tot, ctx->stage[13].as<uint64_t>() + C, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t bytes = *tot;
    if ((rc = ctx->stage[14].ensure(std::max<uint64_t>(bytes, 16)))) return rc;
    if ((rc = fcz_compress_batch_dev(ctx, &b, ctx->stage[13].as<uint64_t>(), ctx->stage[14].as<uint8_t>(), ctx->stage[15].as<int32_t>()))) return rc;
    ctx->ig_fcz_bytes = bytes; *fcz_bytes = bytes;
    return FCZ_OK;
}

int fcz_compress_pdb_fetch(fcz_ctx* ctx, uint64_t* out_off, int32_t* status, uint32_t* chain_file, uint32_t* chain_meta, int32_t* file_status,
                           uint32_t* refused, uint8_t* blob) {
    if (!ctx) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t C = ctx->ig_res.batch.n_chains;
    if (C) {
        if (out_off) HIP_TRY(hipMemcpyAsync(out_off, ctx->stage[13].p, 8 * ((size_t)C + 1), hipMemcpyDeviceToHost, ctx->stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, ctx->stage[15].p, 4 * (size_t)C, hipMemcpyDeviceToHost, ctx->stream));
        if (ctx->ig_fcz_bytes) {
            if (!blob) return FCZ_E_INVALID_ARG;
            HIP_TRY(hipMemcpyAsync(blob, ctx->stage[14].p, ctx->ig_fcz_bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
    } else if (out_off) out_off[0] = 0;
    int rc = ingest_fetch_meta(ctx, chain_file, chain_meta, file_status, refused);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

// ------------------------------------------------------------------------------------------------
// decompress
// ------------------------------------------------------------------------------------------------
static inline uint32_t h_u16(const uint8_t* p) { return p[0] | (p[1] << 8); }
static inline uint32_t h_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

static void parse_entry(const uint8_t* e, uint64_t len, fcz_entry_info* info) {
    memset(info, 0, sizeof *info);
    if (len < 76) { info->status = FCZ_E_TRUNCATED; return; }
    if (memcmp(e, "FCMP", 4) != 0) { info->status = FCZ_E_BAD_MAGIC; return; }
    const uint32_t n = h_u16(e + 4);
    info->n_residues = n;
    info->n_atoms_header = h_u16(e + 6);
    info->first_res_index = (int32_t)h_u16(e + 8);
    info->first_atom_index = (int32_t)h_u16(e + 10);
    info->n_anchors = e[12];
    info->chain_id = (char)e[13];
    info->n_sidechain_torsions = h_u32(e + 16);
    info->first_residue = (char)e[20];
    info->last_residue = (char)e[21];
    info->title_len = h_u32(e + 24);
    info->title_off = 76 + 4 * info->n_anchors;
    if (info->title_len > len || info->n_sidechain_torsions > len) { info->status = FCZ_E_TRUNCATED; return; }
    const rec_layout L = make_layout(n, info->n_anchors, info->title_len, info->n_sidechain_torsions);
    if ((uint64_t)L.size > len) { info->status = FCZ_E_TRUNCATED; return; }
    if (n < 2 || info->n_anchors < 2) { info->status = FCZ_E_TOO_SHORT; return; }
    info->has_oxt = e[L.o_oxt];
    uint32_t na = 0, nsc = 0;
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rc = e[L.o_words + 8 * k] >> 3;
        if (k == 0) { rc = 23; for (int i = 0; i < 24; i++) if (host_tab::h_res1[i] == info->first_residue) { rc = i; break; } }
        if (rc >= 24) rc = 23;
        if (!(rc < 20 || rc == 23)) { info->status = FCZ_E_RESIDUE; return; }
        na += host_tab::h_res_natoms[rc]; nsc += host_tab::h_res_natoms[rc] - 3;
    }
    if (nsc != info->n_sidechain_torsions) { info->status = FCZ_E_TRUNCATED; return; }
    for (uint32_t s = 0; s + 1 < info->n_anchors; s++) {
        const int a = (int)h_u32(e + L.o_aidx + 4 * s), b = (int)h_u32(e + L.o_aidx + 4 * (s + 1));
        if (a < 0 || b < a || b > (int)n - 1 || (s == 0 && a != 0) || (s + 2 == info->n_anchors && b != (int)n - 1)) {
            info->status = FCZ_E_TRUNCATED; return;
        }
    }
    info->n_atoms_out = na + (info->has_oxt ? 1 : 0);
    info->status = FCZ_OK;
}

int fcz_decompress_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n, fcz_entry_info* info, uint32_t* res_off,
                         uint32_t* atom_off) {
    if (!blob || !off || !res_off || !atom_off) return FCZ_E_INVALID_ARG;
    uint64_t r = 0, a = 0;
    for (uint32_t i = 0; i < n; i++) {
        res_off[i] = (uint32_t)r; atom_off[i] = (uint32_t)a;
        fcz_entry_info tmp;
        fcz_entry_info* pi = info ? &info[i] : &tmp;
        parse_entry(blob + off[i], off[i + 1] - off[i], pi);
        if (pi->status == FCZ_OK) { r += pi->n_residues; a += pi->n_atoms_out; }
    }
    res_off[n] = (uint32_t)r; atom_off[n] = (uint32_t)a;
    // offsets are 32-bit: a batch whose residues or atoms reach 2^32 is refused, not wrapped
    if ((r >> 32) || (a >> 32)) return FCZ_E_INVALID_ARG;
    return FCZ_OK;
}

int fcz_check(const uint8_t* e, uint64_t len) {
    fcz_entry_info info;
    parse_entry(e, len, &info);
    if (info.status == FCZ_E_BAD_MAGIC || info.status == FCZ_E_TRUNCATED) return info.status;
    const rec_layout L = make_layout(info.n_residues, info.n_anchors, info.title_len, info.n_sidechain_torsions);
    bool empty_bb = true, empty_sc = true, empty_t = true;
    for (uint32_t k = 0; k < info.n_residues; k++) {
        const uint8_t* b = e + L.o_words + 8 * k;
        if ((b[0] & 7) | b[1] | b[2] | b[3] | b[4]) empty_bb = false;
        if (e[L.o_tbytes + k]) empty_t = false;
    }
    for (uint32_t k = 0; k < info.n_sidechain_torsions; k++) if (e[L.o_sc + k]) empty_sc = false;
    if (empty_bb) return 4;
    if (empty_sc) return 5;
    if (empty_t) return 6;
    return 0;
}

// The sizes pass of the decompress path: per-entry validation and counts (k_entry_sizes), then -- in three launches -- their
// exclusive prefixes, the longest anchor segment of the batch (sizes the ring of k_backbone), the totals, and the entries ordered by
// residue count for k_backbone (counting sort, longest first): k_sizes_reduce / _mid / _apply (fcz_kernels.h). The totals come
// back as ONE 32-byte copy into pinned host words after one stream synchronisation. atom_off_dev may be null (prefix not needed).
static int run_entry_sizes(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, uint32_t* res_off_dev,
                           uint32_t* atom_off_dev) {
    for (int k = 0; k < 8; k++) ctx->pinned[k] = 0;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(res_off_dev, 0, 4, ctx->stream));
        if (atom_off_dev) HIP_TRY(hipMemsetAsync(atom_off_dev, 0, 4, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return FCZ_OK;
    }
    int rc = ctx->cnt.ensure(sizeof(uint32_t) * 4 * (size_t)n); if (rc) return rc;
    uint32_t* cr = ctx->cnt.as<uint32_t>(); uint32_t* ca = cr + n; int32_t* st = (int32_t*)(ca + n); uint32_t* seg = (uint32_t*)(st + n);
    // len_perm: [perm n][hist LEN_BUCKETS][cursor LEN_BUCKETS + 1][pad][maxseg 2][pad 2][totals 8]
    if ((rc = ctx->len_perm.ensure(sizeof(uint32_t) * ((size_t)n + 2 * LEN_BUCKETS + 16)))) return rc;
    uint32_t* perm = ctx->len_perm.as<uint32_t>(); uint32_t* hist = perm + n; uint32_t* cursor = hist + LEN_BUCKETS;
    uint32_t* maxseg = cursor + LEN_BUCKETS + 4; sizes_totals* totals = (sizes_totals*)(maxseg + 4);
    const unsigned nb = grid_for(n, SZ_CHUNK);
    if ((rc = ctx->scan_tmp.ensure(sizeof(unsigned long long) * 2 * ((size_t)nb + 1)))) return rc;
    unsigned long long* part_r = ctx->scan_tmp.as<unsigned long long>(); unsigned long long* part_a = part_r + nb + 1;
    // The residue-code array (k_entry_sizes -> k_res_index) has one slot per 8 bytes of the records; their total size is only known
    // on the device, so the array grows when a pass reports that it needed more -- that pass runs again (the first call, or a larger
    // batch than any before), every later one runs once with no host round trip before the launches.
    for (int attempt = 0; attempt < 2; attempt++) {
        HIP_TRY(hipMemsetAsync(hist, 0, sizeof(uint32_t) * (2 * LEN_BUCKETS + 16), ctx->stream));
        hipLaunchKernelGGL(k_entry_sizes, dim3(grid_for(n, GROUPS_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n, cr, ca, st, seg,
                           ctx->codes.as<uint8_t>(), (uint64_t)ctx->codes.cap);
        hipLaunchKernelGGL(k_sizes_reduce, dim3(nb), dim3(1024), 0, ctx->stream, cr, ca, seg, n, part_r, part_a, hist, maxseg);
        hipLaunchKernelGGL(k_sizes_mid, dim3(1), dim3(1024), 0, ctx->stream, nb, part_r, part_a, hist, cursor, maxseg, totals, off_dev + n);
        hipLaunchKernelGGL(k_sizes_apply, dim3(nb), dim3(1024), 0, ctx->stream, cr, ca, n, part_r, part_a, res_off_dev, atom_off_dev, cursor, perm);
        HIP_TRY(hipGetLastError());
        // pinned: [0] residues [1] atoms [3] longest segment [4] most segments [5] long chains [6] offset overflow; [2], [7] code slots
        HIP_TRY(hipMemcpyAsync(&ctx->pinned[0], totals, sizeof(sizes_totals), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        const uint64_t need = (uint64_t)ctx->pinned[2] | ((uint64_t)ctx->pinned[7] << 32);
        if (need <= ctx->codes.cap) break;
        if (attempt == 1) return FCZ_E_HIP;
        if ((rc = ctx->codes.ensure((size_t)(need + need / 8)))) return rc;
    }
    if (ctx->pinned[6]) return FCZ_E_INVALID_ARG;   // 2^32 residues or atoms in one batch: split it
    return FCZ_OK;
}

int fcz_decompress_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                             uint32_t* res_off_dev, uint32_t* atom_off_dev, uint32_t* total_res, uint32_t* total_atoms) {
    if (!ctx || !blob_dev || !off_dev || !res_off_dev || !atom_off_dev) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->sizes_fresh = false;   // the pass below overwrites what an earlier sizes call left; set again only when it succeeds
    {
        span_guard g(ctx, "decompress_sizes");
        int rc = run_entry_sizes(ctx, blob_dev, off_dev, n, res_off_dev, atom_off_dev);
        if (rc) return rc;
    }
    if (total_res) *total_res = ctx->pinned[0];
    if (total_atoms) *total_atoms = ctx->pinned[1];
    // the fcz_decompress_batch_dev call that follows on the same entries reuses the totals and the length order
    ctx->sized_blob = blob_dev; ctx->sized_off = off_dev; ctx->sized_n = n;
    ctx->sized_R = ctx->pinned[0]; ctx->sized_maxseg = ctx->pinned[3]; ctx->sized_maxnseg = ctx->pinned[4]; ctx->sized_nlong = ctx->pinned[5];
    ctx->sizes_fresh = true;
    return FCZ_OK;
}

// Totals and length order for a batch call: taken from the preceding sizes call on the same entries, else recomputed.
static int ensure_sizes(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, uint32_t* total_res, uint32_t* max_seg_len,
                        uint32_t* max_nseg, uint32_t* n_long) {
    if (ctx->sizes_fresh && ctx->sized_blob == blob_dev && ctx->sized_off == off_dev && ctx->sized_n == n) {
        ctx->sizes_fresh = false;   // single use: the records may be rewritten before the next call
        *total_res = ctx->sized_R; *max_seg_len = ctx->sized_maxseg; *max_nseg = ctx->sized_maxnseg; *n_long = ctx->sized_nlong;
        return FCZ_OK;
    }
    ctx->sizes_fresh = false;   // the pass below overwrites what a remembered sizes call left (length order, residue codes)
    int rc = ctx->stage[16].ensure(sizeof(uint32_t) * ((size_t)n + 1)); if (rc) return rc;
    if ((rc = run_entry_sizes(ctx, blob_dev, off_dev, n, ctx->stage[16].as<uint32_t>(), nullptr))) return rc;
    *total_res = ctx->pinned[0]; *max_seg_len = ctx->pinned[3]; *max_nseg = ctx->pinned[4]; *n_long = ctx->pinned[5];
    return FCZ_OK;
}

int fcz_decompress_batch_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                             const uint32_t* res_off_dev, const uint32_t* atom_off_dev, int alt_order,
                             const fcz_atoms_out* out_dev) {
    if (!ctx || !blob_dev || !off_dev || !res_off_dev || !atom_off_dev || !out_dev) return FCZ_E_INVALID_ARG;
    if (!out_dev->x || !out_dev->y || !out_dev->z || !out_dev->bfac_res) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return FCZ_OK;
    uint32_t R = 0, max_seg = 0, max_nseg = 0, n_long = 0;
    int rc = ensure_sizes(ctx, blob_dev, off_dev, n, &R, &max_seg, &max_nseg, &n_long);
    if (rc) return rc;
    if (R == 0) return FCZ_OK;
    rc = ctx->bb.ensure(sizeof(v3) * 3 * (size_t)R); if (rc) return rc;
    const uint32_t* perm = ctx->len_perm.as<uint32_t>();
    const bool fast_bb = ctx->numerics == FCZ_NUMERICS_FAST, fast_sc = fast_bb;
#ifdef FCZ_PROFILING
    // measurement builds only (tools/hbm_busy_probe.py builds its own library with -DFCZ_PROFILING): FCZ_PROFILE_STAGES=<mask> leaves
    // stages out. The product library never reads the environment here -- a leaked variable must not turn a decompress into a no-op.
    { const char* ps = getenv("FCZ_PROFILE_STAGES"); ctx->profile_stages = ps ? ((unsigned)strtoul(ps, nullptr, 0) & 7u) : 7u; }
#else
    ctx->profile_stages = 7u;
#endif
    if (!(ctx->profile_stages & 1u)) { /* profiling aid: no backbone launch */ }
    else if (fast_bb) {
        // plain-float backbone: 8 chains per wavefront, the forward atoms of a segment stay in LDS; only segments longer than
        // one chunk (FB_K residue steps) park them in a scratch column, and then the launch is cut so that the columns of the
        // wavefronts in flight fit 4 GB
        span_guard g(ctx, "decompress_backbone");
        const uint32_t groups = grid_for(n, FB_CH);
        const bool need_scratch = max_seg > (uint32_t)FB_K + 1;
        const uint32_t col = need_scratch ? 3 * max_seg : 0;
        uint32_t chunk = groups;
        if (need_scratch) {
            chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(groups, ((size_t)4 << 30) / ((size_t)FB_CH * col * sizeof(v3))));
            rc = ctx->fast_scratch.ensure((size_t)chunk * FB_CH * col * sizeof(v3)); if (rc) return rc;
        }
        for (uint32_t g0 = 0; g0 < groups; g0 += chunk) {
            const uint32_t gn = std::min(chunk, groups - g0);
            const uint32_t slots = std::min<uint32_t>(n - g0 * FB_CH, gn * FB_CH);
            hipLaunchKernelGGL(k_backbone_fast, dim3(gn), dim3(WAVE), 0, ctx->stream, blob_dev, off_dev, n, slots, res_off_dev,
                               perm + (size_t)g0 * FB_CH, need_scratch ? ctx->fast_scratch.as<v3>() : nullptr, col, ctx->bb.as<v3>());
        }
    } else {
        const uint32_t ring_rows = 3 * (max_seg ? max_seg : 1);
        const size_t slot_atoms = (size_t)ring_rows * WAVE, slot_trig = (size_t)(ring_rows / 3) * 6 * WAVE;
        // Long chains (>= FCZ_LONG_CHAIN residues, the head of the length order): when there are too few of them to fill the
        // GPU their serial forward pass would hold the launch for ~8 us per residue, so they take the split form (forward
        // pass, then one block per segment for the reverse pass) on a second stream beside the fused kernel of the rest.
        const uint32_t groups_long_all = grid_for(n_long, WAVE);
        const bool split_long = n_long > 0 && max_nseg > 0 && groups_long_all < 2u * 4u * (uint32_t)ctx->n_cu;
        const uint32_t n_split = split_long ? n_long : 0;
        if (split_long) {
            const size_t per_group = (size_t)max_nseg * (slot_atoms * sizeof(v3) + slot_trig * sizeof(float));
            const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(groups_long_all, ((size_t)6 << 30) / per_group));
            rc = ctx->fwd_long.ensure(sizeof(v3) * slot_atoms * max_nseg * chunk); if (rc) return rc;
            rc = ctx->wring_long.ensure(sizeof(float) * slot_trig * max_nseg * chunk); if (rc) return rc;
            HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
            HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            for (uint32_t g0 = 0; g0 < groups_long_all; g0 += chunk) {
                const uint32_t g = std::min(chunk, groups_long_all - g0);
                const uint32_t slots = std::min<uint32_t>(n_long - g0 * WAVE, g * WAVE);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_backbone<1>), dim3(g), dim3(WAVE), 0, ctx->stream2, blob_dev, off_dev, n, slots, res_off_dev,
                                   perm + (size_t)g0 * WAVE, ctx->fwd_long.as<v3>(), ctx->wring_long.as<float>(), ring_rows, max_nseg, ctx->bb.as<v3>(), 0u, (uint32_t*)nullptr);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_backbone<2>), dim3(g * max_nseg), dim3(WAVE), 0, ctx->stream2, blob_dev, off_dev, n, slots, res_off_dev,
                                   perm + (size_t)g0 * WAVE, ctx->fwd_long.as<v3>(), ctx->wring_long.as<float>(), ring_rows, max_nseg, ctx->bb.as<v3>(), 0u, (uint32_t*)nullptr);
            }
            HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
        }
        const uint32_t n_fused = n - n_split;
        const uint32_t groups = grid_for(n_fused, WAVE);
        // persistent grid: as many wavefronts as the chip keeps in flight (FCZ_BACKBONE_MIN_WAVES per SIMD), one ring slot each --
        // 2 048 x 107.5 KB = 220 MB at the headline batch whatever its size, and a slot is rewritten by the wavefront's next
        // group instead of being left behind dirty (one slot per group was 1.68 GB at 1 M chains)
        const uint32_t resident = FCZ_BB_PERSIST ? (uint32_t)ctx->n_cu * 4u * FCZ_BACKBONE_MIN_WAVES : groups;
        const uint32_t blocks0 = std::min(groups, resident);
        rc = ctx->fwd.ensure(sizeof(v3) * (size_t)std::max<uint32_t>(blocks0, 1) * slot_atoms + 64); if (rc) return rc;
        rc = ctx->wring.ensure(sizeof(float) * (size_t)std::max<uint32_t>(blocks0, 1) * slot_trig); if (rc) return rc;
        uint32_t* next_group = (uint32_t*)(ctx->fwd.as<uint8_t>() + sizeof(v3) * (size_t)std::max<uint32_t>(blocks0, 1) * slot_atoms);
        if (FCZ_BB_PERSIST && groups) HIP_TRY(hipMemsetAsync(next_group, 0, sizeof(uint32_t), ctx->stream));
        {
            span_guard g(ctx, "decompress_backbone");
            if (groups)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_backbone<0>), dim3(blocks0), dim3(WAVE), 0, ctx->stream, blob_dev, off_dev, n, n_fused, res_off_dev,
                                   perm + n_split, ctx->fwd.as<v3>(), ctx->wring.as<float>(), ring_rows, 1u, ctx->bb.as<v3>(), groups, next_group);
            if (split_long) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    }
    rc = ctx->res_aoff.ensure(sizeof(uint32_t) * ((size_t)R + 1)); if (rc) return rc;
    rc = ctx->res_rc.ensure((size_t)R); if (rc) return rc;
    rc = ctx->res_sc.ensure(sizeof(uint32_t) * 3 * (size_t)R); if (rc) return rc;
    if (ctx->profile_stages & 2u) {
        span_guard g(ctx, "decompress_index");
        hipLaunchKernelGGL(k_res_index, dim3(grid_for(n, WAVES_PER_BLOCK)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n,
                           res_off_dev, atom_off_dev, R, ctx->res_aoff.as<uint32_t>(), ctx->res_rc.as<uint8_t>(),
                           ctx->res_sc.as<uint32_t>(), *out_dev, ctx->codes.as<uint8_t>());
        // entries of up to 64 residues (k_res_index leaves them): four to a wavefront, a persistent grid over chunks of 16 entries
        hipLaunchKernelGGL(k_res_index_rows, dim3(std::min<uint32_t>(grid_for(grid_for(n, RI_CHUNK), WAVES_PER_BLOCK), (uint32_t)ctx->n_cu * 8u)), dim3(BLOCK), 0, ctx->stream, blob_dev, off_dev, n,
                           res_off_dev, atom_off_dev, R, ctx->res_aoff.as<uint32_t>(), ctx->res_rc.as<uint8_t>(),
                           ctx->res_sc.as<uint32_t>(), *out_dev, ctx->codes.as<uint8_t>());
    }
    if (ctx->profile_stages & 4u) {
        span_guard g(ctx, "decompress_sidechain");
        const uint32_t n_tiles = grid_for(R, SC_TILE);
        const uint32_t blocks = std::min<uint32_t>(n_tiles, (uint32_t)ctx->n_cu * FCZ_SIDECHAIN_MIN_BLOCKS * FCZ_SC_GRID_FACTOR);
        // 256-residue tiles; a tile with more atoms than the staging buffer holds is listed as two 128-residue halves for
        // the second launch (an empty list on any real protein: that launch then costs its table prologue)
        rc = ctx->tile_work.ensure(sizeof(uint32_t) * (2 * (size_t)n_tiles + 4)); if (rc) return rc;
        uint32_t* punt_count = ctx->tile_work.as<uint32_t>(); uint32_t* punt_list = punt_count + 4;
        HIP_TRY(hipMemsetAsync(punt_count, 0, sizeof(uint32_t) * 4, ctx->stream));
        const uint32_t blocks_half = std::min<uint32_t>(2 * n_tiles, (uint32_t)ctx->n_cu);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK), 0, ctx->stream, R, n_tiles, (uint32_t)SC_TILE, (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr, punt_list, punt_count, ctx->res_aoff.as<uint32_t>(), ctx->res_rc.as<uint8_t>(),
                               ctx->res_sc.as<uint32_t>(), ctx->bb.as<v3>(), alt_order, *out_dev);
            hipLaunchKernelGGL(kernel, dim3(blocks_half), dim3(BLOCK), 0, ctx->stream, R, 2 * n_tiles, (uint32_t)SC_TILE / 2, (const uint32_t*)punt_list,
                               (const uint32_t*)punt_count, (uint32_t*)nullptr, (uint32_t*)nullptr, ctx->res_aoff.as<uint32_t>(),
                               ctx->res_rc.as<uint8_t>(), ctx->res_sc.as<uint32_t>(), ctx->bb.as<v3>(), alt_order, *out_dev);
        };
        if (fast_sc) launch(HIP_KERNEL_NAME(k_sidechain<true>)); else launch(HIP_KERNEL_NAME(k_sidechain<false>));
    }
    HIP_TRY(hipGetLastError());
    return FCZ_OK;
}

int fcz_decompress_batch(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, const uint32_t* res_off,
                         const uint32_t* atom_off, int alt_order, const fcz_atoms_out* out) {
    if (!ctx || !blob || !off || !res_off || !atom_off || !out) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->sizes_fresh = false;   // the staging buffers the cache is keyed on are about to be rewritten
    if (n == 0) return FCZ_OK;
    const uint64_t blob_bytes = off[n];
    const uint32_t R = res_off[n], M = atom_off[n];
    int rc;
    if ((rc = ctx->stage[0].ensure(std::max<uint64_t>(blob_bytes, 16)))) return rc;
    if ((rc = ctx->stage[1].ensure(sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->stage[2].ensure(sizeof(uint32_t) * ((size_t)n + 1)))) return rc;
    if ((rc = ctx->stage[3].ensure(sizeof(uint32_t) * ((size_t)n + 1)))) return rc;
    for (int i = 4; i < 7; i++) if ((rc = ctx->stage[i].ensure(std::max<size_t>(sizeof(float) * (size_t)M, 16)))) return rc;
    if ((rc = ctx->stage[7].ensure(std::max<size_t>(sizeof(float) * (size_t)R, 16)))) return rc;
    if ((rc = ctx->stage[8].ensure(std::max<size_t>((size_t)R, 16)))) return rc;
    if ((rc = ctx->stage[9].ensure(std::max<size_t>((size_t)M, 16)))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->stage[0].p, blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[1].p, off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[2].p, res_off, sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->stage[3].p, atom_off, sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    fcz_atoms_out dv;
    dv.x = ctx->stage[4].as<float>(); dv.y = ctx->stage[5].as<float>(); dv.z = ctx->stage[6].as<float>();
    dv.bfac_res = ctx->stage[7].as<float>();
    dv.res_code = out->res_code ? ctx->stage[8].as<uint8_t>() : nullptr;
    dv.atom_code = out->atom_code ? ctx->stage[9].as<uint8_t>() : nullptr;
    rc = fcz_decompress_batch_dev(ctx, ctx->stage[0].as<uint8_t>(), ctx->stage[1].as<uint64_t>(), n, ctx->stage[2].as<uint32_t>(),
                                  ctx->stage[3].as<uint32_t>(), alt_order, &dv);
    if (rc) return rc;
    if (M) {
        HIP_TRY(hipMemcpyAsync(out->x, dv.x, sizeof(float) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out->y, dv.y, sizeof(float) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out->z, dv.z, sizeof(float) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
        if (out->atom_code) HIP_TRY(hipMemcpyAsync(out->atom_code, dv.atom_code, (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (R) {
        HIP_TRY(hipMemcpyAsync(out->bfac_res, dv.bfac_res, sizeof(float) * (size_t)R, hipMemcpyDeviceToHost, ctx->stream));
        if (out->res_code) HIP_TRY(hipMemcpyAsync(out->res_code, dv.res_code, (size_t)R, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}

// ------------------------------------------------------------------------------------------------
// self-test hooks: run the device numerics over caller-chosen float bit patterns so tests can pin
// them against the host libm (tests/test_device_math.py). mode 0: acos_deg, 1: sinf, 2: cosf, 3: deg2rad, 4: norm,
// 5: getCosineTheta, 6-8: place_atom x/y/z, 9/10: sine / cosine of sincosf_pair (the form the kernels call),
// 11: acos_deg_f32 (the float approximation behind the side-chain torsion byte), 12/13: sine / cosine of sincosf_pair_any (any float)
// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace fcz {
// inputs of the multi-argument self tests come from an integer hash of the index so that the host
// checker (oracle/fcz_oracle.c: fcz_oracle_math_sweep) regenerates exactly the same floats
__device__ __forceinline__ float st_hash_float(uint32_t u, uint32_t salt, float scale) {
    uint32_t h = (u ^ salt) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(int32_t)h * scale;
}
__global__ void k_selftest_math(int mode, uint32_t start_bits, uint32_t stride, uint32_t count, float* __restrict__ outv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t u = start_bits + i * stride;
    const float x = __uint_as_float(u);
    const float sc = 0x1p-29f;   // hashed coordinates in (-4, 4)
    float r;
    if (mode == 0) r = acos_deg(x);
    else if (mode == 1) r = sinf_glibc(x);
    else if (mode == 2) r = cosf_glibc(x);
    else if (mode == 3) r = deg2rad(x);
    else if (mode == 11) r = acos_deg_f32(x);
    else if (mode == 9 || mode == 10) { float sn, cs; sincosf_pair(x, &sn, &cs); r = mode == 9 ? sn : cs; }
    else if (mode == 12 || mode == 13) { float sn, cs; sincosf_pair_any(x, &sn, &cs); r = mode == 12 ? sn : cs; }
    else if (mode == 4) r = vnorm(v3{st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)});
    else if (mode == 5) r = vcos_theta(v3{st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)},
                                       v3{st_hash_float(u, 4, sc), st_hash_float(u, 5, sc), st_hash_float(u, 6, sc)});
    else {
        // modes 6..8: x / y / z of place_atom on hashed geometry
        const v3 a{st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)};
        const v3 b{st_hash_float(u, 4, sc), st_hash_float(u, 5, sc), st_hash_float(u, 6, sc)};
        const v3 c{st_hash_float(u, 7, sc), st_hash_float(u, 8, sc), st_hash_float(u, 9, sc)};
        const float L = 1.2f + __builtin_fabsf(st_hash_float(u, 10, 0x1p-33f));
        const float ba = 90.0f + st_hash_float(u, 11, 0x1p-25f);          // (26, 154) degrees
        const float ta = st_hash_float(u, 12, 0x1.6p-24f);                // (-176, 176) degrees
        const v3 d = place_atom(a, b, c, L, ba, ta);
        r = (mode == 6) ? d.x : (mode == 7) ? d.y : d.z;
    }
    outv[i] = r;
}
}  // namespace fcz

namespace fcz {
// float4 per lane, grid-stride; U independent loads in flight per lane before the stores; NT: non-temporal loads and stores
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_f4(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t j = i + (size_t)u * stride; if (j < n) v[u] = NT ? __builtin_nontemporal_load(&src[j]) : src[j]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t j = i + (size_t)u * stride; if (j < n) { if (NT) __builtin_nontemporal_store(v[u], &dst[j]); else dst[j] = v[u]; } }
    }
}
}  // namespace fcz
extern "C" int fcz_selftest_copy(fcz_ctx* ctx, uint64_t bytes, int reps, double* gb_per_s) {
    if (!ctx || !gb_per_s || bytes < 16 || reps < 1) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    *gb_per_s = 0.0;
    const size_t n = (size_t)(bytes / 16);
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n * 16) != hipSuccess) return FCZ_E_NOMEM;
    if (hipMalloc(&b, n * 16) != hipSuccess) { (void)hipFree(a); return FCZ_E_NOMEM; }
    (void)hipMemsetAsync(a, 1, n * 16, ctx->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // the best of a few shapes of the same kernel (wavefronts in flight per CU, loads in flight per lane, cache policy): the ceiling
    // is what the memory system gives the friendliest access pattern, not what one launch shape happens to reach
    double best = 0.0;
    hipError_t err = hipSuccess;
    for (int shape = 0; shape < 12 && err == hipSuccess; shape++) {
        const unsigned grid = (unsigned)ctx->n_cu * (shape % 3 == 0 ? 8u : shape % 3 == 1 ? 16u : 32u);
        const int variant = shape / 3;                          // 0: one load per lane, 1: four, 2: four non-temporal, 3: one non-temporal
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy_f4<1, false>), dim3(grid), dim3(256), 0, ctx->stream, (const f4v*)a, (f4v*)b, n);
            else if (variant == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy_f4<4, false>), dim3(grid), dim3(256), 0, ctx->stream, (const f4v*)a, (f4v*)b, n);
            else if (variant == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy_f4<4, true>), dim3(grid), dim3(256), 0, ctx->stream, (const f4v*)a, (f4v*)b, n);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_copy_f4<1, true>), dim3(grid), dim3(256), 0, ctx->stream, (const f4v*)a, (f4v*)b, n);
        };
        launch();                                               // warm-up
        (void)hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1, ctx->stream);
        err = hipStreamSynchronize(ctx->stream);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (err == hipSuccess && ms > 0.f) best = std::max(best, 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (err != hipSuccess || best <= 0.0) return FCZ_E_HIP;
    *gb_per_s = best;
    return FCZ_OK;
}

extern "C" int fcz_selftest_math(fcz_ctx* ctx, int mode, uint32_t start_bits, uint32_t stride, uint32_t count, float* out_host) {
    if (!ctx || !out_host || mode < 0 || mode > 13) return FCZ_E_INVALID_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (count == 0) return FCZ_OK;
    int rc = ctx->stage[17].ensure(sizeof(float) * (size_t)count); if (rc) return rc;
    hipLaunchKernelGGL(fcz::k_selftest_math, dim3(grid_for(count, 256)), dim3(256), 0, ctx->stream, mode, start_bits, stride, count,
                       ctx->stage[17].as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, ctx->stage[17].p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FCZ_OK;
}


#ifdef FCZ_CW_TIMING
// measurement aid: read and clear the phase counters of k_compress_angles_w
extern "C" int fcz_debug_cw_timing(unsigned long long* out8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fcz::g_cw_timing), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fcz::g_cw_timing), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef FCZ_SC_TIMING
// measurement aid: read and clear the phase counters of k_sidechain
extern "C" int fcz_debug_sc_timing(unsigned long long* out12) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(fcz::g_sc_timing), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fcz::g_sc_timing), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef FCZ_BB_TIMING
// measurement aid: read and clear the phase counters of k_backbone<0>
extern "C" int fcz_debug_bb_timing(unsigned long long* out8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fcz::g_bb_timing), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fcz::g_bb_timing), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef FCZ_IG_TIMING
// measurement aid: read and clear the phase counters of k_ingest_parse
extern "C" int fcz_debug_ig_timing(unsigned long long* out8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fcz::g_ig_timing), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fcz::g_ig_timing), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
