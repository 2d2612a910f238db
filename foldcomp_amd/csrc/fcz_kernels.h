// fcz_kernels.h -- HIP kernels of the FCZ codec for gfx950 (wave64): shared primitives, the size passes and the
// backbone stage of the decompress side. The other stages: fcz_compress.h (compress), fcz_sidechain.h (side chains),
// fcz_pdb.h (PDB text), fcz_extract.h (extract).
//
//   k_compress_sizes   one wavefront per chain: exact record size (Foldcomp::getSize, reference src/foldcomp.cpp:1190-1214)
//   k_entry_sizes      one wavefront per entry: what Foldcomp::read checks (:904-1036) + residue / atom counts, longest
//                      anchor segment and segment count;  k_seg_max reduces the last two over the batch
//   k_len_sort         counting sort of the entries by residue count (longest first) for the lane = chain kernel
//   k_backbone<MODE>   one wavefront per 64 entries, lane = chain: forward NeRF, reverse NeRF, blend (:779-858, 167-273)
//   k_scan_reduce / k_scan_u64 / k_scan_apply   exclusive scans (offsets)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fcz_hip.h"
#include "fcz_math.h"

#define FCZ_TABLE_QUAL __device__ const
#include "aa_tables.h"

namespace fcz {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVE * WAVES_PER_BLOCK;

// ---- unaligned access (FCZ records are byte-packed at arbitrary offsets) ---------------------------
// gfx950 under ROCm runs with unaligned global access enabled; the memcpy forms compile to single
// global_load/store_dword[x2] instructions instead of byte-wise sequences.
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ float ld_f32(const uint8_t* p) { float v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void st_u16(uint8_t* p, uint32_t v) { uint16_t h = (uint16_t)v; __builtin_memcpy(p, &h, 2); }
__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void st_f32(uint8_t* p, float v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ v3 ld_v3(const uint8_t* p) { v3 v; __builtin_memcpy(&v, p, 12); return v; }
// 16-byte load from a 4-byte aligned float address
struct __attribute__((packed, aligned(4))) f4_u { float x, y, z, w; };
__device__ __forceinline__ float4 ld_f4(const float* p) { const f4_u t = *reinterpret_cast<const f4_u*>(p); return float4{t.x, t.y, t.z, t.w}; }

// ---- wave-level primitives -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}
// exclusive prefix sum over the wavefront on the DPP network: Hillis-Steele inside each row of 16 (zeros shift in), then the
// row totals travel with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3). Six dependent VALU instructions instead of
// six LDS-crossbar round trips (shuffles)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32_or0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_excl_scan_dpp(uint32_t x, uint32_t* total) {
    uint32_t v = x;
    v += dpp_u32_or0<0x111, 0xf>(v);   // row_shr:1
    v += dpp_u32_or0<0x112, 0xf>(v);   // row_shr:2
    v += dpp_u32_or0<0x114, 0xf>(v);   // row_shr:4
    v += dpp_u32_or0<0x118, 0xf>(v);   // row_shr:8
    v += dpp_u32_or0<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_u32_or0<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    *total = (uint32_t)__builtin_amdgcn_readlane((int)v, WAVE - 1);
    return v - x;
}
// every call site runs with all 64 lanes active (wave-uniform control flow); `lane` is kept for the callers' signature
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t* total) { (void)lane; return wave_excl_scan_dpp(v, total); }


// running extremum with std::min_element / std::max_element semantics (first occurrence wins,
// reference src/discretizer.cpp:27-28): order by value, then by position
struct ext { float v; uint32_t i; };
__device__ __forceinline__ void ext_min_upd(ext& a, float v, uint32_t i) {
    if (v < a.v || (v == a.v && i < a.i)) { a.v = v; a.i = i; }
}
__device__ __forceinline__ void ext_max_upd(ext& a, float v, uint32_t i) {
    if (a.v < v || (v == a.v && i < a.i)) { a.v = v; a.i = i; }
}
__device__ __forceinline__ float wave_ext_min(ext a) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        float v = __shfl_xor(a.v, d, WAVE); uint32_t i = __shfl_xor(a.i, d, WAVE);
        ext_min_upd(a, v, i);
    }
    return a.v;
}
__device__ __forceinline__ float wave_ext_max(ext a) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        float v = __shfl_xor(a.v, d, WAVE); uint32_t i = __shfl_xor(a.i, d, WAVE);
        ext_max_upd(a, v, i);
    }
    return a.v;
}

__device__ __forceinline__ bool res_code_ok(uint32_t rc) { return rc < 20u || rc == 23u; }

// FCZ record geometry (reference Foldcomp::getSize, src/foldcomp.cpp:1190-1214)
struct rec_layout {
    uint32_t o_aidx, o_title, o_anchor, o_oxt, o_words, o_sc, o_tmp, o_tbytes, size;
};
__host__ __device__ __forceinline__ rec_layout make_layout(uint32_t n_res, uint32_t n_anchor, uint32_t title_len, uint32_t n_sc) {
    rec_layout L;
    L.o_aidx = 76;
    L.o_title = 76 + 4 * n_anchor;
    L.o_anchor = L.o_title + title_len;
    L.o_oxt = L.o_anchor + 36 * n_anchor;
    L.o_words = L.o_oxt + 13;
    L.o_sc = L.o_words + 8 * n_res;
    L.o_tmp = L.o_sc + n_sc;
    L.o_tbytes = L.o_tmp + 8;
    L.size = L.o_tbytes + n_res;
    return L;
}

// ==================================================================================================
// compress
// ==================================================================================================

// ---- sub-wavefront groups: GROUP lanes (one DPP row) per chain / entry. The per-chain bookkeeping kernels are chains of
// dependent loads (offset -> header -> arrays) over a few hundred bytes: with one wavefront per chain a SIMD has 8 chains in
// flight and waits; with a row per chain it has 32, and a row's load of 16 consecutive words is still one whole cache line.
constexpr int GROUP = 16;
constexpr int GROUPS_PER_BLOCK = BLOCK / GROUP;
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
    for (int d = GROUP / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, GROUP);
    return v;
}
__device__ __forceinline__ uint32_t group_max(uint32_t v) {
#pragma unroll
    for (int d = GROUP / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(v, d, GROUP); v = o > v ? o : v; }
    return v;
}

// FCZ size of every chain (Foldcomp::getSize). One 16-lane group per chain (coalesced residue-code reads).
__global__ __launch_bounds__(BLOCK) void k_compress_sizes(fcz_chain_batch in, uint64_t* __restrict__ sizes) {
    const int sub = threadIdx.x & (GROUP - 1);
    const uint32_t c = blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
    const bool live = c < in.n_chains;
    const uint32_t r0 = live ? in.res_off[c] : 0u, n = live ? in.res_off[c + 1] - r0 : 0u;
    uint32_t nsc = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += 8 * GROUP) {   // eight independent loads in flight per memory round trip
        uint32_t rcs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t k = k0 + u * GROUP + sub; rcs[u] = in.res_code[r0 + (k < n ? k : n - 1)]; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (k0 + u * GROUP + sub < n) nsc += fcz_res_natoms[rcs[u] < 24 ? rcs[u] : 23] - 3;
    }
    nsc = group_sum(nsc);
    if (live && sub == 0) {
        const uint32_t n_anchor = n / (uint32_t)in.anchor_threshold + 2;
        sizes[c] = make_layout(n, n_anchor, in.title_off[c + 1] - in.title_off[c], nsc).size;
    }
}

// ---- multi-block exclusive scan: per-chunk sums -> single-block scan of the sums -> per-chunk rescan ----
constexpr int SCAN_CHUNK = 4096;   // elements per block (1024 threads x 4)
template <class T>
__global__ __launch_bounds__(1024) void k_scan_reduce(uint32_t n, const T* __restrict__ in, unsigned long long* __restrict__ partial) {
    __shared__ unsigned long long s_w[16];
    const uint32_t base = blockIdx.x * SCAN_CHUNK;
    unsigned long long v = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint32_t i = base + u * 1024 + threadIdx.x; if (i < n) v += (unsigned long long)in[i]; }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < 16; w++) t += s_w[w]; partial[blockIdx.x] = t; }
}
template <class T>
__global__ __launch_bounds__(1024) void k_scan_apply(uint32_t n, const T* __restrict__ in, const unsigned long long* __restrict__ partial_excl,
                                                     T* __restrict__ out, uint32_t* __restrict__ overflow) {
    __shared__ unsigned long long s_w[16];
    const uint32_t base = blockIdx.x * SCAN_CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long carry = partial_excl[blockIdx.x];
    for (int u = 0; u < 4; u++) {
        const uint32_t i = base + u * 1024 + threadIdx.x;
        const unsigned long long v = (i < n) ? (unsigned long long)in[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { unsigned long long t = __shfl_up(inc, d, WAVE); if (lane >= d) inc += t; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned long long pre = carry;
        for (int w = 0; w < wave; w++) pre += s_w[w];
        if (i < n) out[i] = (T)(pre + inc - v);
        unsigned long long tot = 0;
        for (int w = 0; w < 16; w++) tot += s_w[w];
        carry += tot;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        out[n] = (T)carry;
        if (overflow && sizeof(T) < 8 && (carry >> (8 * sizeof(T) < 64 ? 8 * sizeof(T) : 63))) *overflow = 1u;   // the total does not fit T
    }
}

// LDS slot store of one lane's residue: [slot][comp][lane] floats -> conflict-free
struct slot_store {
    float* base;  // this wave's region: 14 * 3 * 64 floats
    int lane;
    __device__ __forceinline__ void put(int slot, v3 p) {
        base[(slot * 3 + 0) * WAVE + lane] = p.x;
        base[(slot * 3 + 1) * WAVE + lane] = p.y;
        base[(slot * 3 + 2) * WAVE + lane] = p.z;
    }
    __device__ __forceinline__ v3 get(int slot) const {
        return v3{base[(slot * 3 + 0) * WAVE + lane], base[(slot * 3 + 1) * WAVE + lane], base[(slot * 3 + 2) * WAVE + lane]};
    }
};

// ==================================================================================================
// decompress
// ==================================================================================================

// Parsed view of one FCZ entry (Foldcomp::read, reference src/foldcomp.cpp:904-1036)
struct entry_view {
    const uint8_t* e;
    uint32_t n, n_anchor, title_len, n_sc;
    rec_layout L;
};
__device__ __forceinline__ entry_view view_entry(const uint8_t* e) {
    entry_view v;
    v.e = e;
    v.n = ld_u16(e + 4);
    v.n_anchor = e[12];
    v.n_sc = ld_u32(e + 16);
    v.title_len = ld_u32(e + 24);
    v.L = make_layout(v.n, v.n_anchor, v.title_len, v.n_sc);
    return v;
}
__device__ __forceinline__ int res_code_from_letter(uint8_t ch) {
    for (int i = 0; i < 24; i++) if ((uint8_t)fcz_res1[i] == ch) return i;
    return 23;
}

// One 16-lane group per entry: validate + count (residues, output atoms, status); seg_info[i] = longest anchor segment << 16 |
// number of segments (0 for a skipped entry); codes[(record offset >> 3) + k] = residue code of residue k (see below)
__global__ __launch_bounds__(BLOCK) void k_entry_sizes(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off,
                                                       uint32_t n_entries, uint32_t* __restrict__ cnt_res,
                                                       uint32_t* __restrict__ cnt_atoms, int32_t* __restrict__ status,
                                                       uint32_t* __restrict__ seg_info, uint8_t* __restrict__ codes, uint64_t codes_cap) {
    const int sub = threadIdx.x & (GROUP - 1);
    const uint32_t i = blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
    const bool live = i < n_entries;
    uint32_t seg_max = 0;
    const uint64_t o0 = live ? off[i] : 0ull;
    const uint8_t* e = blob + o0;
    const uint64_t len = live ? off[i + 1] - o0 : 0ull;
    int st = FCZ_OK;
    uint32_t n = 0, na_total = 0, n_seg = 0;
    // every branch below is decided by the entry, i.e. uniform inside a group: the group reductions run with all its lanes
    uint32_t na = 0, nsc = 0, bad = 0, n_sc_hdr = 0, has_oxt = 0;
    bool counted = false;
    if (!live) st = FCZ_E_TRUNCATED;
    else if (len < 76) st = FCZ_E_TRUNCATED;
    else if (!(e[0] == 'F' && e[1] == 'C' && e[2] == 'M' && e[3] == 'P')) st = FCZ_E_BAD_MAGIC;
    else {
        entry_view v = view_entry(e);
        n = v.n;
        if ((uint64_t)v.L.size > len || v.title_len > len || v.n_sc > len) st = FCZ_E_TRUNCATED;
        else if (n < 2 || v.n_anchor < 2) st = FCZ_E_TOO_SHORT;
        else {
            counted = true; n_sc_hdr = v.n_sc; n_seg = v.n_anchor - 1;
            const uint32_t rc_first = (uint32_t)res_code_from_letter(e[20]);  // header.firstResidue, src/foldcomp.cpp:863
            has_oxt = e[v.L.o_oxt] ? 1u : 0u;
            for (uint32_t k0 = 0; k0 < n; k0 += 8 * GROUP) {   // eight independent loads in flight per memory round trip
                uint32_t wb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const uint32_t k = k0 + u * GROUP + sub; wb[u] = e[v.L.o_words + 8 * (size_t)(k < n ? k : n - 1)]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + u * GROUP + sub;
                    if (k >= n) continue;
                    uint32_t rc = k == 0 ? rc_first : (wb[u] >> 3);
                    if (rc >= 24) rc = 23;
                    if (!res_code_ok(rc)) bad |= 1u;
                    na += fcz_res_natoms[rc]; nsc += fcz_res_natoms[rc] - 3;
                    // the residue codes, one byte each, where k_res_index finds them without walking the 8-byte words again: a record
                    // holds at least 8 bytes per residue, so slot (record offset / 8) + k is this residue's alone
                    const uint64_t ci = (o0 >> 3) + k;
                    if (ci < codes_cap) codes[ci] = (uint8_t)rc;
                }
            }
            // anchor indices must be usable as segment bounds
            for (uint32_t s = sub; s + 1 < v.n_anchor; s += GROUP) {
                int a = (int)ld_u32(e + v.L.o_aidx + 4 * s), b = (int)ld_u32(e + v.L.o_aidx + 4 * (s + 1));
                if (a < 0 || b < a || b > (int)n - 1) bad |= 2u;
                if (s == 0 && a != 0) bad |= 2u;
                if (s + 2 == v.n_anchor && b != (int)n - 1) bad |= 2u;
                if (b >= a && (uint32_t)(b - a + 1) > seg_max) seg_max = (uint32_t)(b - a + 1);
            }
        }
    }
    // (whole wavefront here: groups that skipped the counting carry zeros)
    na = group_sum(na); nsc = group_sum(nsc);
    const uint32_t bad1 = group_max(bad & 1u), bad2 = group_max(bad & 2u);
    seg_max = group_max(seg_max);
    if (counted) {
        if (bad1) st = FCZ_E_RESIDUE;
        else if (bad2 || nsc != n_sc_hdr) st = FCZ_E_TRUNCATED;
        else na_total = na + has_oxt;
    }
    if (live && sub == 0) {
        const bool ok = st == FCZ_OK;
        cnt_res[i] = ok ? n : 0; cnt_atoms[i] = ok ? na_total : 0;
        status[i] = st;
        // longest segment and segment count of the entry; k_sizes_reduce reduces them over the batch (a global atomic per entry
        // serialises on its address: 2.2 ms per 100 000 entries of mixed length)
        seg_info[i] = ok ? ((seg_max < 0xffffu ? seg_max : 0xffffu) << 16) | (n_seg < 0xffffu ? n_seg : 0xffffu) : 0u;
    }
}

// uint64 exclusive scan (FCZ byte offsets), single block
__global__ __launch_bounds__(1024) void k_scan_u64(uint32_t n, const uint64_t* in, uint64_t* out) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        unsigned long long v = i < n ? in[i] : 0ull, inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            unsigned long long t = __shfl_up(inc, d, WAVE);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned long long pre = s_carry;
        for (int w = 0; w < wave; w++) pre += s_w[w];
        if (i < n) out[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = s_carry; for (int w = 0; w < 16; w++) t += s_w[w]; s_carry = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = s_carry;
}

// unpacked + de-quantised backbone word (convertBytesToBackboneChain src/foldcomp.cpp:60-77,
// decompressBackboneChain :122-153)
struct bb_word { float phi, psi, omega, nca, can, cna; uint32_t res; };
struct bb_params { float mn[6], cf[6]; };
__device__ __forceinline__ bb_params load_params(const uint8_t* e) {
    bb_params P;
#pragma unroll
    for (int q = 0; q < 6; q++) { P.mn[q] = ld_f32(e + 28 + 4 * q); P.cf[q] = ld_f32(e + 52 + 4 * q); }
    return P;
}
__device__ __forceinline__ bb_word decode_word(uint64_t raw, const bb_params& P) {
    const uint32_t b0 = (uint32_t)raw & 0xffu, b1 = (uint32_t)(raw >> 8) & 0xffu, b2 = (uint32_t)(raw >> 16) & 0xffu,
                   b3 = (uint32_t)(raw >> 24) & 0xffu, b4 = (uint32_t)(raw >> 32) & 0xffu;
    bb_word r;
    r.res = b0 >> 3;
    const uint32_t om = ((b0 & 7u) << 8) | b1, ps = (b2 << 4) | (b3 >> 4), ph = ((b3 & 0xfu) << 8) | b4;
    r.phi = dequant(ph, P.mn[0], P.cf[0]);
    r.psi = dequant(ps, P.mn[1], P.cf[1]);
    r.omega = dequant(om, P.mn[2], P.cf[2]);
    r.nca = dequant((uint32_t)(raw >> 56) & 0xffu, P.mn[3], P.cf[3]);
    r.can = dequant((uint32_t)(raw >> 40) & 0xffu, P.mn[4], P.cf[4]);
    r.cna = dequant((uint32_t)(raw >> 48) & 0xffu, P.mn[5], P.cf[5]);
    return r;
}

// ---- chains ordered by length (longest first) for the lane = chain kernel ---------------------------------------
// k_backbone walks 64 chains per wavefront in lock step, so a wavefront lasts as long as its longest chain. A counting
// sort on the residue count (16-residue buckets, descending) puts chains of similar length into the same wavefront and
// dispatches the long ones first; with a uniform batch it degenerates to blocks of 64 consecutive chains. One atomic
// per distinct bucket per wavefront (a contended atomic per chain would serialise at ~88 per microsecond).
constexpr int LEN_BUCKETS = 4096;
constexpr uint32_t FCZ_LONG_CHAIN = 1024;   // residues; a multiple of the 16-residue bucket width (see k_backbone MODE 1/2)
__host__ __device__ __forceinline__ uint32_t len_bucket(uint32_t n_res) {
    const uint32_t b = n_res >> 4;
    return (uint32_t)LEN_BUCKETS - 1u - (b < (uint32_t)LEN_BUCKETS ? b : (uint32_t)LEN_BUCKETS - 1u);
}
// ---- after k_entry_sizes: offsets, totals and the length order in three launches -----------------------------------------------
// (round 3 ran 13: two three-launch scans, a segment-maximum reduction, and a two-pass counting sort whose wavefront-aggregated
// atomics still serialised -- a uniform batch puts one atomic per wavefront on ONE address, 15 625 of them at ~88 per
// microsecond = the 0.18 ms each pass took.) Chunks of 4 096 entries per block:
//   k_sizes_reduce  chunk sums of residue and atom counts (scan partials), batch maxima of segment length / count, and the
//                   chunk's length histogram built in LDS -- one global atomic per bucket the chunk actually holds;
//   k_sizes_mid     one block: exclusive scans of the chunk sums and of the histogram (bucket cursors), the totals word
//                   {residues, atoms, -, longest segment, most segments, long chains, overflow};
//   k_sizes_apply   per chunk: the offsets (block scan + chunk carry) and the scatter of the length order -- LDS histogram again,
//                   one global atomic per held bucket hands the chunk its run of the bucket, ranks inside the run from LDS.
constexpr int SZ_CHUNK = 4096;
struct sizes_totals { uint32_t residues, atoms, codes_lo, max_seg, max_nseg, n_long, overflow, codes_hi; };

__global__ __launch_bounds__(1024) void k_sizes_reduce(const uint32_t* __restrict__ cnt_res, const uint32_t* __restrict__ cnt_atoms,
                                                       const uint32_t* __restrict__ seg_info, uint32_t n,
                                                       unsigned long long* __restrict__ part_res, unsigned long long* __restrict__ part_atoms,
                                                       uint32_t* __restrict__ hist, uint32_t* __restrict__ maxseg) {
    __shared__ uint32_t s_h[LEN_BUCKETS];
    __shared__ unsigned long long s_r[16], s_a[16];
    __shared__ uint32_t s_ms[16], s_mn[16];
    for (int k = threadIdx.x; k < LEN_BUCKETS; k += 1024) s_h[k] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * SZ_CHUNK;
    unsigned long long sr = 0, sa = 0; uint32_t ms = 0, mn = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t i = base + u * 1024 + threadIdx.x;
        const bool act = i < n;
        const uint32_t r = act ? cnt_res[i] : 0u, a = act ? cnt_atoms[i] : 0u, sg = act ? seg_info[i] : 0u;
        sr += r; sa += a;
        ms = (sg >> 16) > ms ? (sg >> 16) : ms; mn = (sg & 0xffffu) > mn ? (sg & 0xffffu) : mn;
        const uint32_t b = len_bucket(r);
        // (every lane of the wavefront takes part in the ballots)
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        const unsigned long long all = __ballot(act), same = __ballot(act && b == b0);
        if (all != 0ull && same == all && act) { if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(all)) atomicAdd(&s_h[b], (uint32_t)__builtin_popcountll(all)); }
        else if (act) atomicAdd(&s_h[b], 1u);
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
        sr += __shfl_xor(sr, d, WAVE); sa += __shfl_xor(sa, d, WAVE);
        const uint32_t o1 = __shfl_xor(ms, d, WAVE), o2 = __shfl_xor(mn, d, WAVE);
        ms = o1 > ms ? o1 : ms; mn = o2 > mn ? o2 : mn;
    }
    if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6] = sr; s_a[threadIdx.x >> 6] = sa; s_ms[threadIdx.x >> 6] = ms; s_mn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { sr += s_r[w]; sa += s_a[w]; ms = s_ms[w] > ms ? s_ms[w] : ms; mn = s_mn[w] > mn ? s_mn[w] : mn; }
        part_res[blockIdx.x] = sr; part_atoms[blockIdx.x] = sa;
        if (ms) atomicMax(maxseg, ms);
        if (mn) atomicMax(maxseg + 1, mn);
    }
    for (int k = threadIdx.x; k < LEN_BUCKETS; k += 1024) { const uint32_t v = s_h[k]; if (v) atomicAdd(&hist[k], v); }
}

// block-wide exclusive scan of one value per thread (1 024 threads); *total = the sum. Two barriers.
__device__ __forceinline__ unsigned long long block_excl_scan_1024(unsigned long long v, unsigned long long* s_w, unsigned long long* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, WAVE); if (lane >= d) inc += t; }
    __syncthreads();                         // (s_w may still be read from the previous call)
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned long long pre = 0, tot = 0;
    for (int w = 0; w < 16; w++) { const unsigned long long x = s_w[w]; if (w < wave) pre += x; tot += x; }
    *total = tot;
    return pre + inc - v;
}

__global__ __launch_bounds__(1024) void k_sizes_mid(uint32_t nb, unsigned long long* __restrict__ part_res, unsigned long long* __restrict__ part_atoms,
                                                    const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
                                                    const uint32_t* __restrict__ maxseg, sizes_totals* __restrict__ totals,
                                                    const uint64_t* __restrict__ off_end) {
    __shared__ unsigned long long s_w[16];
    unsigned long long carry_r = 0, carry_a = 0, tot;
    for (uint32_t base = 0; base < nb; base += 1024) {            // the chunk sums, in place: sum -> sum of the chunks before
        const uint32_t i = base + threadIdx.x;
        const unsigned long long vr = i < nb ? part_res[i] : 0ull, va = i < nb ? part_atoms[i] : 0ull;
        const unsigned long long er = block_excl_scan_1024(vr, s_w, &tot); if (i < nb) part_res[i] = carry_r + er; carry_r += tot;
        const unsigned long long ea = block_excl_scan_1024(va, s_w, &tot); if (i < nb) part_atoms[i] = carry_a + ea; carry_a += tot;
    }
    // bucket cursors: four consecutive buckets per thread
    uint32_t h[4]; unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { h[k] = hist[4 * threadIdx.x + k]; mine += h[k]; }
    unsigned long long ex = block_excl_scan_1024(mine, s_w, &tot);
    const uint32_t long_b = len_bucket(FCZ_LONG_CHAIN);            // chains of FCZ_LONG_CHAIN residues or more lead the order: buckets 0 .. long_b
#pragma unroll
    for (int k = 0; k < 4; k++) {
        cursor[4 * threadIdx.x + k] = (uint32_t)ex; ex += h[k];
        if (4 * threadIdx.x + k == long_b) totals->n_long = (uint32_t)ex;
    }
    if (threadIdx.x == 0) {
        cursor[LEN_BUCKETS] = (uint32_t)tot;
        totals->residues = (uint32_t)carry_r; totals->atoms = (uint32_t)carry_a;
        const uint64_t need = (*off_end >> 3) + 1;                  // slots of the residue-code array (k_entry_sizes)
        totals->codes_lo = (uint32_t)need; totals->codes_hi = (uint32_t)(need >> 32);
        totals->max_seg = maxseg[0]; totals->max_nseg = maxseg[1];
        // offsets are 32-bit: a batch whose residues or atoms reach 2^32 is refused, not wrapped (the sums are 64-bit)
        totals->overflow = ((carry_r >> 32) || (carry_a >> 32)) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(1024) void k_sizes_apply(const uint32_t* __restrict__ cnt_res, const uint32_t* __restrict__ cnt_atoms, uint32_t n,
                                                      const unsigned long long* __restrict__ part_res, const unsigned long long* __restrict__ part_atoms,
                                                      uint32_t* __restrict__ res_off, uint32_t* __restrict__ atom_off,
                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm) {
    __shared__ uint32_t s_h[LEN_BUCKETS];      // the chunk's entries per bucket, then the next free slot of the chunk's run
    __shared__ unsigned long long s_w[16];
    for (int k = threadIdx.x; k < LEN_BUCKETS; k += 1024) s_h[k] = 0;
    const uint32_t base = blockIdx.x * SZ_CHUNK;
    unsigned long long carry_r = part_res[blockIdx.x], carry_a = part_atoms[blockIdx.x], tot;
    uint32_t r[4];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t i = base + u * 1024 + threadIdx.x;
        const bool act = i < n;
        r[u] = act ? cnt_res[i] : 0u;
        const uint32_t a = act ? cnt_atoms[i] : 0u;
        const unsigned long long er = block_excl_scan_1024(r[u], s_w, &tot); if (act) res_off[i] = (uint32_t)(carry_r + er); carry_r += tot;
        if (atom_off) { const unsigned long long ea = block_excl_scan_1024(a, s_w, &tot); if (act) atom_off[i] = (uint32_t)(carry_a + ea); carry_a += tot; }
        const uint32_t b = len_bucket(r[u]);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        const unsigned long long all = __ballot(act), same = __ballot(act && b == b0);
        if (all != 0ull && same == all && act) { if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(all)) atomicAdd(&s_h[b], (uint32_t)__builtin_popcountll(all)); }
        else if (act) atomicAdd(&s_h[b], 1u);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { res_off[n] = (uint32_t)carry_r; if (atom_off) atom_off[n] = (uint32_t)carry_a; }
    __syncthreads();
    // the chunk's run of every bucket it holds: count -> first slot
    for (int k = threadIdx.x; k < LEN_BUCKETS; k += 1024) { const uint32_t v = s_h[k]; if (v) s_h[k] = atomicAdd(&cursor[k], v); }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t i = base + u * 1024 + threadIdx.x;
        const bool act = i < n;
        const uint32_t b = len_bucket(r[u]);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        const unsigned long long all = __ballot(act), same = __ballot(act && b == b0);
        if (all != 0ull && same == all && act) {
            // the whole wavefront in one bucket: one LDS atomic, ranks by lane order
            const int leader = __builtin_ctzll(all);
            uint32_t slot = 0;
            if ((int)(threadIdx.x & 63) == leader) slot = atomicAdd(&s_h[b], (uint32_t)__builtin_popcountll(all));
            slot = __shfl(slot, leader, WAVE);
            perm[slot + (uint32_t)__builtin_popcountll(all & ((1ull << (threadIdx.x & 63)) - 1ull))] = i;
        } else if (act) perm[atomicAdd(&s_h[b], 1u)] = i;
    }
}

#ifndef FCZ_BACKBONE_MIN_WAVES
#define FCZ_BACKBONE_MIN_WAVES 2
#endif
#ifndef FCZ_BB_PERSIST
#define FCZ_BB_PERSIST 1
#endif
// Backbone reconstruction, one wavefront per group of 64 consecutive entries, lane = chain.
// Reference: segment loop of Foldcomp::decompress (src/foldcomp.cpp:814-858): per anchor segment a forward
// NeRF (reconstructBackboneAtoms :167-246), then reconstructBackboneReverse (:248-273: bond angles re-measured
// on the forward atoms, Nerf::reconstructWithReversed src/nerf.cpp:342-379 from the next anchor, weightedAverage
// src/atom_coordinate.cpp:145-163); the next segment starts from the blended last three atoms (:855-857).
// The forward atoms of the current segment live in a per-group ring [atom][lane] (every lane only ever
// re-reads its own column, the transposed layout is purely for coalescing: one 768-byte row per step), so the
// intermediate never makes a strided trip through HBM. A second ring carries cos/sin of the three torsions of every
// word: the reverse pass places its atoms with the same torsions, so it neither decodes the word nor repeats the trig. bb receives the final backbone (3 atoms per residue,
// chain-major).
// window of blended backbone atoms per lane, flushed with wave-cooperative contiguous stores
constexpr int BW = 16;                      // atoms per lane and window
struct backbone_lds {
    float atom[WAVE][BW * 3 + 3];           // [lane][slot*3+comp], odd dword stride: conflict-free
    unsigned long long base[WAVE];          // byte address of slot 0 of the lane's window (chain-major bb)
    int lo[WAVE], hi[WAVE];                 // filled slots [lo, hi)
};

// MODE 0 (the default): forward and reverse pass of every segment fused as described above; one ring slot per group.
// A lone wavefront needs ~8 us per residue in that form, so a 2 700-residue chain would hold the whole launch for 22 ms.
// Chains beyond FCZ_LONG_CHAIN residues therefore take two launches instead:
// MODE 1: forward pass only (the serial part: every segment starts from the carry of the previous one), forward atoms and
//         torsion trig of segment s parked in ring slot (group, s);
// MODE 2: one block per (group, segment): reverse pass + blend of that segment from its slot -- the segments of a chain
//         run side by side.
#ifdef FCZ_BB_TIMING
// measurement aid (not built into the product): wavefront-cycles in the parts of k_backbone
__device__ unsigned long long g_bb_timing[8];
#define BB_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define BB_STAMP(i)
#endif
// Measurement builds (results wrong by design, the time is the answer; profiles/r6_ab_ring_ablation.txt): the same instruction stream
// with the ring rows folded into a window of a few rows per wavefront that stays in the L2 -- FCZ_ABL_NO_RING: forward atoms and
// torsion trig; FCZ_ABL_NO_TRING: the trig ring only. What k_backbone would gain if its 120 B/residue of ring traffic cost nothing.
#if defined(FCZ_ABL_NO_RING)
#define BB_RROW(x) ((x) & 7)
#define BB_TROW(x) ((x) & 15)
#elif defined(FCZ_ABL_NO_TRING)
#define BB_RROW(x) (x)
#define BB_TROW(x) ((x) & 15)
#else
#define BB_RROW(x) (x)
#define BB_TROW(x) (x)
#endif
// one group of 64 chains (lane = chain) of k_backbone; `home` = the ring slot of a MODE 0 wavefront (its block index: the grid is
// persistent, so the ring is as large as the wavefronts in flight, not as the batch)
template <int MODE>
__device__ __forceinline__ void backbone_group(
        backbone_lds& S, const uint32_t grp, const uint32_t seg_only, const uint32_t home,
        const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries, uint32_t n_slots,
        const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ perm, v3* __restrict__ ring,
        float* __restrict__ tring, uint32_t ring_rows, uint32_t seg_slots, v3* __restrict__ bb) {
    const int lane = threadIdx.x;
#ifdef FCZ_BB_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    const uint32_t slot = grp * WAVE + lane;
    const uint32_t c = slot < n_slots ? perm[slot] : n_entries;   // chains grouped by length, longest first
    const bool valid = c < n_entries && res_off[c + 1] != res_off[c];
    // ring slot of segment sg: atom row j at Rg[j * WAVE]; cos/sin of the three torsions of word i at Tg rows 6i .. 6i+5
    // (the reverse pass needs nothing else of the word)
    auto ring_slot = [&](uint32_t sg) -> size_t { return (MODE == 0) ? (size_t)home : (size_t)grp * seg_slots + sg; };
    v3* Rg = ring + ring_slot(seg_only) * ring_rows * WAVE + lane;
    float* Tg = tring + ring_slot(seg_only) * (ring_rows / 3) * 6 * WAVE + lane;
    const uint8_t* e = blob + (valid ? off[c] : off[0]);
    entry_view v; v.n = 0; v.n_anchor = 1; v.e = e; v.L = make_layout(0, 0, 0, 0);
    bb_params P{};
    if (valid) { v = view_entry(e); if (MODE != 2) P = load_params(e); }
    // can the parameters make an angle whose radians reach glibc's large-argument reduction (|x| >= 120 = 6 875 degrees)? An angle
    // is minimum + q * step with q in [0, bins]: its bounds are the two ends. Anything not certainly below 6 000 degrees -- NaN and
    // infinite parameters with it -- sends the wavefront through the sine / cosine that takes any float
    bool wild = false;
    if (MODE != 2 && valid) {
        const float bins[6] = {4095.0f, 4095.0f, 2047.0f, 255.0f, 255.0f, 255.0f};
#pragma unroll
        for (int q = 0; q < 6; q++) wild = wild || !(__builtin_fabsf(P.mn[q]) < 6000.0f && __builtin_fabsf(P.mn[q] + bins[q] * P.cf[q]) < 6000.0f);
    }
    const bool any_wild = MODE != 2 && __any(wild);
    const uint8_t* words = e + v.L.o_words;
    const uint8_t* last_word = words + 8 * (size_t)(v.n ? v.n - 1 : 0);
    const uint32_t nseg = valid ? v.n_anchor - 1 : 0;
    uint32_t maxseg = nseg;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(maxseg, d, WAVE); maxseg = o > maxseg ? o : maxseg; }
    if (MODE == 2 && seg_only >= maxseg) return;     // no chain of the group has this segment (uniform)
    v3* Bc = bb + 3 * (size_t)(valid ? res_off[c] : 0);
    v3 p0{0.f, 0.f, 0.f}, p1 = p0, p2 = p0;
    int first = 0, next = 0;
    const uint8_t* wp = words;
    uint64_t w_cur = 0, w_nxt = 0;
    if (MODE != 2) {
        if (valid) {
            p0 = ld_v3(e + v.L.o_anchor); p1 = ld_v3(e + v.L.o_anchor + 12); p2 = ld_v3(e + v.L.o_anchor + 24);
            first = (int)ld_u32(e + v.L.o_aidx); next = (int)ld_u32(e + v.L.o_aidx + 4);
            wp = words + 8 * (size_t)first;
            w_cur = ld_u64(wp);
            w_nxt = ld_u64(wp + 8 <= last_word ? wp + 8 : last_word);
        }
    } else if (valid && seg_only < nseg) {
        // the segment's bounds come straight from the anchor index list; its forward atoms wait in the ring slot
        first = (int)ld_u32(e + v.L.o_aidx + 4 * (size_t)seg_only); next = (int)ld_u32(e + v.L.o_aidx + 4 * (size_t)(seg_only + 1));
        const int len0 = next - first + 1;
        p0 = Rg[(size_t)(3 * len0 - 3) * WAVE]; p1 = Rg[(size_t)(3 * len0 - 2) * WAVE]; p2 = Rg[(size_t)(3 * len0 - 1) * WAVE];
    }
    // ---- output window: blended atoms are parked in LDS and leave as contiguous runs per chain ----
    long long win = -1;           // window index (chain-major atom index / BW) this lane is filling, -1 = none
    int w_lo = 0, w_hi = 0;       // filled slots [w_lo, w_hi) of that window (registers; published to LDS at a flush)
    auto flush = [&]() {
        BB_STAMP(3)
        S.lo[lane] = w_lo; S.hi[lane] = w_hi;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int t = 0; t < BW; t++) {                   // 64 lanes x 16 rounds cover 64 chains x 16 atoms, an atom per lane
            const int flat = t * WAVE + lane;
            const int cl = flat / BW, slot = flat - cl * BW;
            if (slot >= S.lo[cl] && slot < S.hi[cl]) {
                const v3 a{S.atom[cl][3 * slot], S.atom[cl][3 * slot + 1], S.atom[cl][3 * slot + 2]};
                *reinterpret_cast<v3*>(S.base[cl] + 12ull * (unsigned)slot) = a;
            }
        }
        __builtin_amdgcn_wave_barrier();
        w_lo = 0; w_hi = 0;
        win = -1;
        BB_STAMP(4)
    };
    auto emit = [&](bool on, long long bi, v3 a) {       // every lane calls (wave-uniform control flow)
        const long long w = on ? bi / BW : win;
        if (__any(on && win >= 0 && w != win)) flush();
        if (on) {
            const int slot = (int)(bi - w * BW);
            if (win < 0) { win = w; S.base[lane] = (unsigned long long)(Bc + w * BW); w_lo = slot; w_hi = slot + 1; }
            else { w_lo = slot < w_lo ? slot : w_lo; w_hi = slot + 1 > w_hi ? slot + 1 : w_hi; }
            S.atom[lane][3 * slot] = a.x; S.atom[lane][3 * slot + 1] = a.y; S.atom[lane][3 * slot + 2] = a.z;
        }
    };
    const uint32_t s_begin = (MODE == 2) ? seg_only : 0u, s_end = (MODE == 2) ? seg_only + 1 : maxseg;
    BB_STAMP(0)
    for (uint32_t s = s_begin; s < s_end; s++) {
        const bool act = s < nseg;
        const int len = act ? next - first + 1 : 0;
        if (MODE == 1) {
            Rg = ring + ring_slot(s) * ring_rows * WAVE + lane;
            Tg = tring + ring_slot(s) * (ring_rows / 3) * 6 * WAVE + lane;
        }
        int maxlen = len;
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) { const int o = __shfl_xor(maxlen, d, WAVE); maxlen = o > maxlen ? o : maxlen; }
        int next2 = next;
        v3 A0{0.f, 0.f, 0.f}, A1 = A0, A2 = A0;
        if (act) {
            next2 = (int)ld_u32(e + v.L.o_aidx + 4 * (size_t)(s + 2 <= nseg ? s + 2 : nseg));
            const uint8_t* anc = e + v.L.o_anchor + 36 * (size_t)(s + 1);
            A0 = ld_v3(anc); A1 = ld_v3(anc + 12); A2 = ld_v3(anc + 24);   // next anchor: carry + reverse start
            if (MODE != 2) { Rg[0] = p0; Rg[WAVE] = p1; Rg[2 * WAVE] = p2; }
        }
        BB_STAMP(1)
        // ---- forward NeRF of the segment ----
        // (ANY: a chain of the wavefront has quantiser parameters that can make an angle of 120 radians or more -- a record no
        //  compressor writes --: the same step with the sine / cosine that takes any float, any_wild is wave-uniform)
        auto forward_step = [&](int i, auto any_tag) {
            constexpr bool ANY = decltype(any_tag)::value;
            const uint8_t* pf = wp + 16;
            const uint64_t w_pre = ld_u64(pf <= last_word ? pf : last_word);
            const bb_word w = decode_word(w_cur, P);
            float s_psi, c_psi, s_om, c_om, s_phi, c_phi;
            if (ANY) { sincosf_pair_any(deg2rad(w.psi), &s_psi, &c_psi); sincosf_pair_any(deg2rad(w.omega), &s_om, &c_om); sincosf_pair_any(deg2rad(w.phi), &s_phi, &c_phi); }
            else { sincosf_pair(deg2rad(w.psi), &s_psi, &c_psi); sincosf_pair(deg2rad(w.omega), &s_om, &c_om); sincosf_pair(deg2rad(w.phi), &s_phi, &c_phi); }
            float* Tw = Tg + (size_t)BB_TROW(6 * i) * WAVE;
            Tw[0] = c_psi; Tw[WAVE] = s_psi; Tw[2 * WAVE] = c_om; Tw[3 * WAVE] = s_om; Tw[4 * WAVE] = c_phi; Tw[5 * WAVE] = s_phi;
            const v3 N = place_atom_d2(p0, p1, p2, nerf_d2_trig_t<ANY>((float)1.3311, w.can, c_psi, s_psi));
            const float l_nca = (w.res != FCZ_RES_PRO) ? (float)1.4581 : (float)1.353;  // src/foldcomp.cpp:204-212
            const v3 CA = place_atom_d2(p1, p2, N, nerf_d2_trig_t<ANY>(l_nca, w.cna, c_om, s_om));
            const v3 C = place_atom_d2(p2, N, CA, nerf_d2_trig_t<ANY>((float)1.5281, w.nca, c_phi, s_phi));
            Rg[(size_t)BB_RROW(3 * i + 3) * WAVE] = N; Rg[(size_t)BB_RROW(3 * i + 4) * WAVE] = CA; Rg[(size_t)BB_RROW(3 * i + 5) * WAVE] = C;
            p0 = N; p1 = CA; p2 = C;
            w_cur = w_nxt; w_nxt = w_pre; wp += 8;
        };
        for (int i = 0; MODE != 2 && i + 1 < maxlen; i++) {
            if (i + 1 >= len) continue;
            if (__builtin_expect(any_wild, 0)) forward_step(i, std::true_type{}); else forward_step(i, std::false_type{});
        }
        BB_STAMP(2)
        // ---- reverse NeRF + blend of the same segment ----
        const int T = 3 * len;
        const float Tf = (float)T;
        const long long b0 = 3ll * first;               // chain-major index of the segment's atom 0
        v3 r3 = A2, r2 = A1, r1 = A0;                   // R[T-1], R[T-2], R[T-3]: the anchor itself
        const float j0 = (float)(T - 3), j1 = (float)(T - 2), j2 = (float)(T - 1);
        // weightedAverage (src/atom_coordinate.cpp:157-159): three divisions by the same T per atom -> vdiv3 (one reciprocal
        // per segment once the compiler has hoisted it; results identical to '/')
        const v3 c0 = vdiv3(v3{(p0.x * 3.0f) + (A0.x * j0), (p0.y * 3.0f) + (A0.y * j0), (p0.z * 3.0f) + (A0.z * j0)}, Tf);
        const v3 c1 = vdiv3(v3{(p1.x * 2.0f) + (A1.x * j1), (p1.y * 2.0f) + (A1.y * j1), (p1.z * 2.0f) + (A1.z * j1)}, Tf);
        const v3 c2 = vdiv3(v3{(p2.x * 1.0f) + (A2.x * j2), (p2.y * 1.0f) + (A2.y * j2), (p2.z * 1.0f) + (A2.z * j2)}, Tf);
        if (MODE != 1) {
            // only the last segment keeps its final three atoms (src/foldcomp.cpp:847-851)
            const bool fin = act && (s + 1 == nseg);
            emit(fin, b0 + T - 1, c2); emit(fin, b0 + T - 2, c1); emit(fin, b0 + T - 3, c0);
        }
        // forward atoms f+2, f+1 for f = T-4 are the last-but-one and last-but-two forward atoms = p1, p0
        v3 f2 = p1, f1 = p0;
        int wi0 = len - 2;
        float tq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // cos, sin of psi, omega, phi of the current word
        v3 fa{0.f, 0.f, 0.f}, fb = fa, fc = fa;
        if (MODE != 1 && wi0 >= 0) {
#pragma unroll
            for (int u = 0; u < 6; u++) tq[u] = Tg[(size_t)(BB_TROW(6 * wi0) + u) * WAVE];
            fa = Rg[(size_t)BB_RROW(3 * wi0 + 2) * WAVE]; fb = Rg[(size_t)BB_RROW(3 * wi0 + 1) * WAVE]; fc = Rg[(size_t)BB_RROW(3 * wi0) * WAVE];
        }
        for (int wi = maxlen - 2; MODE != 1 && wi >= 0; wi--) {       // wave-uniform trip count; lanes join when wi <= len-2
            const bool on = wi <= wi0;
            float tn[6];
#pragma unroll
            for (int u = 0; u < 6; u++) tn[u] = tq[u];
            v3 na = fa, nb = fb, nc = fc;
            if (on) {
                const int wn = wi > 0 ? wi - 1 : 0;
#pragma unroll
                for (int u = 0; u < 6; u++) tn[u] = Tg[(size_t)(BB_TROW(6 * wn) + u) * WAVE];
                na = Rg[(size_t)BB_RROW(3 * wn + 2) * WAVE]; nb = Rg[(size_t)BB_RROW(3 * wn + 1) * WAVE]; nc = Rg[(size_t)BB_RROW(3 * wn) * WAVE];
            }
#pragma unroll
            for (int q = 2; q >= 0; q--) {
                const int f = 3 * wi + q;
                const v3 f0 = (q == 2) ? fa : (q == 1) ? fb : fc;
                v3 Bv{0.f, 0.f, 0.f};
                if (on) {
                    const float ba = bond_angle_deg(f0, f1, f2);  // angle at forward atom f+1 (getBondAngles on forward atoms)
                    const float Lb = (q == 0) ? 1.4581f : (q == 1) ? 1.5281f : 1.3311f;  // src/nerf.h:40-41
                    // torsion of the step: psi, omega, phi for q = 0, 1, 2 (cos at tq[2q], sin at tq[2q+1])
                    const v3 Rv = place_atom_d2(r3, r2, r1, nerf_d2_trig(Lb, ba, tq[2 * q], tq[2 * q + 1]));   // a = R[f+3], b = R[f+2], c = R[f+1]
                    const float wf = (float)(T - f), wr = (float)f;
                    Bv = vdiv3(v3{(f0.x * wf) + (Rv.x * wr), (f0.y * wf) + (Rv.y * wr), (f0.z * wf) + (Rv.z * wr)}, Tf);
                    r3 = r2; r2 = r1; r1 = Rv;
                    f2 = f1; f1 = f0;
                }
                emit(on, b0 + f, Bv);
            }
            if (on) {
#pragma unroll
                for (int u = 0; u < 6; u++) tq[u] = tn[u];
                fa = na; fb = nb; fc = nc;
            }
        }
        if (act) {
            // carry into the next segment: blended last three atoms (indices T-3..T-1)
            p0 = c0; p1 = c1; p2 = c2;
            first = next; next = next2;
        }
        BB_STAMP(3)
    }
    if (MODE != 1 && __any(win >= 0)) flush();
#ifdef FCZ_BB_TIMING
    if (MODE == 0 && lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_bb_timing[i], tacc[i]);
#endif
}

// MODE 0 runs as a persistent grid (FCZ_BB_PERSIST): n_cu x 8 wavefronts, the first round of groups by block index, every later
// one from a counter (groups are in length order, longest first), ring slot = block index. MODE 1 / 2: one block per group /
// (group, segment). n_groups and next_group are only read by MODE 0.
template <int MODE>
__global__ __launch_bounds__(WAVE, FCZ_BACKBONE_MIN_WAVES) void k_backbone(
        const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries, uint32_t n_slots,
        const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ perm, v3* __restrict__ ring,
        float* __restrict__ tring, uint32_t ring_rows, uint32_t seg_slots, v3* __restrict__ bb,
        uint32_t n_groups, uint32_t* __restrict__ next_group) {
    __shared__ backbone_lds S;
    if (MODE == 0 && FCZ_BB_PERSIST) {
        uint32_t grp = blockIdx.x;
        while (grp < n_groups) {
            backbone_group<MODE>(S, grp, 0u, blockIdx.x, blob, off, n_entries, n_slots, res_off, perm, ring, tring, ring_rows, seg_slots, bb);
            uint32_t nx = 0;
            if (threadIdx.x == 0) nx = atomicAdd(next_group, 1u);
            grp = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
        }
    } else {
        const uint32_t grp = (MODE == 2) ? blockIdx.x / seg_slots : blockIdx.x;
        const uint32_t seg_only = (MODE == 2) ? blockIdx.x - grp * seg_slots : 0u;
        backbone_group<MODE>(S, grp, seg_only, grp, blob, off, n_entries, n_slots, res_off, perm, ring, tring, ring_rows, seg_slots, bb);
    }
}

}  // namespace fcz
