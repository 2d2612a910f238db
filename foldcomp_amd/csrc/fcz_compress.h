// fcz_compress.h -- the compress path (Foldcomp::preprocess/compress/writeStream, reference
// src/foldcomp.cpp:450-606, 1038-1109) as three stages.
//
// Every angle of a chain is a function of one residue (side-chain torsions) or of one residue and its successor
// (the six backbone values of a residue window), so the heavy part ignores chain boundaries:
//
//   k_compress_index   one wavefront per chain. res_sc_addr[r] = byte offset, inside the output blob, of residue r's
//                      first side-chain torsion byte; bit 63 marks the last residue of a chain (no window).
//   k_compress_angles  persistent blocks over flat tiles of 256 consecutive residues of the batch (any chains).
//                      The tile's atoms (one contiguous range of the SoA arrays, plus the successor of the last
//                      residue) are parked in LDS as {x, y, z, code} records; one thread per residue records the first
//                      atom of every canonical name (findFirstAtomCoords, reference src/sidechain.cpp:140-147; a missing
//                      name points at an all-zero record, which is what the reference reads for it). Then the angles:
//                      thread = residue window for the 3 dihedrals + 3 bond angles of the backbone, thread = work item
//                      (one dihedral per side-chain atom, taken round-robin from a flat list) for the side chains, so
//                      every round is a full wavefront. The next tile's atoms are prefetched into registers meanwhile and
//                      results stay in registers until the tile is done (no store drains the prefetch). The backbone
//                      values go to the [6][R] scratch `ang` (coalesced per type) as COSINES (enc_torsion_cos below: the
//                      float getCosineTheta returns, dihedrals with their sign bit); side-chain torsions are quantised
//                      (FixedAngleDiscretizer(255), src/foldcomp.cpp:532-538) and stored straight into the FCZ record
//                      (consecutive items = consecutive bytes).
//   k_compress_angles_w  the same stage with wavefront-private 63-residue tiles: what runs on protein input; the kernel
//                      above takes the tiles this one lists (atom-rich stretches, the tail of the arrays).
//   k_compress_pack    one wavefront per chain: validation, acos -> degrees -> float of the six backbone cosines
//                      (src/torsion_angle.cpp:74-94, src/float3d.h:55-65), per-chain quantiser parameters (min/max with
//                      std::min_element semantics, src/discretizer.cpp:22-33), the packed 8-byte words
//                      (src/foldcomp.cpp:582-601, convertBackboneChainToBytes :33-52), B-factor bytes, anchors
//                      (_setAnchor :745-761), OXT (:474-482), title and header (:1038-1109). Chains beyond 128 residues
//                      (4 or 6 rounds of 64 residues in registers by length class; beyond 384 in blocks of 384).
//   k_compress_pack_rows<U>  the same work for chains of up to 128 residues, FOUR to a wavefront: one chain per 16-lane DPP
//                      row, U = 1 / 2 / 4 / 8 rounds of 16 residues by length class -- what a chain costs before its first
//                      residue is shared by the four chains of the wavefront (round 5).
//   Non-finite input (a NaN / infinity in a coordinate of a named atom or a CA B-factor) is found by the angle kernels on the
//   values they stage and refused by the pack kernels (FCZ_E_NONFINITE).
#pragma once
#include "fcz_kernels.h"

namespace fcz {

// ---- wave reductions on the DPP cross-lane network (no LDS traffic) -------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// four DPP steps leave every lane with the extremum of its row of 16; the four rows are combined through scalar reads
__device__ __forceinline__ float wave_min_f32(float v) {
    v = __builtin_fminf(v, dpp_f32<0xB1, 0xf>(v));    // quad_perm [1,0,3,2]
    v = __builtin_fminf(v, dpp_f32<0x4E, 0xf>(v));    // quad_perm [2,3,0,1]
    v = __builtin_fminf(v, dpp_f32<0x141, 0xf>(v));   // row_half_mirror
    v = __builtin_fminf(v, dpp_f32<0x140, 0xf>(v));   // row_mirror
    const int b = __float_as_int(v);   // readlane is an integer intrinsic: move bits, not values
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return __builtin_fminf(__builtin_fminf(r0, r1), __builtin_fminf(r2, r3));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = __builtin_fmaxf(v, dpp_f32<0xB1, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x4E, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x141, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x140, 0xf>(v));
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}

// std::min_element / std::max_element keep the FIRST of equal elements (reference src/discretizer.cpp:27-28).
// Equal floats with different bits are only +0/-0, so a plain value reduction is exact unless the extremum is a
// zero; only then this (value, index) reduction over the stored values runs.
// ---- hand-over of the backbone angles: angle kernels -> k_compress_pack ---------------------------------------------------
// The angle kernels stop at the cosine, the exact float getCosineTheta returns (reference src/float3d.h:38-43); the double part
// (acos -> degrees -> float, src/torsion_angle.cpp:74-94, src/float3d.h:55-65) runs in k_compress_pack, which waits for memory
// most of its time, on the value it has just loaded. A dihedral also carries its sign bit: |cos| <= 1 leaves bit 30 of the
// float (exponent >= 128) free. acos of a NaN or of |cos| > 1 is NaN, which getTorsionFromXYZ replaces by 180 (cos < 0) or 0
// (:77-84): exactly what acos gives for -1 and +1, so those cosines are stored as -1 / +1. Bond angles have no such guard in
// the reference and keep their cosine as it is.
constexpr uint32_t CK_NEG_BIT = 0x40000000u;
__device__ __forceinline__ float enc_torsion_cos(float ct, bool neg) {
    const float c = (__builtin_fabsf(ct) < 1.0f) ? ct : ((ct < 0.0f) ? -1.0f : 1.0f);
    return __uint_as_float(__float_as_uint(c) | (neg ? CK_NEG_BIT : 0u));
}
__device__ __forceinline__ float dec_torsion_deg(float e) {
    const uint32_t b = __float_as_uint(e);
    float v = acos_deg(__uint_as_float(b & ~CK_NEG_BIT));
    if (b & CK_NEG_BIT) v = -1.0f * v;
    return v;
}
// array q of the hand-over (0..2 dihedrals, 3..5 bond angles; q = 6: B-factors, plain values)
template <int Q> __device__ __forceinline__ float dec_angle(float e) { return Q < 3 ? dec_torsion_deg(e) : (Q < 6 ? acos_deg(e) : e); }

struct lo_hi { float lo, hi; };
// mode 0: src holds values; 1: encoded dihedrals; 2: bond-angle cosines
__device__ __noinline__ lo_hi first_extrema(const float* __restrict__ src, uint32_t cnt, int lane, int mode) {
    const float kInf = __builtin_huge_valf();
    ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
    for (uint32_t k = lane; k < cnt; k += WAVE) {
        float v = src[k];
        if (mode == 1) v = dec_torsion_deg(v); else if (mode == 2) v = acos_deg(v);
        ext_min_upd(mn, v, k); ext_max_upd(mx, v, k);
    }
    return lo_hi{wave_ext_min(mn), wave_ext_max(mx)};
}

constexpr uint64_t CK_LAST = 1ull << 63;   // res_sc_addr flag: last residue of its chain
constexpr int CK_TILE = BLOCK;              // residues per tile
constexpr int CK_CAP = 2288;                // staged atom records per pass (a typical tile: 257 * 8.35 = 2146 +- 42); with the tables
                                            // below the block's LDS stays under a third of the CU's 160 KB at the allocation granularity
constexpr int CK_ZERO = CK_CAP;             // index of the all-zero record (missing atoms)

// res_sc_addr in memory: five bytes per residue instead of eight (round 4: the index kernel is a write stream, 8 of its 9.4 bytes per
// residue were this array) -- a dword plane with the address's low 32 bits, then a byte plane with bits 32..38 and the
// last-of-chain flag in bit 7 (a blob of up to 512 GB). The kernels keep handling the 64-bit value with CK_LAST in bit 63.
__device__ __forceinline__ void sc_addr_put(uint64_t* base, uint32_t n_res, size_t r, uint64_t addr, bool last) {
    reinterpret_cast<uint32_t*>(base)[r] = (uint32_t)addr;
    (reinterpret_cast<uint8_t*>(base) + 4 * (size_t)n_res)[r] = (uint8_t)(((addr >> 32) & 0x7fu) | (last ? 0x80u : 0u));
}
__device__ __forceinline__ unsigned long long sc_addr_get(const uint64_t* base, size_t n_res, size_t r) {
    const uint32_t lo = reinterpret_cast<const uint32_t*>(base)[r];
    const uint32_t hi = (reinterpret_cast<const uint8_t*>(base) + 4 * n_res)[r];
    return (unsigned long long)lo | ((unsigned long long)(hi & 0x7fu) << 32) | ((unsigned long long)(hi >> 7) << 63);
}

// ---- non-finite input ----------------------------------------------------------------------------------------------
// A chain with a NaN or an infinity in a coordinate of a named atom, or in a CA B-factor, is REFUSED (FCZ_E_NONFINITE). The
// reference's readers can produce such values (gemmi: mmCIF `?` / `.` -> NaN, lib/gemmi/numb.hpp:19-40; "nan" in a PDB column,
// lib/gemmi/pdb.hpp:49-54) and its compressor then writes a record whose quantiser parameters are NaNs with the input's sign and
// payload carried through SSE arithmetic (src/discretizer.cpp:22-33) -- a record that decodes to no structure. Here the angle
// kernels test every coordinate they stage (one v_cmp_class per value; the values are in registers anyway) and, only when one
// fires, find the atom's chain by two binary searches and set its bit; k_compress_pack refuses the flagged chains.
__device__ __forceinline__ bool nonfinite_f32(float v) { return __builtin_amdgcn_classf(v, 0x207); }   // sNaN | qNaN | -inf | +inf
__device__ __forceinline__ void flag_nonfinite_atom(const fcz_chain_batch& in, uint32_t atom, uint32_t* __restrict__ chain_bits) {
    uint32_t lo = 0, hi = in.n_residues;          // residue of the atom: the last r with atom_off[r] <= atom
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (in.atom_off[mid] <= atom) lo = mid; else hi = mid; }
    const uint32_t r = lo;
    lo = 0; hi = in.n_chains;                     // chain of the residue: the last c with res_off[c] <= r
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (in.res_off[mid] <= r) lo = mid; else hi = mid; }
    atomicOr(&chain_bits[lo >> 5], 1u << (lo & 31u));
}

// =====================================================================================================================
// k_compress_index
// =====================================================================================================================
// One 16-lane group (a DPP row) per chain, lane = residue k0 + sub: four chains' dependent loads (offsets -> codes) in flight per
// wavefront instead of one (round 4; the kernel is a write stream of 5 B/residue behind two round trips per chain).
__global__ __launch_bounds__(BLOCK) void k_compress_index(fcz_chain_batch in, const uint64_t* __restrict__ out_off,
                                                          uint64_t* __restrict__ res_sc_addr) {
    const int sub = threadIdx.x & (GROUP - 1);
    const uint32_t c = blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
    const bool live = c < in.n_chains;
    const uint32_t r0 = live ? in.res_off[c] : 0u, n = live ? in.res_off[c + 1] - r0 : 0u;
    const uint32_t thr = in.anchor_threshold > 0 ? (uint32_t)in.anchor_threshold : 1u;
    const rec_layout RL = make_layout(n, n / thr + 2, live ? in.title_off[c + 1] - in.title_off[c] : 0u, 0);   // o_sc does not depend on n_sc
    const uint64_t base = (live ? out_off[c] : 0ull) + RL.o_sc;
    uint32_t run = 0;
    // exclusive prefix inside the row: Hillis-Steele on the DPP network (zeros shift in), the row's total from its last lane
    auto row_scan = [&](uint32_t x, uint32_t* total) -> uint32_t {
        uint32_t v = x;
        v += dpp_u32_or0<0x111, 0xf>(v); v += dpp_u32_or0<0x112, 0xf>(v); v += dpp_u32_or0<0x114, 0xf>(v); v += dpp_u32_or0<0x118, 0xf>(v);
        *total = (uint32_t)__shfl((int)v, GROUP - 1, GROUP);
        return v - x;
    };
    constexpr int U = 8;                                   // residue codes in flight per lane and round trip (128 residues per round)
    for (uint32_t k0 = 0; k0 < n; k0 += U * GROUP) {       // (the trip count differs between the rows of a wavefront: rows that are done idle)
        uint32_t cnt[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint32_t k = k0 + (uint32_t)(u * GROUP + sub); cnt[u] = in.res_code[r0 + (k < n ? k : n - 1)]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = k0 + (uint32_t)(u * GROUP + sub);
            const uint32_t c3 = k < n ? (uint32_t)fcz_res_natoms[cnt[u] < 24 ? cnt[u] : 23] - 3u : 0u;
            uint32_t tot;
            const uint32_t ex = run + row_scan(c3, &tot);
            run += tot;
            if (k < n) sc_addr_put(res_sc_addr, in.n_residues, r0 + k, base + ex, k == n - 1);
        }
    }
}

// =====================================================================================================================
// k_compress_angles
// =====================================================================================================================
struct alignas(16) compress_lds {
    float4 atom[CK_CAP + 1];                    // {x, y, z, code bits}; [CK_ZERO] = zeros
    uint16_t idx[FCZ_MAX_RES_ATOMS][CK_TILE + 8];   // [canonical slot][residue in tile] -> atom record (row CK_TILE.. = successor)
    // Atom orders a residue is recognised in without a per-atom name lookup, per residue code, 16 bytes each:
    //   canon[j] = atom code of canonical slot j,  altc[j] = atom code at position j of the alternative order (what AlphaFold
    //   files and `decompress -a` use),  inv[sl] = position of canonical slot sl in the alternative order
    uint32_t ord_canon[FCZ_N_RES_CODES][4], ord_altc[FCZ_N_RES_CODES][4], ord_inv[FCZ_N_RES_CODES][4];
    unsigned long long sc_addr[CK_TILE];        // res_sc_addr of the tile's residues
    uint32_t olo[CK_TILE + 2];                  // atom_off of residues 0..CK_TILE+1 of the tile (clamped)
    uint16_t scpre[CK_TILE];                    // pass-local exclusive prefix of side-chain torsion counts
    uint8_t rc[CK_TILE + 8];
    uint8_t item_res[CK_TILE * 11];             // side-chain item -> residue in tile
    uint32_t wave_tot[WAVES_PER_BLOCK];
    uint32_t first_bad;
    uint8_t slot_of[FCZ_N_RES_CODES][40];       // atom code -> canonical slot, 255 = not in residue
    uint16_t prev[FCZ_N_RES_CODES][FCZ_MAX_RES_ATOMS];
    uint8_t natoms[32];
};

__device__ __forceinline__ v3 tile_atom(const compress_lds& L, uint32_t res, uint32_t slot) {
    const float4 a = L.atom[L.idx[slot][res]];
    return v3{a.x, a.y, a.z};
}

// last few atoms of the whole batch: element-wise loads that never run past the end of the arrays
struct atom_quad { float4 x, y, z; uint32_t c; };
__device__ __noinline__ atom_quad load_atoms_tail(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                  const uint8_t* __restrict__ code, uint32_t a, uint32_t left) {
    const bool h1 = left > 1, h2 = left > 2, h3 = left > 3;
    atom_quad q;
    q.x = float4{x[a], h1 ? x[a + 1] : 0.f, h2 ? x[a + 2] : 0.f, h3 ? x[a + 3] : 0.f};
    q.y = float4{y[a], h1 ? y[a + 1] : 0.f, h2 ? y[a + 2] : 0.f, h3 ? y[a + 3] : 0.f};
    q.z = float4{z[a], h1 ? z[a + 1] : 0.f, h2 ? z[a + 2] : 0.f, h3 ? z[a + 3] : 0.f};
    q.c = (uint32_t)code[a] | (h1 ? (uint32_t)code[a + 1] << 8 : 0u) | (h2 ? (uint32_t)code[a + 2] << 16 : 0u) |
          (h3 ? (uint32_t)code[a + 3] << 24 : 0u);
    return q;
}

#ifndef FCZ_COMPRESS_MIN_BLOCKS
#define FCZ_COMPRESS_MIN_BLOCKS 3
#endif

__global__ __launch_bounds__(BLOCK, FCZ_COMPRESS_MIN_BLOCKS)
void k_compress_angles(fcz_chain_batch in, uint32_t n_tiles_all, const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ tile_count,
                       const uint64_t* __restrict__ res_sc_addr, uint8_t* __restrict__ out, float* __restrict__ ang, uint32_t* __restrict__ nonfinite) {
    // tile_list != null: only the listed 256-residue tiles (the ones k_compress_angles_w left to this kernel); else all of them
    const uint32_t n_tiles = tile_list ? *tile_count : n_tiles_all;
    auto tile_of = [&](uint32_t k) -> uint32_t { return tile_list ? tile_list[k < n_tiles ? k : (n_tiles ? n_tiles - 1 : 0)] : k; };
    __shared__ compress_lds L;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < FCZ_N_RES_CODES * 40; i += BLOCK) (&L.slot_of[0][0])[i] = 255;
    if (t < 32) L.natoms[t] = fcz_res_natoms[t < 24 ? t : 23];
    if (t == 0) L.atom[CK_ZERO] = float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = t; i < FCZ_N_RES_CODES * FCZ_MAX_RES_ATOMS; i += BLOCK) {
        const int rc = i / FCZ_MAX_RES_ATOMS, j = i % FCZ_MAX_RES_ATOMS;
        if (j < fcz_res_natoms[rc]) L.slot_of[rc][fcz_res_atom[rc][j]] = (uint8_t)j;
        L.prev[rc][j] = fcz_res_prev[rc][j];
    }
    if (t < FCZ_N_RES_CODES) {
        const int rc = t, na = fcz_res_natoms[rc];
        uint32_t can[4] = {0, 0, 0, 0}, alt[4] = {0, 0, 0, 0}, inv[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; j++) {
            const bool in = j < na;
            const uint32_t aj = in ? fcz_res_alt_slot[rc][j] : 0u;
            can[j >> 2] |= (in ? (uint32_t)fcz_res_atom[rc][j] : 0xffu) << (8 * (j & 3));
            alt[j >> 2] |= (in ? (uint32_t)fcz_res_atom[rc][aj] : 0xffu) << (8 * (j & 3));
            if (in) inv[aj >> 2] |= (uint32_t)j << (8 * (aj & 3));
        }
        for (int d = 0; d < 4; d++) { L.ord_canon[rc][d] = can[d]; L.ord_altc[rc][d] = alt[d]; L.ord_inv[rc][d] = inv[d]; }
    }
    __syncthreads();

    const uint32_t R = in.n_residues;
    const size_t Rz = R;
    constexpr int NV = CK_CAP / (4 * BLOCK);   // full float4 rounds; the rest of the staging buffer is one atom per thread
    static_assert(CK_CAP - NV * 4 * BLOCK <= BLOCK, "tail round covers one atom per thread");

    // ---- software pipeline: while tile T computes, the per-residue metadata and the atoms of tile T+1 are in flight.
    //      All loads are unconditional (clamped indices): a conditional merge would make the compiler wait for the data
    //      at the load instead of a tile later. No global store is issued between these loads and the end of the tile's
    //      items (results stay in registers), so nothing forces the memory queue to drain early. ----
    struct meta { uint32_t olo, ox, rc, rc_succ, a0_next, e_next; unsigned long long sa; };
    auto load_meta = [&](uint32_t tile, uint32_t tile_next) -> meta {
        const size_t r_lo = (size_t)tile * CK_TILE;
        const size_t r = r_lo + (size_t)t;
        auto cl = [&](size_t x) -> size_t { return x < Rz ? x : Rz; };
        meta m;
        m.olo = in.atom_off[cl(r)];
        m.ox = in.atom_off[cl(r_lo + CK_TILE + (size_t)(t & 1))];            // olo[256], olo[257]
        m.rc = in.res_code[r < Rz ? r : Rz - 1];
        m.rc_succ = in.res_code[r_lo + CK_TILE < Rz ? r_lo + CK_TILE : Rz - 1];
        m.sa = sc_addr_get(res_sc_addr, Rz, r < Rz ? r : Rz - 1);
        const size_t r_n = (size_t)tile_next * CK_TILE;                      // this block's next tile
        m.a0_next = in.atom_off[cl(r_n)];
        m.e_next = in.atom_off[cl(r_n + CK_TILE + 1)];
        return m;
    };
    // rounds 0..NV-1: four atoms per thread (16-byte loads); the remaining CK_CAP - NV*4*BLOCK atoms: one atom per thread
    float4 px[NV], py[NV], pz[NV];
    uint32_t pc[NV];
    float tx = 0.f, ty = 0.f, tz = 0.f; uint32_t tc = 0;
    auto issue_atoms = [&](uint32_t A0, uint32_t cnt) {
        // clamp instead of predicating: every load is issued, out-of-range groups re-read the last in-range one
        const uint32_t last4 = cnt > 4 ? ((cnt - 1) & ~3u) : 0u;
#pragma unroll
        for (int u = 0; u < NV; u++) {
            uint32_t i4 = 4 * ((uint32_t)u * BLOCK + (uint32_t)t);
            i4 = i4 < cnt ? i4 : last4;
            const size_t g = (size_t)A0 + i4;                                // simple_tile(): reads stay inside the arrays
            px[u] = ld_f4(in.x + g); py[u] = ld_f4(in.y + g); pz[u] = ld_f4(in.z + g); pc[u] = ld_u32(in.atom_code + g);
        }
        uint32_t i1 = NV * 4 * BLOCK + (uint32_t)t;
        i1 = i1 < cnt ? i1 : cnt - 1;
        const size_t g = (size_t)A0 + i1;
        tx = in.x[g]; ty = in.y[g]; tz = in.z[g]; tc = in.atom_code[g];
    };
    // the prefetch covers residues 0..256 of a tile (the successor row always included); it is used when that fits
    // (and the 16-byte reads of its last group stay inside the arrays: the last tile of the batch takes the other path)
    auto simple_tile = [&](uint32_t a0, uint32_t e) -> bool { return e - a0 <= (uint32_t)CK_CAP && (size_t)e + 4 <= (size_t)in.n_atoms; };

    if (blockIdx.x >= n_tiles) return;
    meta mcur = load_meta(tile_of(blockIdx.x), tile_of(blockIdx.x + gridDim.x));
    uint32_t a0_cur, e_cur;
    {
        const size_t r_lo = (size_t)tile_of(blockIdx.x) * CK_TILE;
        a0_cur = in.atom_off[r_lo < Rz ? r_lo : Rz];
        e_cur = in.atom_off[r_lo + CK_TILE + 1 < Rz ? r_lo + CK_TILE + 1 : Rz];
    }
    bool pre_cur = simple_tile(a0_cur, e_cur);
    if (pre_cur) issue_atoms(a0_cur, e_cur - a0_cur);

    for (uint32_t tk = blockIdx.x; tk < n_tiles; tk += gridDim.x) {
        const uint32_t tile = tile_of(tk);
        const uint32_t r_lo = tile * CK_TILE;
        const uint32_t nres = (R - r_lo < (uint32_t)CK_TILE) ? R - r_lo : (uint32_t)CK_TILE;
        // ---- park the prefetched metadata (and atoms) in LDS ----
        {
            L.olo[t] = mcur.olo;
            if (t < 2) L.olo[CK_TILE + t] = mcur.ox;
            const bool in_r = r_lo + (uint32_t)t < R;
            L.rc[t] = (uint8_t)(in_r && mcur.rc < 24 ? mcur.rc : 23);
            L.sc_addr[t] = in_r ? mcur.sa : CK_LAST;
            if (t == 0) L.rc[CK_TILE] = (uint8_t)(mcur.rc_succ < 24 ? mcur.rc_succ : 23);
        }
        const uint32_t A0s = a0_cur, cnts = e_cur - a0_cur;
        const bool simple = pre_cur;
        if (simple) {
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i4 = 4 * ((uint32_t)u * BLOCK + (uint32_t)t);
                if (i4 < cnts) {
                    const float4 x = px[u], y = py[u], z = pz[u]; const uint32_t c = pc[u];
                    L.atom[i4 + 0] = float4{x.x, y.x, z.x, __uint_as_float(c & 0xffu)};
                    L.atom[i4 + 1] = float4{x.y, y.y, z.y, __uint_as_float((c >> 8) & 0xffu)};
                    L.atom[i4 + 2] = float4{x.z, y.z, z.z, __uint_as_float((c >> 16) & 0xffu)};
                    L.atom[i4 + 3] = float4{x.w, y.w, z.w, __uint_as_float(c >> 24)};
                }
            }
            const uint32_t i1 = NV * 4 * BLOCK + (uint32_t)t;
            if (i1 < cnts) L.atom[i1] = float4{tx, ty, tz, __uint_as_float(tc)};
        }
        // ---- prefetch of this block's next tile ----
        const uint32_t tk_n = tk + gridDim.x;
        const uint32_t a0_n = mcur.a0_next, e_n = mcur.e_next;
        const bool pre_n = tk_n < n_tiles && simple_tile(a0_n, e_n);
        const meta mnext = load_meta(tk_n < n_tiles ? tile_of(tk_n) : tile, tile_of(tk_n + gridDim.x));
        if (pre_n) issue_atoms(a0_n, e_n - a0_n);
        __syncthreads();

        // ---- passes: residues [s, e) whose atoms (plus the successor of e-1) are parked; one pass for a normal tile ----
        uint32_t s = 0;
        while (s < nres) {
            uint32_t e = nres, A0 = A0s, staged = simple ? cnts : 0u;
            bool tail_succ = true;
            if (!simple) {
                A0 = L.olo[s];
                // atoms needed when the pass ends before residue x (x > s); the successor only if it is in the same chain
                auto need_end = [&](uint32_t x) -> uint32_t {
                    const bool last = (L.sc_addr[x - 1] & CK_LAST) != 0;
                    return last ? L.olo[x] : L.olo[x + 1];
                };
                if (need_end(nres) - A0 > (uint32_t)CK_CAP) {
                    // unusually atom-rich stretch (explicit hydrogens, ...): shorten the pass. need_end is monotone.
                    if (t == 0) L.first_bad = nres + 1;
                    __syncthreads();
                    const uint32_t x = s + 1 + (uint32_t)t;   // candidate ends s+1 .. s+256
                    if (x <= nres && need_end(x) - A0 > (uint32_t)CK_CAP) atomicMin(&L.first_bad, x);
                    __syncthreads();
                    e = L.first_bad - 1;
                    __syncthreads();
                    if (e <= s) { s++; continue; }   // residue s (+ successor) alone exceeds the capacity: k_compress_pack rejects its chain
                }
                tail_succ = (L.sc_addr[e - 1] & CK_LAST) == 0;   // successor row needed (it exists then)
                const uint32_t cnt = (tail_succ ? L.olo[e + 1] : L.olo[e]) - A0;
                for (uint32_t i4 = 4 * (uint32_t)t; i4 < cnt; i4 += 4 * BLOCK) {
                    float4 qx, qy, qz; uint32_t qc;
                    if ((size_t)A0 + i4 + 4 <= (size_t)in.n_atoms) {
                        qx = ld_f4(in.x + A0 + i4); qy = ld_f4(in.y + A0 + i4); qz = ld_f4(in.z + A0 + i4);
                        qc = ld_u32(in.atom_code + A0 + i4);
                    } else {
                        const atom_quad q = load_atoms_tail(in.x, in.y, in.z, in.atom_code, A0 + i4, cnt - i4);
                        qx = q.x; qy = q.y; qz = q.z; qc = q.c;
                    }
                    L.atom[i4 + 0] = float4{qx.x, qy.x, qz.x, __uint_as_float(qc & 0xffu)};
                    L.atom[i4 + 1] = float4{qx.y, qy.y, qz.y, __uint_as_float((qc >> 8) & 0xffu)};
                    L.atom[i4 + 2] = float4{qx.z, qy.z, qz.z, __uint_as_float((qc >> 16) & 0xffu)};
                    L.atom[i4 + 3] = float4{qx.w, qy.w, qz.w, __uint_as_float(qc >> 24)};
                }
                __syncthreads();
                staged = cnt;
            } else {
                tail_succ = r_lo + nres < R;   // row nres exists in the batch
            }
            // non-finite coordinates of named atoms among the records of this pass (this kernel only sees the atom-rich tiles and
            // the tail of the batch: a plain walk over the parked records)
            for (uint32_t i = (uint32_t)t; i < staged; i += BLOCK) {
                const float4 a = L.atom[i];
                if (__builtin_expect((nonfinite_f32(a.x) | nonfinite_f32(a.y) | nonfinite_f32(a.z)) && __float_as_uint(a.w) != 255u, 0))
                    flag_nonfinite_atom(in, A0 + i, nonfinite);
            }

            // ---- slot index table: first atom of each canonical name; one thread per residue row ----
            const uint32_t last_row = tail_succ ? e : e - 1;
            for (uint32_t rr = s + (uint32_t)t; rr <= last_row; rr += BLOCK) {
                const uint32_t rc = L.rc[rr];
                const uint32_t lo = L.olo[rr] - A0, hi = L.olo[rr + 1] - A0;
                uint32_t codes[16];
#pragma unroll
                for (int j = 0; j < 16; j++) codes[j] = (lo + j < hi) ? __float_as_uint(L.atom[lo + j].w) : 255u;
                // Known order? The residue's first natoms codes equal the canonical list or the alternative-order list (four
                // dword compares each). Then slot sl sits at lo + sl, resp. lo + inv[sl]: fourteen stores, no per-atom lookup.
                // Names are distinct within a residue type, so "first atom of each name" is exactly that whatever follows.
                const uint32_t na = L.natoms[rc];
                uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 16; j++) pk[j >> 2] |= (codes[j] & 0xffu) << (8 * (j & 3));
                uint32_t dc = 0u, da = 0u;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int vb = (int)na - 4 * d;                                       // bytes of this dword that belong to the residue
                    const uint32_t m = vb >= 4 ? 0xffffffffu : (vb <= 0 ? 0u : (1u << (8 * vb)) - 1u);
                    dc |= (pk[d] ^ L.ord_canon[rc][d]) & m; da |= (pk[d] ^ L.ord_altc[rc][d]) & m;
                }
                const bool is_can = dc == 0u && hi - lo >= na, is_alt = da == 0u && hi - lo >= na;
                if (is_can || is_alt) {
#pragma unroll
                    for (int sl = 0; sl < FCZ_MAX_RES_ATOMS; sl++) {
                        const uint32_t pos = is_can ? (uint32_t)sl : ((L.ord_inv[rc][sl >> 2] >> (8 * (sl & 3))) & 0xffu);
                        L.idx[sl][rr] = (uint16_t)((uint32_t)sl < na ? lo + pos : (uint32_t)CK_ZERO);
                    }
                    continue;
                }
                uint32_t slots[16];
#pragma unroll
                for (int j = 0; j < 16; j++) slots[j] = (codes[j] < 40u) ? L.slot_of[rc][codes[j]] : 255u;
                // every slot starts at the zero record; then descending j, so that the first occurrence of a name is
                // the write that lands last (LDS operations of one wave execute in issue order)
#pragma unroll
                for (int sl = 0; sl < FCZ_MAX_RES_ATOMS; sl++) L.idx[sl][rr] = (uint16_t)CK_ZERO;
                uint32_t filled = 0;
#pragma unroll
                for (int j = 15; j >= 0; j--) {
                    const uint32_t sl = slots[j];
                    if (sl != 255u) { filled |= 1u << sl; L.idx[sl][rr] = (uint16_t)(lo + j); }
                }
                for (uint32_t i = lo + 16; i < hi; i++) {   // residues with more than 16 atoms (explicit hydrogens)
                    const uint32_t code = __float_as_uint(L.atom[i].w);
                    const uint32_t sl = code < 40u ? L.slot_of[rc][code] : 255u;
                    if (sl != 255u && !((filled >> sl) & 1u)) { filled |= 1u << sl; L.idx[sl][rr] = (uint16_t)i; }
                }
            }

            // ---- side-chain item numbering of the pass: block scan of the per-residue torsion counts ----
            const bool mine = (uint32_t)t >= s && (uint32_t)t < e;
            const bool my_win = mine && (L.sc_addr[t] & CK_LAST) == 0;
            const uint32_t my_cnt = mine ? (uint32_t)L.natoms[L.rc[t]] - 3u : 0u;
            uint32_t inc = my_cnt;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) { const uint32_t u = __shfl_up(inc, d, WAVE); if (lane >= d) inc += u; }
            if (lane == WAVE - 1) L.wave_tot[wave] = inc;
            __syncthreads();
            uint32_t my_pre = inc - my_cnt, n_sc = 0;
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; w++) { const uint32_t v = L.wave_tot[w]; if (w < wave) my_pre += v; n_sc += v; }
            if (mine) {
                L.scpre[t] = (uint16_t)my_pre;
                for (uint32_t j = 0; j < my_cnt; j++) L.item_res[my_pre + j] = (uint8_t)t;
            }
            __syncthreads();

            // ---- items. Every item is "angle between two vectors": for a dihedral the two plane normals
            //      (getTorsionFromXYZ, reference src/torsion_angle.cpp:50-94), for a bond angle the two bond vectors
            //      (angle(), src/float3d.h:55-65; getBondAngles src/nerf.cpp:495-508). Results stay in registers. ----
            // backbone: thread = residue window (k, k+1). psi = (N0,CA0,C0,N1), omega = (CA0,C0,N1,CA1),
            // phi = (C0,N1,CA1,C1) -> arrays 1, 2, 0 (split src/foldcomp.cpp:488-492); ca_c_n = (CA0,C0,N1),
            // c_n_ca = (C0,N1,CA1), n_ca_c = (N1,CA1,C1) -> arrays 4, 5, 3 (split :497-505)
            // The six items of a window share their ingredients: five bond vectors, four plane normals and their squared
            // lengths are formed once (the reference forms each of them up to three times: same operands, same operations,
            // same bits; a bond angle's vectors a - b are the exact negatives of the dihedrals' b - a, and negation commutes
            // with every rounding involved), then the double part runs once per item (q uniform: one copy of its code).
            float bb0 = 0.f, bb1 = 0.f, bb2 = 0.f, bb3 = 0.f, bb4 = 0.f, bb5 = 0.f;
#ifdef FCZ_ABL_NO_BB      // timing experiment only: the kernel without its backbone items
            if (false) {
#else
            if (my_win) {
#endif
                const v3 N0 = tile_atom(L, (uint32_t)t, 0), CA0 = tile_atom(L, (uint32_t)t, 1), C0 = tile_atom(L, (uint32_t)t, 2);
                const v3 N1 = tile_atom(L, (uint32_t)t + 1u, 0), CA1 = tile_atom(L, (uint32_t)t + 1u, 1), C1 = tile_atom(L, (uint32_t)t + 1u, 2);
                const v3 e0 = vsub(CA0, N0), e1 = vsub(C0, CA0), e2 = vsub(N1, C0), e3 = vsub(CA1, N1), e4 = vsub(C1, CA1);
                const v3 u0 = vcross(e0, e1), u1 = vcross(e1, e2), u2 = vcross(e2, e3), u3 = vcross(e3, e4);
                const float su0 = vdot_ref(u0, u0), su1 = vdot_ref(u1, u1), su2 = vdot_ref(u2, u2), su3 = vdot_ref(u3, u3);
                const float se1 = vdot_ref(e1, e1), se2 = vdot_ref(e2, e2), se3 = vdot_ref(e3, e3), se4 = vdot_ref(e4, e4);
                // item q: inner product, the two squared lengths, and (dihedrals) the sign test (u_a . (u_b x d2) < 0)
                const float ip0 = vdot_ref(u0, u1), ip1 = vdot_ref(u1, u2), ip2 = vdot_ref(u2, u3);
                const bool ng0 = vdot_ref(u0, vcross(u1, e1)) < 0.0f, ng1 = vdot_ref(u1, vcross(u2, e2)) < 0.0f, ng2 = vdot_ref(u2, vcross(u3, e3)) < 0.0f;
                // angle(a, b, c) = acos of cos(a - b, c - b): (CA0, C0, N1) -> (-e1, e2), (C0, N1, CA1) -> (-e2, e3), (N1, CA1, C1) -> (-e3, e4)
                const float ip3 = -vdot_ref(e1, e2), ip4 = -vdot_ref(e2, e3), ip5 = -vdot_ref(e3, e4);
#pragma unroll 1
                for (uint32_t q = 0; q < 6; q++) {
                    const float ip = q == 0 ? ip0 : q == 1 ? ip1 : q == 2 ? ip2 : q == 3 ? ip3 : q == 4 ? ip4 : ip5;
                    const float sa = q == 0 ? su0 : q == 1 ? su1 : q == 2 ? su2 : q == 3 ? se1 : q == 4 ? se2 : se3;
                    const float sb = q == 0 ? su1 : q == 1 ? su2 : q == 2 ? su3 : q == 3 ? se2 : q == 4 ? se3 : se4;
                    const float ct = vcos_theta_pre(ip, sa, sb);
                    // the cosine is handed over (dihedrals with their sign, see enc_torsion_cos); k_compress_pack finishes it
                    float v = ct;
                    if (q < 3) v = enc_torsion_cos(ct, q == 0 ? ng0 : q == 1 ? ng1 : ng2);
                    // psi, omega, phi -> arrays 1, 2, 0; ca_c_n, c_n_ca, n_ca_c -> arrays 4, 5, 3
                    bb1 = q == 0 ? v : bb1; bb2 = q == 1 ? v : bb2; bb0 = q == 2 ? v : bb0;
                    bb4 = q == 3 ? v : bb4; bb5 = q == 4 ? v : bb5; bb3 = q == 5 ? v : bb3;
                }
            }
            // side-chain torsions (calculateTorsionAnglesInResidue, reference src/sidechain.cpp:149-168): flat list
            uint32_t scb[3] = {0u, 0u, 0u};
#pragma unroll 1
            for (uint32_t i = 0; i < 11; i++) {
                const uint32_t ts = (uint32_t)t + i * BLOCK;
                if (i * BLOCK >= n_sc) break;
#ifdef FCZ_ABL_NO_SC      // timing experiment only: the kernel without its side-chain items
                break;
#endif
                uint32_t q = 0;
                if (ts < n_sc) {
                    const uint32_t res = L.item_res[ts];
                    const uint32_t j = 3 + ts - L.scpre[res];
                    const uint32_t pk = L.prev[L.rc[res]][j];
                    const v3 a = tile_atom(L, res, pk & 15u), b = tile_atom(L, res, (pk >> 4) & 15u), cc = tile_atom(L, res, (pk >> 8) & 15u);
                    const v3 d = tile_atom(L, res, j);
                    q = sidechain_torsion_byte(a, b, cc, d) & 0xffu;   // src/foldcomp.cpp:532-538
                }
                const uint32_t sh = q << (8 * (i & 3u));
                scb[0] |= (i < 4) ? sh : 0u; scb[1] |= (i >= 4 && i < 8) ? sh : 0u; scb[2] |= (i >= 8) ? sh : 0u;
            }
            // ---- results out: 6 coalesced float stores per window, consecutive bytes for consecutive side-chain items ----
            if (my_win) {
                float* ap = ang + r_lo + (uint32_t)t;
                ap[0] = bb0; ap[Rz] = bb1; ap[2 * Rz] = bb2; ap[3 * Rz] = bb3; ap[4 * Rz] = bb4; ap[5 * Rz] = bb5;
            }
#pragma unroll 1
            for (uint32_t i = 0; i < 11; i++) {
                const uint32_t ts = (uint32_t)t + i * BLOCK;
                if (i * BLOCK >= n_sc) break;
                if (ts < n_sc) {
                    const uint32_t res = L.item_res[ts];
                    const uint32_t jj = ts - L.scpre[res];
                    const uint32_t w = (i < 4) ? scb[0] : (i < 8 ? scb[1] : scb[2]);
                    out[(L.sc_addr[res] & ~CK_LAST) + jj] = (uint8_t)(w >> (8 * (i & 3u)));
                }
            }
            __syncthreads();
            s = e;
        }
        mcur = mnext; a0_cur = a0_n; e_cur = e_n; pre_cur = pre_n;
    }
}

// =====================================================================================================================
// k_compress_angles_w: the same stage with wavefront-private tiles
// =====================================================================================================================
// k_compress_angles synchronises its four wavefronts five times per tile; its phases add up (measured: 3.2 ms of staging /
// table / list work + 1.95 ms backbone items + 1.95 ms side-chain items = the kernel's 7.4 ms per 262 144 chains) because all
// wavefronts of a block are in the same phase. Here a wavefront owns its tile: 64 residue rows = 63 residues + the successor
// of the last one, lane = row. It loads, stages, indexes and evaluates only what it staged itself, so nothing but the order of
// its own LDS operations synchronises it, and the twelve wavefronts of a CU drift apart and fill each other's stalls. There is no
// register prefetch (the other wavefronts hide the load latency). A tile that does not fit (more than CW_CAP atoms: explicit
// hydrogens; or the last atoms of the batch, where 16-byte reads would run past the arrays) is not processed here: its
// 256-residue tile(s) go on a list for k_compress_angles, which owns every special case.
#ifndef FCZ_CW_UNROLL_BB
#define FCZ_CW_UNROLL_BB 2
#endif
#ifndef FCZ_CW_UNROLL_SC
#define FCZ_CW_UNROLL_SC 2
#endif
constexpr int CW_ROWS = WAVE;               // residue rows per wavefront tile
constexpr int CW_RES = WAVE - 1;            // residues a tile owns (the last row is the next tile's first residue)
constexpr int CW_CAP = 592;                 // staged atom records per tile (a typical tile: 64 * 8.35 = 535 +- 21)
constexpr int CW_ZERO = CW_CAP;
constexpr int CW_NA = (CW_CAP + WAVE - 1) / WAVE;             // rounds of one atom per lane
constexpr int CW_NC = (CW_CAP + 4 * WAVE - 1) / (4 * WAVE);   // rounds of four atom codes per lane
constexpr uint32_t CW_ABSENT = 255u;        // idx8 entry of a canonical atom the residue does not have

// LDS layout notes (SQ_LDS_BANK_CONFLICT of the first version of this kernel was 3.6x its LDS-active cycles):
//  * atoms are staged one record per lane and round (dword loads, consecutive lanes -> consecutive 16-byte records: a
//    ds_write_b128 is serviced in groups of 8 consecutive lanes over 32 banks, so this is conflict-free; four consecutive
//    records per lane, the float4-load layout, put lanes l and l+2 on the same banks: 4-way)
//  * the slot table is one 16-byte row per residue row, byte [slot] = position of the canonical atom inside the residue
//    (items of one residue read one or two dwords of one row: broadcast or distinct banks; rows of neighbouring residues are
//    4 dwords apart: distinct banks for the ~6 residues a 32-lane group covers. The old [slot][row] uint16 layout put every
//    slot of a row on ONE bank: 5- to 11-way)
//  * atom codes live in their own byte array (the float4 .w they used to ride in is on banks 3 mod 4 only)
typedef float cw_f4 __attribute__((ext_vector_type(4)));
struct alignas(16) compress_wave_lds {
    cw_f4 atom[CW_CAP + 1];                     // {x, y, z, -}; [CW_ZERO] = zeros
    uint32_t idx8[CW_ROWS][4];                  // 16 bytes per row: [slot] -> atom position relative to the row's first record
    uint32_t code4[CW_CAP / 4 + 8];             // atom codes, one byte per record (+ slack: a row reads 5 dwords from its start)
    uint32_t rowinfo[CW_ROWS];                  // first record (10 bits) | exclusive prefix of side-chain items << 10 | residue code << 20
    uint32_t sc_rel[CW_ROWS];                   // res_sc_addr of the row minus that of row 0
    uint8_t item_res[CW_RES * 11 + 11];         // side-chain item -> row
};
struct alignas(16) compress_tables_lds {        // per block, read-only after the prologue
    uint8_t slot_of[FCZ_N_RES_CODES][40];
    uint16_t prev[FCZ_N_RES_CODES][FCZ_MAX_RES_ATOMS];
    uint8_t natoms[32];
    uint32_t ord_canon[FCZ_N_RES_CODES][4], ord_altc[FCZ_N_RES_CODES][4], ord_inv[FCZ_N_RES_CODES][4];
};

// ordering point inside one wavefront: LDS operations of a wavefront execute in issue order, so all that is needed is that the
// compiler keeps the memory operations on their side of it
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// record of the atom at position `pos` of the residue whose first record is `lo`
__device__ __forceinline__ v3 wtile_rec(const compress_wave_lds& W, uint32_t lo, uint32_t pos) {
    // a vector-typed element: the 12 bytes in use come in one ds_read_b96 (as a struct of four floats the compiler read them as
    // ds_read_b64 + ds_read_b32: 4.8 -> 4.6 ms; forcing the full ds_read_b128 was no faster)
    const cw_f4 a = W.atom[pos == CW_ABSENT ? (uint32_t)CW_ZERO : lo + pos];
    return v3{a.x, a.y, a.z};
}

#ifdef FCZ_CW_TIMING
// measurement aid (not built into the product): wavefront-cycles spent between the phase boundaries of k_compress_angles_w
__device__ unsigned long long g_cw_timing[8];
#define CW_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define CW_STAMP(i)
#endif
__global__ __launch_bounds__(BLOCK, 3)
void k_compress_angles_w(fcz_chain_batch in, uint32_t n_wtiles, const uint64_t* __restrict__ res_sc_addr, uint8_t* __restrict__ out,
                         float* __restrict__ ang, uint32_t* __restrict__ tile_flags, uint32_t* __restrict__ tile_list, uint32_t* __restrict__ tile_count,
                         uint32_t* __restrict__ nonfinite) {
    __shared__ compress_wave_lds WL[WAVES_PER_BLOCK];
    __shared__ compress_tables_lds T;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < FCZ_N_RES_CODES * 40; i += BLOCK) (&T.slot_of[0][0])[i] = 255;
    if (t < 32) T.natoms[t] = fcz_res_natoms[t < 24 ? t : 23];
    __syncthreads();
    for (int i = t; i < FCZ_N_RES_CODES * FCZ_MAX_RES_ATOMS; i += BLOCK) {
        const int rc = i / FCZ_MAX_RES_ATOMS, j = i % FCZ_MAX_RES_ATOMS;
        if (j < fcz_res_natoms[rc]) T.slot_of[rc][fcz_res_atom[rc][j]] = (uint8_t)j;
        T.prev[rc][j] = fcz_res_prev[rc][j];
    }
    if (t < FCZ_N_RES_CODES) {
        const int rc = t, na = fcz_res_natoms[rc];
        uint32_t can[4] = {0, 0, 0, 0}, alt[4] = {0, 0, 0, 0}, inv[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; j++) {
            const bool inr = j < na;
            const uint32_t aj = inr ? fcz_res_alt_slot[rc][j] : 0u;
            can[j >> 2] |= (inr ? (uint32_t)fcz_res_atom[rc][j] : 0xffu) << (8 * (j & 3));
            alt[j >> 2] |= (inr ? (uint32_t)fcz_res_atom[rc][aj] : 0xffu) << (8 * (j & 3));
            if (inr) inv[aj >> 2] |= (uint32_t)j << (8 * (aj & 3));
        }
        for (int d = 0; d < 4; d++) { T.ord_canon[rc][d] = can[d]; T.ord_altc[rc][d] = alt[d]; T.ord_inv[rc][d] = inv[d]; }
    }
    __syncthreads();                             // the only block-level synchronisation of the kernel

    compress_wave_lds& W = WL[wave];
    const uint32_t R = in.n_residues;
    const size_t Rz = R;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    // row metadata of a tile (clamped indices: a tile past the end reads the last entries and is never used)
    struct wmeta { uint32_t olo, ohi, rc; unsigned long long sa; };
    auto load_meta = [&](uint32_t wt_) {
        const size_t r_ = (size_t)wt_ * CW_RES + (size_t)lane;
        wmeta m;
        m.olo = in.atom_off[r_ < Rz ? r_ : Rz]; m.ohi = in.atom_off[r_ + 1 < Rz ? r_ + 1 : Rz];
        m.rc = in.res_code[r_ < Rz ? r_ : Rz - 1];
        m.sa = sc_addr_get(res_sc_addr, Rz, r_ < Rz ? r_ : Rz - 1);
        return m;
    };
    const uint32_t wt0 = blockIdx.x * WAVES_PER_BLOCK + (uint32_t)wave;
    wmeta cur = load_meta(wt0 < n_wtiles ? wt0 : 0u), nxt = cur;
#ifdef FCZ_CW_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (uint32_t wt = wt0; wt < n_wtiles; wt += n_waves, cur = nxt) {
        const uint32_t r_lo = wt * CW_RES;
        const size_t r = (size_t)r_lo + (size_t)lane;
        const bool in_r = r < Rz;
        const uint32_t olo_g = cur.olo, ohi_g = cur.ohi, rc_g = cur.rc;
        const unsigned long long sa = in_r ? cur.sa : CK_LAST;
        // the next tile's metadata is requested a whole tile ahead: one memory round trip less on this wavefront's critical path
        { const uint32_t wtn = wt + n_waves; nxt = load_meta(wtn < n_wtiles ? wtn : wt); }
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)olo_g);
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ohi_g, WAVE - 1);
        const uint32_t cnt = e - a0;
        // positions inside a residue are bytes: a residue of 255 or more atom records is not for this kernel either
        if (cnt > (uint32_t)CW_CAP || (size_t)e + 4 > (size_t)in.n_atoms || __any(ohi_g - olo_g >= CW_ABSENT)) {
            // the 256-residue tiles that hold the tile's own residues go on the list of k_compress_angles
            if (lane == 0) {
                const uint32_t own_last = (r_lo + CW_RES - 1 < R ? r_lo + CW_RES - 1 : R - 1);
                for (uint32_t tk = r_lo / CK_TILE; tk <= own_last / CK_TILE; tk++)
                    if (atomicExch(&tile_flags[tk], 1u) == 0u) tile_list[atomicAdd(tile_count, 1u)] = tk;
            }
            continue;
        }
        wave_sync();          // the previous tile's last LDS reads are issued before anything is overwritten
        CW_STAMP(0)
        // ---- stage: atoms of the tile (record per lane and round), codes (dword of four per lane and round) ----
        const uint32_t rc = (in_r && rc_g < 24u) ? rc_g : 23u;
        const uint32_t lo = olo_g - a0, hi = ohi_g - a0;
        const unsigned long long sa0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sa >> 32)) << 32) |
                                       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sa);
        const unsigned long long base0 = sa0 & ~CK_LAST;
        W.sc_rel[lane] = (uint32_t)((sa & ~CK_LAST) - base0);
        const bool is_last = (sa & CK_LAST) != 0;
        if (lane == 0) W.atom[CW_ZERO] = cw_f4{0.f, 0.f, 0.f, 0.f};
        bool odd = false;                        // a NaN or an infinity among this lane's coordinates (lanes past the end read zero)
        {
            float px[CW_NA], py[CW_NA], pz[CW_NA];
            uint32_t pc[CW_NC];
            // buffer loads: base (SGPRs) = the array at the tile's first atom, range = the tile's atoms, offset = 4 * lane + a
            // constant per round: no per-load address arithmetic on the VALU, lanes past the end read zero
            {
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in.x) + a0, 0, (int)(cnt * 4u), 0x00020000);
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in.y) + a0, 0, (int)(cnt * 4u), 0x00020000);
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in.z) + a0, 0, (int)(cnt * 4u), 0x00020000);
                const int vo = 4 * lane;
#pragma unroll
                for (int u = 0; u < CW_NA; u++) {
                    px[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vo + u * 4 * WAVE, 0, 0));
                    py[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo + u * 4 * WAVE, 0, 0));
                    pz[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, vo + u * 4 * WAVE, 0, 0));
                }
            }
#pragma unroll
            for (int u = 0; u < CW_NC; u++) {
                const uint32_t i4 = 4 * ((uint32_t)u * WAVE + (uint32_t)lane);
                pc[u] = ld_u32(in.atom_code + (size_t)a0 + (i4 < cnt ? i4 : 0u));
            }
#pragma unroll
            for (int u = 0; u < CW_NA; u++) {
                const uint32_t i = (uint32_t)u * WAVE + (uint32_t)lane;
                if (i < cnt) W.atom[i] = cw_f4{px[u], py[u], pz[u], 0.f};
                odd |= nonfinite_f32(px[u]) | nonfinite_f32(py[u]) | nonfinite_f32(pz[u]);
            }
#pragma unroll
            for (int u = 0; u < CW_NC; u++) {
                const uint32_t i4 = 4 * ((uint32_t)u * WAVE + (uint32_t)lane);
                if (i4 < cnt) W.code4[i4 >> 2] = pc[u];
            }
        }
        wave_sync();          // LDS operations of a wavefront execute in issue order: rows read what was staged
        if (__builtin_expect(__any(odd), 0)) {
            // never on a real file: which atoms, are they named ones (code != 255), whose chain -- read back from the staged records
            const uint8_t* code8 = reinterpret_cast<const uint8_t*>(&W.code4[0]);
            for (uint32_t i = (uint32_t)lane; i < cnt; i += WAVE) {
                const cw_f4 a = W.atom[i];
                if ((nonfinite_f32(a.x) | nonfinite_f32(a.y) | nonfinite_f32(a.z)) && code8[i] != 255u) flag_nonfinite_atom(in, a0 + i, nonfinite);
            }
        }
        CW_STAMP(1)
        // ---- slot table row of this lane's residue ----
        const uint32_t na = T.natoms[rc];
        uint32_t row0;        // positions of N, CA, C (bytes 0..2)
        {
            // the 16 codes from the row's first record on, as four packed dwords (bytes past the residue are masked below)
            const uint32_t d0 = lo >> 2, sh = lo & 3u;
            const uint32_t w0 = W.code4[d0], w1 = W.code4[d0 + 1], w2 = W.code4[d0 + 2], w3 = W.code4[d0 + 3], w4 = W.code4[d0 + 4];
            uint32_t pk[4];
            pk[0] = __builtin_amdgcn_alignbyte(w1, w0, sh); pk[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
            pk[2] = __builtin_amdgcn_alignbyte(w3, w2, sh); pk[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
            uint32_t dc = 0u, da = 0u, mk[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int vb = (int)na - 4 * d;
                mk[d] = vb >= 4 ? 0xffffffffu : (vb <= 0 ? 0u : (1u << (8 * vb)) - 1u);
                dc |= (pk[d] ^ T.ord_canon[rc][d]) & mk[d]; da |= (pk[d] ^ T.ord_altc[rc][d]) & mk[d];
            }
            const bool is_can = dc == 0u && hi - lo >= na, is_alt = da == 0u && hi - lo >= na;
            if (is_can || is_alt) {
                uint32_t rw[4];
#pragma unroll
                for (int d = 0; d < 4; d++) rw[d] = ((is_can ? 0x03020100u + 0x04040404u * (uint32_t)d : T.ord_inv[rc][d]) & mk[d]) | ~mk[d];
                *reinterpret_cast<uint4*>(&W.idx8[lane][0]) = uint4{rw[0], rw[1], rw[2], rw[3]};
                row0 = rw[0];
            } else {
                *reinterpret_cast<uint4*>(&W.idx8[lane][0]) = uint4{~0u, ~0u, ~0u, ~0u};
                uint8_t* row8 = reinterpret_cast<uint8_t*>(&W.idx8[lane][0]);
                uint32_t filled = 0;
                // first occurrence of a name wins: positions written from the back
#pragma unroll
                for (int j = 15; j >= 0; j--) {
                    const uint32_t code = (pk[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t sl = (lo + (uint32_t)j < hi && code < 40u) ? T.slot_of[rc][code] : 255u;
                    if (sl != 255u) { filled |= 1u << sl; row8[sl] = (uint8_t)j; }
                }
                const uint8_t* code8 = reinterpret_cast<const uint8_t*>(&W.code4[0]);
                for (uint32_t i = lo + 16; i < hi; i++) {
                    const uint32_t code = code8[i];
                    const uint32_t sl = code < 40u ? T.slot_of[rc][code] : 255u;
                    if (sl != 255u && !((filled >> sl) & 1u)) { filled |= 1u << sl; row8[sl] = (uint8_t)(i - lo); }
                }
                row0 = W.idx8[lane][0];
            }
        }
        // ---- side-chain item numbering: rows 0..62 own items ----
        const bool mine = in_r && lane < CW_RES;
        const bool my_win = mine && !is_last;
        const uint32_t my_cnt = mine ? na - 3u : 0u;
        uint32_t n_sc;
        const uint32_t my_pre = wave_excl_scan_dpp(my_cnt, &n_sc);
        W.rowinfo[lane] = lo | (my_pre << 10) | (rc << 20);
        for (uint32_t j = 0; j < my_cnt; j++) W.item_res[my_pre + j] = (uint8_t)lane;
        const uint32_t row0_next = (uint32_t)__shfl_down((int)row0, 1, WAVE);
        wave_sync();
        CW_STAMP(2)
        // ---- backbone items of the window (this row, next row): shared ingredients as in k_compress_angles ----
        float bb0 = 0.f, bb1 = 0.f, bb2 = 0.f, bb3 = 0.f, bb4 = 0.f, bb5 = 0.f;
        if (my_win) {
            const v3 N0 = wtile_rec(W, lo, row0 & 0xffu), CA0 = wtile_rec(W, lo, (row0 >> 8) & 0xffu), C0 = wtile_rec(W, lo, (row0 >> 16) & 0xffu);
            const v3 N1 = wtile_rec(W, hi, row0_next & 0xffu), CA1 = wtile_rec(W, hi, (row0_next >> 8) & 0xffu), C1 = wtile_rec(W, hi, (row0_next >> 16) & 0xffu);
            const v3 e0 = vsub(CA0, N0), e1 = vsub(C0, CA0), e2 = vsub(N1, C0), e3 = vsub(CA1, N1), e4 = vsub(C1, CA1);
            const v3 u0 = vcross(e0, e1), u1 = vcross(e1, e2), u2 = vcross(e2, e3), u3 = vcross(e3, e4);
            const float su0 = vdot_ref(u0, u0), su1 = vdot_ref(u1, u1), su2 = vdot_ref(u2, u2), su3 = vdot_ref(u3, u3);
            const float se1 = vdot_ref(e1, e1), se2 = vdot_ref(e2, e2), se3 = vdot_ref(e3, e3), se4 = vdot_ref(e4, e4);
            const float ip0 = vdot_ref(u0, u1), ip1 = vdot_ref(u1, u2), ip2 = vdot_ref(u2, u3);
            const bool ng0 = vdot_ref(u0, vcross(u1, e1)) < 0.0f, ng1 = vdot_ref(u1, vcross(u2, e2)) < 0.0f, ng2 = vdot_ref(u2, vcross(u3, e3)) < 0.0f;
            const float ip3 = -vdot_ref(e1, e2), ip4 = -vdot_ref(e2, e3), ip5 = -vdot_ref(e3, e4);
            // the cosines are handed over (dihedrals with their sign, see enc_torsion_cos); k_compress_pack finishes them.
            // psi, omega, phi -> arrays 1, 2, 0; ca_c_n, c_n_ca, n_ca_c -> arrays 4, 5, 3
            bb1 = enc_torsion_cos(vcos_theta_pre(ip0, su0, su1), ng0);
            bb2 = enc_torsion_cos(vcos_theta_pre(ip1, su1, su2), ng1);
            bb0 = enc_torsion_cos(vcos_theta_pre(ip2, su2, su3), ng2);
            bb4 = vcos_theta_pre(ip3, se1, se2);
            bb5 = vcos_theta_pre(ip4, se2, se3);
            bb3 = vcos_theta_pre(ip5, se3, se4);
        }
        // angles out right away: the six registers are free for the side-chain items
        if (my_win) {
            float* ap = ang + r;
            ap[0] = bb0; ap[Rz] = bb1; ap[2 * Rz] = bb2; ap[3 * Rz] = bb3; ap[4 * Rz] = bb4; ap[5 * Rz] = bb5;
        }
        CW_STAMP(3)
        // ---- side-chain torsion bytes: flat item list, lane = item; each byte leaves as soon as it exists (consecutive items
        //      are consecutive bytes of the record's side-chain section) ----
#pragma unroll FCZ_CW_UNROLL_SC
        for (uint32_t i = 0; i < 11; i++) {
            const uint32_t ts = (uint32_t)lane + i * WAVE;
            if (i * WAVE >= n_sc) break;
            if (ts < n_sc) {
                const uint32_t res = W.item_res[ts];
                const uint32_t info = W.rowinfo[res];
                const uint32_t rlo = info & 0x3ffu;
                const uint32_t jj = ts - ((info >> 10) & 0x3ffu), j = 3 + jj;
                const uint32_t pk2 = T.prev[info >> 20][j];
                const uint8_t* row8 = reinterpret_cast<const uint8_t*>(&W.idx8[res][0]);
                const v3 a = wtile_rec(W, rlo, row8[pk2 & 15u]), b = wtile_rec(W, rlo, row8[(pk2 >> 4) & 15u]), cc = wtile_rec(W, rlo, row8[(pk2 >> 8) & 15u]);
                const v3 d = wtile_rec(W, rlo, row8[j]);
                out[base0 + W.sc_rel[res] + jj] = (uint8_t)sidechain_torsion_byte(a, b, cc, d);
            }
        }
        CW_STAMP(4)
        CW_STAMP(5)
    }
#ifdef FCZ_CW_TIMING
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_cw_timing[i], tacc[i]);
#endif
}

// =====================================================================================================================
// k_compress_pack
// =====================================================================================================================
// One chain by one wavefront. U = rounds of 64 residues whose values stay in registers (chains of up to U x 64 residues take the
// register path; U = 6 covers the 350-residue headline, 2 and 4 the shorter classes; chains of up to 64 residues are
// compress_pack_rows' below).
template <int U>
__device__ __forceinline__ void compress_pack_chain(const fcz_chain_batch& in, const uint32_t c, const uint32_t r0, const uint32_t n,
                                                    const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                    int32_t* __restrict__ status, float* __restrict__ ang, int keep_first_angle,
                                                    const uint32_t* __restrict__ nonfinite) {
    const int lane = threadIdx.x & 63;
    const uint32_t title_len = in.title_off[c + 1] - in.title_off[c];
    const uint32_t thr = (uint32_t)in.anchor_threshold;
    uint8_t* rec = out + out_off[c];
    const uint32_t rec_size = (uint32_t)(out_off[c + 1] - out_off[c]);
    const int32_t h_first_res = in.first_res_index[c], h_first_atom = in.first_atom_index[c];
    const char h_chain = in.chain_id[c];
    const uint32_t a_first = in.atom_off[r0], a_end = in.atom_off[r0 + n];
    const uint32_t h_rc_first = n ? in.res_code[r0] : 23u, h_rc_last = n ? in.res_code[r0 + n - 1] : 23u;

    // ---- every load of a normal chain (<= 384 residues) is issued here, unconditionally and from clamped indices, so
    //      that validation, anchors and quantisation do not each pay a memory round trip ----
    const bool small = n >= 2 && n <= (uint32_t)(U * WAVE);
    const size_t R = in.n_residues;
    const uint32_t m = n ? n - 1 : 0;
    float* a_arr = ang + (r0 < R ? r0 : (R ? R - 1 : 0));   // array q of this chain = a_arr + q*R : phi psi omega n_ca_c ca_c_n c_n_ca
    float va[7][U];
    uint32_t rcs[U], o0[U], o2[U];
#pragma unroll
    for (int u = 0; u < U; u++) { rcs[u] = 0; o0[u] = o2[u] = 0; for (int q = 0; q < 7; q++) va[q][u] = 0.f; }
    if (small) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            const uint32_t kw = k < m ? k : m - 1, kr = k < n ? k : n - 1;
#pragma unroll
            for (int q = 0; q < 6; q++) va[q][u] = a_arr[(size_t)q * R + kw];
            va[6][u] = in.bfac_ca[r0 + kr];
            rcs[u] = in.res_code[r0 + kr];
            o0[u] = in.atom_off[r0 + kr];
            o2[u] = in.atom_off[r0 + (k + 2 < n ? k + 2 : n)];
        }
    }

    // ---- the double part of the six backbone angles (hand-over above): acos of what was just loaded ----
    if (small) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if ((uint32_t)(u * WAVE) >= m) continue;      // a round no residue of this chain falls into (wave-uniform): short chains
            va[0][u] = dec_angle<0>(va[0][u]); va[1][u] = dec_angle<1>(va[1][u]); va[2][u] = dec_angle<2>(va[2][u]);
            va[3][u] = dec_angle<3>(va[3][u]); va[4][u] = dec_angle<4>(va[4][u]); va[5][u] = dec_angle<5>(va[5][u]);
        }
    }

    // ---- validation (the reference aborts on these inputs) + total side-chain torsion count ----
    int bad = (n < 2) ? FCZ_E_TOO_SHORT : (thr < 1 ? FCZ_E_INVALID_ARG : 0);
    // nResidue is a uint16 and nAnchor a uint8 in the header (src/foldcomp.h:120-125): a chain beyond them would get a record
    // whose layout uses the full values and whose header holds wrapped ones (the reference writes exactly that, unreadable)
    if (!bad && (n > 65535u || n / thr + 2u > 255u)) bad = FCZ_E_INVALID_ARG;
    uint32_t nsc = 0;
    // a non-finite coordinate of a named atom (found by the angle kernels) or CA B-factor: refused, see flag_nonfinite_atom
    if (!bad && ((nonfinite[c >> 5] >> (c & 31u)) & 1u)) bad = FCZ_E_NONFINITE;
    auto check = [&](uint32_t rc, uint32_t span, float bf) {
        if (!res_code_ok(rc)) bad = bad ? bad : FCZ_E_RESIDUE;
        // a residue and its successor must fit the staging buffer of k_compress_angles (not a protein otherwise)
        if (span > (uint32_t)CK_CAP) bad = bad ? bad : FCZ_E_INVALID_ARG;
        if (nonfinite_f32(bf)) bad = bad ? bad : FCZ_E_NONFINITE;
        nsc += fcz_res_natoms[rc < 24 ? rc : 23] - 3;
    };
    if (small) {
#pragma unroll
        for (int u = 0; u < U; u++) if ((uint32_t)(u * WAVE + lane) < n) check(rcs[u], o2[u] - o0[u], va[6][u]);
    } else if (U > 1) {
        for (uint32_t k = lane; k < n; k += WAVE)
            check(in.res_code[r0 + k], in.atom_off[r0 + (k + 2 < n ? k + 2 : n)] - in.atom_off[r0 + k], in.bfac_ca[r0 + k]);
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { int o = __shfl_xor(bad, d, WAVE); bad = o < bad ? o : bad; }
    if (bad) {
        for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
        if (lane == 0 && status) status[c] = bad;
        return;
    }
    nsc = wave_sum(nsc);

    const uint32_t n_anchor = n / thr + 2;
    const uint32_t interval = n / (n_anchor - 1);
    const rec_layout RL = make_layout(n, n_anchor, title_len, nsc);

    // ---- anchors (Foldcomp::_setAnchor src/foldcomp.cpp:745-761, written :1045-1059): one lane per anchor; N, CA, C are
    //      the first atoms of those names in the residue ----
    for (uint32_t sl = lane; sl < n_anchor; sl += WAVE) {
        const uint32_t k = (sl + 1 < n_anchor) ? sl * interval : n - 1;
        const uint32_t lo = in.atom_off[r0 + k], hi = in.atom_off[r0 + k + 1];
        // the first 8 atom codes in one batch of loads (N, CA, C lead every residue of a normal file); a serial scan
        // with one dependent load per atom only when a name is still missing after those
        uint32_t at[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t codes[8];
#pragma unroll
        for (int j = 0; j < 8; j++) codes[j] = (lo + j < hi) ? (uint32_t)in.atom_code[lo + j] : 255u;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            at[0] = codes[j] == 0u ? lo + j : at[0];
            at[1] = codes[j] == 1u ? lo + j : at[1];
            at[2] = codes[j] == 2u ? lo + j : at[2];
        }
        if (__builtin_expect((at[0] & at[1] & at[2]) == 0xffffffffu || at[0] == 0xffffffffu || at[1] == 0xffffffffu || at[2] == 0xffffffffu, 0)) {
            for (uint32_t i = lo + 8; i < hi; i++) {
                const uint32_t code = in.atom_code[i];
                if (code == 0u && at[0] == 0xffffffffu) at[0] = i;
                if (code == 1u && at[1] == 0xffffffffu) at[1] = i;
                if (code == 2u && at[2] == 0xffffffffu) at[2] = i;
            }
        }
        v3 p[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const bool have = at[q] != 0xffffffffu;
            const uint32_t i = have ? at[q] : lo;   // any valid index; masked below
            const bool ok = have && lo < hi;
            p[q] = ok ? v3{in.x[i], in.y[i], in.z[i]} : v3{0.f, 0.f, 0.f};
        }
        uint8_t* q = rec + RL.o_anchor + 36 * sl;
        st_f32(q, p[0].x); st_f32(q + 4, p[0].y); st_f32(q + 8, p[0].z);
        st_f32(q + 12, p[1].x); st_f32(q + 16, p[1].y); st_f32(q + 20, p[1].z);
        st_f32(q + 24, p[2].x); st_f32(q + 28, p[2].y); st_f32(q + 32, p[2].z);
        st_u32(rec + RL.o_aidx + 4 * sl, k);
        if (keep_first_angle && sl == 0) a_arr[3 * R + (n - 1)] = bond_angle_deg(p[0], p[1], p[2]);
    }

    // ---- per-chain quantiser parameters (Discretizer::Discretizer src/discretizer.cpp:22-33), then the
    //      packed words (src/foldcomp.cpp:582-601, convertBackboneChainToBytes :33-52) + B-factor bytes ----
    const float kInf = __builtin_huge_valf();
    float qmin[7], qdisc[7], qcont[7];
    const float nbins[7] = {4095.0f, 4095.0f, 2047.0f, 255.0f, 255.0f, 255.0f, 255.0f};
    auto pack_store = [&](uint32_t k, uint32_t res, float v0, float v1, float v2, float v3_, float v4, float v5, float v6) {
        uint32_t om = 0, ps = 0, ph = 0, b1 = 0, b2 = 0, b3 = 0;
        if (k < m) {
            ph = quant_round(v0, qmin[0], qdisc[0]) & 0xfffu;
            ps = quant_round(v1, qmin[1], qdisc[1]) & 0xfffu;
            om = quant_round(v2, qmin[2], qdisc[2]) & 0x7ffu;
            b3 = quant_round(v3_, qmin[3], qdisc[3]) & 0xffu;
            b1 = quant_round(v4, qmin[4], qdisc[4]) & 0xffu;
            b2 = quant_round(v5, qmin[5], qdisc[5]) & 0xffu;
        }
        const uint32_t w0 = ((res & 0x1fu) << 3) | (om >> 8), w1 = om & 0xffu, w2 = ps >> 4,
                       w3 = ((ps & 0xfu) << 4) | (ph >> 8), w4 = ph & 0xffu;
        const uint64_t word = (uint64_t)w0 | ((uint64_t)w1 << 8) | ((uint64_t)w2 << 16) | ((uint64_t)w3 << 24) |
                              ((uint64_t)w4 << 32) | ((uint64_t)b1 << 40) | ((uint64_t)b2 << 48) | ((uint64_t)b3 << 56);
        st_u64(rec + RL.o_words + 8 * (size_t)k, word);
        rec[RL.o_tbytes + k] = (uint8_t)quant_round(v6, qmin[6], qdisc[6]);
    };
    // first = element 0 of the array (wave-uniform). std::min_element / max_element start from it and replace it only when a
    // comparison says so: a NaN there stays (nothing compares below or above it), a NaN anywhere else is never picked -- which is
    // what the fminf / fmaxf reductions give
    auto finish_q = [&](int q, float first, float lo, float hi, const float* src, uint32_t cntq, int mode) {
        if (__builtin_expect(lo == 0.0f || hi == 0.0f, 0)) { const lo_hi e = first_extrema(src, cntq, lane, mode); lo = e.lo; hi = e.hi; }
        qmin[q] = lo; qdisc[q] = nbins[q] / (hi - lo); qcont[q] = (hi - lo) / nbins[q];
        // a NaN at the head is minimum and maximum at once, and x86-64 hands its bits through the subtraction and the division
        // (NaN - NaN = the first operand, quiet already): the record's header holds those bits twice (acos_deg_exact makes them)
        if (__builtin_expect(first != first, 0)) { qmin[q] = first; qdisc[q] = first; qcont[q] = first; }
    };
    if (small) {
        // everything of the chain is in registers already: no reload for the quantisation pass
#pragma unroll
        for (int q = 0; q < 7; q++) {
            float lo = kInf, hi = -kInf;
            const uint32_t cntq = (q < 6) ? m : n;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool on = (uint32_t)(u * WAVE + lane) < cntq;
                lo = __builtin_fminf(lo, on ? va[q][u] : kInf);
                hi = __builtin_fmaxf(hi, on ? va[q][u] : -kInf);
            }
            const float first = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(va[q][0])));
            finish_q(q, first, wave_min_f32(lo), wave_max_f32(hi), (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0), cntq, q < 3 ? 1 : (q < 6 ? 2 : 0));
        }
        if (keep_first_angle) {
            // fcz_compress_angles reads the scratch back: leave the finished angles there
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = u * WAVE + lane;
#pragma unroll
                for (int q = 0; q < 6; q++) if (k < m) a_arr[(size_t)q * R + k] = va[q][u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            if (k < n) pack_store(k, rcs[u], va[0][u], va[1][u], va[2][u], va[3][u], va[4][u], va[5][u], va[6][u]);
            __builtin_amdgcn_sched_barrier(0);   // one word at a time: interleaving all of them only inflates the live set
        }
    } else if (U > 1) {
        // longer chains: blocks of U x 64 residues in the register path's shape -- every load of a block in flight at once, the
        // rounds past the chain's end skipped (wave-uniform) -- first for the extrema, the finished angles written back in place
        // (each lane its own entries), then again (from the L2) for the words. (Round 4 walked 128 residues per memory round trip;
        // this form is 2 % faster on 1 000-residue chains: the path is as VALU-bound as the rest, profiles/r5_short_chains.txt.)
        float lo[7], hi[7], first[7];
#pragma unroll
        for (int q = 0; q < 7; q++) { lo[q] = kInf; hi[q] = -kInf; first[q] = 0.f; }
        for (uint32_t b0 = 0; b0 < n; b0 += (uint32_t)(U * WAVE)) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (b0 + (uint32_t)(u * WAVE) >= n) continue;
                const uint32_t k = b0 + u * WAVE + lane, kw = k < m ? k : m - 1, kr = k < n ? k : n - 1;
#pragma unroll
                for (int q = 0; q < 6; q++) va[q][u] = a_arr[(size_t)q * R + kw];
                va[6][u] = in.bfac_ca[r0 + kr];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (b0 + (uint32_t)(u * WAVE) >= n) continue;
                const uint32_t k = b0 + u * WAVE + lane;
                va[0][u] = dec_angle<0>(va[0][u]); va[1][u] = dec_angle<1>(va[1][u]); va[2][u] = dec_angle<2>(va[2][u]);
                va[3][u] = dec_angle<3>(va[3][u]); va[4][u] = dec_angle<4>(va[4][u]); va[5][u] = dec_angle<5>(va[5][u]);
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    if (u == 0 && b0 == 0) first[q] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(va[q][0])));
                    const bool on = k < ((q < 6) ? m : n);
                    if (q < 6 && on) a_arr[(size_t)q * R + k] = va[q][u];
                    lo[q] = __builtin_fminf(lo[q], on ? va[q][u] : kInf); hi[q] = __builtin_fmaxf(hi[q], on ? va[q][u] : -kInf);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 7; q++)
            finish_q(q, first[q], wave_min_f32(lo[q]), wave_max_f32(hi[q]), (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0), (q < 6) ? m : n, 0);
        for (uint32_t b0 = 0; b0 < n; b0 += (uint32_t)(U * WAVE)) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (b0 + (uint32_t)(u * WAVE) >= n) continue;
                const uint32_t k = b0 + u * WAVE + lane, kw = k < m ? k : m - 1, kr = k < n ? k : n - 1;
#pragma unroll
                for (int q = 0; q < 6; q++) va[q][u] = a_arr[(size_t)q * R + kw];
                va[6][u] = in.bfac_ca[r0 + kr];
                rcs[u] = in.res_code[r0 + kr];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (b0 + (uint32_t)(u * WAVE) >= n) continue;
                const uint32_t k = b0 + u * WAVE + lane;
                if (k < n) pack_store(k, rcs[u], va[0][u], va[1][u], va[2][u], va[3][u], va[4][u], va[5][u], va[6][u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    for (uint32_t i = lane; i < title_len; i += WAVE) rec[RL.o_title + i] = (uint8_t)in.titles[in.title_off[c] + i];

    // ---- header (CompressedFileHeader src/foldcomp.h:118-136; get_header src/foldcomp.cpp:1340) ----
    if (lane == 0) {
        rec[0] = 'F'; rec[1] = 'C'; rec[2] = 'M'; rec[3] = 'P';
        uint8_t* h = rec + 4;
        st_u16(h + 0, n);
        st_u16(h + 2, a_end - a_first);
        st_u16(h + 4, (uint32_t)h_first_res);
        st_u16(h + 6, (uint32_t)h_first_atom);
        h[8] = (uint8_t)n_anchor;
        h[9] = (uint8_t)h_chain;
        h[10] = 0; h[11] = 0;  // struct padding: the reference leaves it uninitialised
        st_u32(h + 12, nsc);
        h[16] = (uint8_t)fcz_res1[h_rc_first];
        h[17] = (uint8_t)fcz_res1[h_rc_last];
        h[18] = 0; h[19] = 0;
        st_u32(h + 20, title_len);
        // a NaN among the angle parameters is the NaN of the chain's first angle with the bits x86-64 / glibc give it (acos_deg_exact:
        // negative from a NaN cosine, positive from a cosine beyond +-1), carried unchanged through the quantiser (finish_q)
        auto x86_nan = [](float v) { return v; };
#pragma unroll
        for (int q = 0; q < 6; q++) { st_f32(h + 24 + 4 * q, x86_nan(qmin[q])); st_f32(h + 48 + 4 * q, x86_nan(qcont[q])); }
        // OXT (src/foldcomp.cpp:474-482): the last atom of the span
        const uint32_t la = a_end - 1;
        const bool has_oxt = a_end > a_first && in.atom_code[la] == FCZ_ATOM_OXT;
        uint8_t* o = rec + RL.o_oxt;
        o[0] = has_oxt ? 1 : 0;
        st_f32(o + 1, has_oxt ? in.x[la] : 0.0f);
        st_f32(o + 5, has_oxt ? in.y[la] : 0.0f);
        st_f32(o + 9, has_oxt ? in.z[la] : 0.0f);
        st_f32(rec + RL.o_tmp, qmin[6]);
        st_f32(rec + RL.o_tmp + 4, qcont[6]);
        if (status) status[c] = FCZ_OK;
    }
}


// =====================================================================================================================
// Several short chains per wavefront: one chain per G-lane group (G = 16: a DPP row, four chains to a wavefront)
// =====================================================================================================================
// A chain costs k_compress_pack ~1 250 VALU wave-instructions before its first residue (validation, anchors, seven min / max
// reductions, record layout, title, header: profiles/r5_short_chains.txt), all of it executed by a whole wavefront whatever the
// chain's length. Here the lanes of a group carry their own chain's scalars, so that cost is shared by the four chains of the
// wavefront; the reductions stay inside a group (four DPP steps cover a row of 16). A chain longer than the group takes several
// rounds of G residues (lane sub holds residues sub, sub + G, ...): every chain of up to 64 residues goes four to a wavefront.
// Same arithmetic, same order, same first-occurrence rules as compress_pack_chain (the results are the same bits).
template <int G> __device__ __forceinline__ float grp_min_f32(float v) {
    v = __builtin_fminf(v, dpp_f32<0xB1, 0xf>(v)); v = __builtin_fminf(v, dpp_f32<0x4E, 0xf>(v));
    v = __builtin_fminf(v, dpp_f32<0x141, 0xf>(v)); v = __builtin_fminf(v, dpp_f32<0x140, 0xf>(v));
    if (G == 32) v = __builtin_fminf(v, __shfl_xor(v, 16, WAVE));
    return v;
}
template <int G> __device__ __forceinline__ float grp_max_f32(float v) {
    v = __builtin_fmaxf(v, dpp_f32<0xB1, 0xf>(v)); v = __builtin_fmaxf(v, dpp_f32<0x4E, 0xf>(v));
    v = __builtin_fmaxf(v, dpp_f32<0x141, 0xf>(v)); v = __builtin_fmaxf(v, dpp_f32<0x140, 0xf>(v));
    if (G == 32) v = __builtin_fmaxf(v, __shfl_xor(v, 16, WAVE));
    return v;
}
template <int G> __device__ __forceinline__ int grp_min_i32(int v) {
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) { const int o = __shfl_xor(v, d, WAVE); v = o < v ? o : v; }
    return v;
}
template <int G> __device__ __forceinline__ uint32_t grp_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, WAVE);
    return v;
}
// std::min_element / max_element over the group's values with their positions (first occurrence wins): only when an extremum is a zero
template <int G> __device__ __forceinline__ lo_hi grp_first_extrema(ext mn, ext mx) {
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) {
        const float v1 = __shfl_xor(mn.v, d, WAVE); const uint32_t i1 = (uint32_t)__shfl_xor((int)mn.i, d, WAVE);
        const float v2 = __shfl_xor(mx.v, d, WAVE); const uint32_t i2 = (uint32_t)__shfl_xor((int)mx.i, d, WAVE);
        ext_min_upd(mn, v1, i1); ext_max_upd(mx, v2, i2);
    }
    return lo_hi{mn.v, mx.v};
}

// One chain per G-lane group, U rounds of G residues each (a chain of up to G x U residues): lane `sub` of the group holds residues
// sub, sub + G, ... The group's scalars live in its lanes; `live` false = a group without a chain (it runs along on chain 0's
// addresses with everything it would write switched off).
template <int G, int U>
__device__ __forceinline__ void compress_pack_rows(const fcz_chain_batch& in, const uint32_t c_in, const bool live, const uint32_t r0_in, const uint32_t n_in,
                                                   const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                   int32_t* __restrict__ status, float* __restrict__ ang, int keep_first_angle,
                                                   const uint32_t* __restrict__ nonfinite) {
    const uint32_t sub = (uint32_t)(threadIdx.x & (G - 1));
    const uint32_t c = live ? c_in : 0u, r0 = live ? r0_in : 0u, n = live ? n_in : 0u;
    const uint32_t title_len = in.title_off[c + 1] - in.title_off[c];
    const uint32_t thr = (uint32_t)in.anchor_threshold;
    uint8_t* rec = out + out_off[c];
    const uint32_t rec_size = (uint32_t)(out_off[c + 1] - out_off[c]);
    const int32_t h_first_res = in.first_res_index[c], h_first_atom = in.first_atom_index[c];
    const char h_chain = in.chain_id[c];
    const size_t R = in.n_residues;
    const uint32_t a_first = in.atom_off[r0], a_end = in.atom_off[r0 + n];
    const uint32_t rl = r0 + (n ? n - 1 : 0);
    const uint32_t h_rc_first = in.res_code[r0 < R ? r0 : R - 1], h_rc_last = in.res_code[rl < R ? rl : R - 1];
    const uint32_t m = n ? n - 1 : 0;
    float* a_arr = ang + (r0 < R ? r0 : (R ? R - 1 : 0));
    // ---- the group's values: every load issued up front from clamped indices ----
    float va[7][U];
    uint32_t rcs[U], o0[U], o2[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        const uint32_t kw = k < m ? k : (m ? m - 1 : 0), kr = k < n ? k : (n ? n - 1 : 0);
        const size_t rr = (size_t)r0 + kr < R ? (size_t)r0 + kr : R - 1;
#pragma unroll
        for (int q = 0; q < 6; q++) va[q][u] = a_arr[(size_t)q * R + kw];
        va[6][u] = in.bfac_ca[rr];
        rcs[u] = in.res_code[rr];
        o0[u] = in.atom_off[rr];
        o2[u] = in.atom_off[r0 + (k + 2 < n ? k + 2 : n)];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (!__any((uint32_t)u * G < m)) continue;        // a round none of the wavefront's chains reaches
        va[0][u] = dec_angle<0>(va[0][u]); va[1][u] = dec_angle<1>(va[1][u]); va[2][u] = dec_angle<2>(va[2][u]);
        va[3][u] = dec_angle<3>(va[3][u]); va[4][u] = dec_angle<4>(va[4][u]); va[5][u] = dec_angle<5>(va[5][u]);
    }

    // ---- validation, as compress_pack_chain orders it ----
    int bad = (n < 2) ? FCZ_E_TOO_SHORT : (thr < 1 ? FCZ_E_INVALID_ARG : 0);
    if (!bad && (n > 65535u || n / thr + 2u > 255u)) bad = FCZ_E_INVALID_ARG;
    if (!bad && ((nonfinite[c >> 5] >> (c & 31u)) & 1u)) bad = FCZ_E_NONFINITE;
    uint32_t nsc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        if (k < n) {
            if (!res_code_ok(rcs[u])) bad = bad ? bad : FCZ_E_RESIDUE;
            if (o2[u] - o0[u] > (uint32_t)CK_CAP) bad = bad ? bad : FCZ_E_INVALID_ARG;
            if (nonfinite_f32(va[6][u])) bad = bad ? bad : FCZ_E_NONFINITE;
            nsc += fcz_res_natoms[rcs[u] < 24 ? rcs[u] : 23] - 3;
        }
    }
    bad = grp_min_i32<G>(bad);
    nsc = grp_sum_u32<G>(nsc);
    {
        const uint32_t zlim = (live && bad) ? rec_size : 0u;         // a refused chain leaves zeros
        for (uint32_t i = sub; __any(i < zlim); i += G) if (i < zlim) rec[i] = 0;
        if (live && bad && sub == 0 && status) status[c] = bad;
    }
    const bool on = live && !bad;

    const uint32_t n_anchor = n / (thr ? thr : 1u) + 2;
    const uint32_t interval = n / (n_anchor - 1);
    const rec_layout RL = make_layout(n, n_anchor, title_len, nsc);

    // ---- anchors (Foldcomp::_setAnchor src/foldcomp.cpp:745-761, written :1045-1059): one lane per anchor ----
    for (uint32_t sl = sub; __any(on && sl < n_anchor); sl += G) {
        if (!(on && sl < n_anchor)) continue;
        const uint32_t ka = (sl + 1 < n_anchor) ? sl * interval : n - 1;
        const uint32_t lo = in.atom_off[r0 + ka], hi = in.atom_off[r0 + ka + 1];
        uint32_t at[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint32_t codes[8];
#pragma unroll
        for (int j = 0; j < 8; j++) codes[j] = (lo + j < hi) ? (uint32_t)in.atom_code[lo + j] : 255u;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            at[0] = codes[j] == 0u ? lo + j : at[0];
            at[1] = codes[j] == 1u ? lo + j : at[1];
            at[2] = codes[j] == 2u ? lo + j : at[2];
        }
        if (__builtin_expect(at[0] == 0xffffffffu || at[1] == 0xffffffffu || at[2] == 0xffffffffu, 0)) {
            for (uint32_t i = lo + 8; i < hi; i++) {
                const uint32_t code = in.atom_code[i];
                if (code == 0u && at[0] == 0xffffffffu) at[0] = i;
                if (code == 1u && at[1] == 0xffffffffu) at[1] = i;
                if (code == 2u && at[2] == 0xffffffffu) at[2] = i;
            }
        }
        v3 p[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const bool have = at[q] != 0xffffffffu;
            const uint32_t i = have ? at[q] : lo;
            const bool ok = have && lo < hi;
            p[q] = ok ? v3{in.x[i], in.y[i], in.z[i]} : v3{0.f, 0.f, 0.f};
        }
        uint8_t* q = rec + RL.o_anchor + 36 * sl;
        st_f32(q, p[0].x); st_f32(q + 4, p[0].y); st_f32(q + 8, p[0].z);
        st_f32(q + 12, p[1].x); st_f32(q + 16, p[1].y); st_f32(q + 20, p[1].z);
        st_f32(q + 24, p[2].x); st_f32(q + 28, p[2].y); st_f32(q + 32, p[2].z);
        st_u32(rec + RL.o_aidx + 4 * sl, ka);
        if (keep_first_angle && sl == 0) a_arr[3 * R + (n - 1)] = bond_angle_deg(p[0], p[1], p[2]);
    }

    // ---- per-chain quantiser parameters (Discretizer::Discretizer src/discretizer.cpp:22-33) ----
    const float kInf = __builtin_huge_valf();
    float qmin[7], qdisc[7], qcont[7];
    const float nbins[7] = {4095.0f, 4095.0f, 2047.0f, 255.0f, 255.0f, 255.0f, 255.0f};
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const uint32_t cntq = (q < 6) ? m : n;
        float lo = kInf, hi = -kInf;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool act = (uint32_t)u * G + sub < cntq;
            lo = __builtin_fminf(lo, act ? va[q][u] : kInf); hi = __builtin_fmaxf(hi, act ? va[q][u] : -kInf);
        }
        lo = grp_min_f32<G>(lo); hi = grp_max_f32<G>(hi);
        const float first = __shfl(va[q][0], (int)((threadIdx.x & 63u) & ~(uint32_t)(G - 1)), WAVE);     // element 0 of the group's array
        if (__builtin_expect(__any(on && (lo == 0.0f || hi == 0.0f)), 0)) {
            ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = (uint32_t)u * G + sub;
                if (k < cntq) { ext_min_upd(mn, va[q][u], k); ext_max_upd(mx, va[q][u], k); }
            }
            const lo_hi e = grp_first_extrema<G>(mn, mx);
            if (lo == 0.0f || hi == 0.0f) { lo = e.lo; hi = e.hi; }
        }
        qmin[q] = lo; qdisc[q] = nbins[q] / (hi - lo); qcont[q] = (hi - lo) / nbins[q];
        // a NaN at the head is minimum and maximum at once, and x86-64 hands its bits through the subtraction and the division
        // (NaN - NaN = the first operand, quiet already): the record's header holds those bits twice (acos_deg_exact makes them)
        if (__builtin_expect(first != first, 0)) { qmin[q] = first; qdisc[q] = first; qcont[q] = first; }
    }
    // ---- the packed words (src/foldcomp.cpp:582-601, convertBackboneChainToBytes :33-52) + B-factor bytes ----
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        if (keep_first_angle && on && k < m) {
#pragma unroll
            for (int q = 0; q < 6; q++) a_arr[(size_t)q * R + k] = va[q][u];
        }
        if (on && k < n) {
            uint32_t om = 0, ps = 0, ph = 0, b1 = 0, b2 = 0, b3 = 0;
            if (k < m) {
                ph = quant_round(va[0][u], qmin[0], qdisc[0]) & 0xfffu;
                ps = quant_round(va[1][u], qmin[1], qdisc[1]) & 0xfffu;
                om = quant_round(va[2][u], qmin[2], qdisc[2]) & 0x7ffu;
                b3 = quant_round(va[3][u], qmin[3], qdisc[3]) & 0xffu;
                b1 = quant_round(va[4][u], qmin[4], qdisc[4]) & 0xffu;
                b2 = quant_round(va[5][u], qmin[5], qdisc[5]) & 0xffu;
            }
            const uint32_t w0 = ((rcs[u] & 0x1fu) << 3) | (om >> 8), w1 = om & 0xffu, w2 = ps >> 4,
                           w3 = ((ps & 0xfu) << 4) | (ph >> 8), w4 = ph & 0xffu;
            const uint64_t word = (uint64_t)w0 | ((uint64_t)w1 << 8) | ((uint64_t)w2 << 16) | ((uint64_t)w3 << 24) |
                                  ((uint64_t)w4 << 32) | ((uint64_t)b1 << 40) | ((uint64_t)b2 << 48) | ((uint64_t)b3 << 56);
            st_u64(rec + RL.o_words + 8 * (size_t)k, word);
            rec[RL.o_tbytes + k] = (uint8_t)quant_round(va[6][u], qmin[6], qdisc[6]);
        }
    }
    {
        const uint32_t tlim = on ? title_len : 0u;
        const uint32_t t0 = in.title_off[c];
        for (uint32_t i = sub; __any(i < tlim); i += G) if (i < tlim) rec[RL.o_title + i] = (uint8_t)in.titles[t0 + i];
    }
    // ---- header (CompressedFileHeader src/foldcomp.h:118-136; get_header src/foldcomp.cpp:1340) ----
    if (on && sub == 0) {
        rec[0] = 'F'; rec[1] = 'C'; rec[2] = 'M'; rec[3] = 'P';
        uint8_t* h = rec + 4;
        st_u16(h + 0, n);
        st_u16(h + 2, a_end - a_first);
        st_u16(h + 4, (uint32_t)h_first_res);
        st_u16(h + 6, (uint32_t)h_first_atom);
        h[8] = (uint8_t)n_anchor;
        h[9] = (uint8_t)h_chain;
        h[10] = 0; h[11] = 0;
        st_u32(h + 12, nsc);
        h[16] = (uint8_t)fcz_res1[h_rc_first];
        h[17] = (uint8_t)fcz_res1[h_rc_last];
        h[18] = 0; h[19] = 0;
        st_u32(h + 20, title_len);
        auto x86_nan = [](float v) { return v; };          // (the bits acos_deg_exact gave the chain's first angle: see k_compress_pack)
#pragma unroll
        for (int q = 0; q < 6; q++) { st_f32(h + 24 + 4 * q, x86_nan(qmin[q])); st_f32(h + 48 + 4 * q, x86_nan(qcont[q])); }
        const uint32_t la = a_end - 1;
        const bool has_oxt = a_end > a_first && in.atom_code[la] == FCZ_ATOM_OXT;
        uint8_t* o = rec + RL.o_oxt;
        o[0] = has_oxt ? 1 : 0;
        st_f32(o + 1, has_oxt ? in.x[la] : 0.0f);
        st_f32(o + 5, has_oxt ? in.y[la] : 0.0f);
        st_f32(o + 9, has_oxt ? in.z[la] : 0.0f);
        st_f32(rec + RL.o_tmp, qmin[6]);
        st_f32(rec + RL.o_tmp + 4, qcont[6]);
        if (status) status[c] = FCZ_OK;
    }
}

// Chains of 2 .. CP_SHORT residues belong to k_compress_pack_rows, everything else (incl. what is refused) to k_compress_pack.
#ifndef FCZ_PACK_ROWS_MAX_ROUNDS
#define FCZ_PACK_ROWS_MAX_ROUNDS 8     // chains of up to 16 x this many residues go four to a wavefront (8: up to 128; 202 VGPRs for that class)
#endif
constexpr uint32_t CP_SHORT = 16u * FCZ_PACK_ROWS_MAX_ROUNDS;
#ifndef FCZ_PACK_CLASSES
#define FCZ_PACK_CLASSES 1
#endif


__global__ __launch_bounds__(BLOCK, 3) void k_compress_pack(fcz_chain_batch in, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                         int32_t* __restrict__ status, float* __restrict__ ang, int keep_first_angle,
                                                         const uint32_t* __restrict__ nonfinite) {
    const int wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= in.n_chains) return;
    const uint32_t r0 = in.res_off[c], n = in.res_off[c + 1] - r0;
    if (n >= 2 && n <= CP_SHORT) return;                      // k_compress_pack_rows'
#if FCZ_PACK_CLASSES
    // rounds of 64 residues held in registers, by length class (wave-uniform): a 100-residue chain does not issue the loads,
    // reductions and stores of a 350-residue one
    if (n <= 4u * WAVE) compress_pack_chain<4>(in, c, r0, n, out_off, out, status, ang, keep_first_angle, nonfinite);
    else
#endif
    compress_pack_chain<6>(in, c, r0, n, out_off, out, status, ang, keep_first_angle, nonfinite);
}

// Short chains (2 .. CP_SHORT residues: peptides, fragments, the low end of a metagenomic set). One wavefront per chain costs a
// short chain what it costs a long one -- ~1 250 VALU wave-instructions of validation, anchors, reductions, layout and header before
// its first residue -- so k_compress_pack_rows<U> puts FOUR chains into a wavefront (below); k_compress_pack leaves them alone.
constexpr int CP_CHUNK = 16;
// Chains of 2 .. CP_SHORT (= 16 x FCZ_PACK_ROWS_MAX_ROUNDS = 128) residues, FOUR to a wavefront: one chain per 16-lane DPP row, in
// U = 1, 2, 4 or 8 rounds of 16 residues by length class (2..16, 17..32, 33..64, 65..128; one launch per class). A persistent grid:
// wavefront w takes the chunks of CP_CHUNK consecutive chains w, w + W, ...; one coalesced load gives the chunk's lengths, a ballot
// the chains of the launch's class, which are handed to the rows four at a time. A batch without short chains costs one load per
// 16 chains per launch.
template <int U>
__device__ __forceinline__ void pack_rows_class(unsigned long long todo, const uint32_t c0, const uint32_t ro, const uint32_t nn, const fcz_chain_batch& in,
                                                const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out, int32_t* __restrict__ status,
                                                float* __restrict__ ang, int keep_first_angle, const uint32_t* __restrict__ nonfinite) {
    const uint32_t g16 = (uint32_t)(threadIdx.x & 63) >> 4;
    while (todo) {
        uint32_t cc = 0, rr0 = 0, rn = 0; bool live = false;
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            if (!todo) break;
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t r0g = (uint32_t)__builtin_amdgcn_readlane((int)ro, l), ng = (uint32_t)__builtin_amdgcn_readlane((int)nn, l);
            if (g16 == g) { cc = c0 + (uint32_t)l; rr0 = r0g; rn = ng; live = true; }
        }
        compress_pack_rows<16, U>(in, cc, live, rr0, rn, out_off, out, status, ang, keep_first_angle, nonfinite);
    }
}
// (one kernel per length class: the one-round form fits four wavefronts per SIMD, the four-round form three)
template <int U>
__global__ __launch_bounds__(BLOCK, (U >= 8 ? 2 : (U >= 4 ? 3 : 4))) void k_compress_pack_rows(fcz_chain_batch in, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                                                                      int32_t* __restrict__ status, float* __restrict__ ang, int keep_first_angle,
                                                                      const uint32_t* __restrict__ nonfinite) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    const uint32_t n_chunks = (in.n_chains + CP_CHUNK - 1) / CP_CHUNK;
    constexpr uint32_t LO = U == 1 ? 2u : (uint32_t)(U / 2) * 16u + 1u, HI = (uint32_t)U * 16u;     // 2..16, 17..32, 33..64
    for (uint32_t ch = blockIdx.x * WAVES_PER_BLOCK + wave; ch < n_chunks; ch += n_waves) {
        const uint32_t c0 = ch * CP_CHUNK;
        const uint32_t ci = c0 + (uint32_t)lane;
        const uint32_t ro = in.res_off[ci <= in.n_chains ? ci : in.n_chains];          // lanes 0 .. CP_CHUNK: the chunk's offsets
        const uint32_t nn = (uint32_t)__shfl_down((int)ro, 1, WAVE) - ro;
        const unsigned long long todo = __ballot(lane < CP_CHUNK && ci < in.n_chains && nn >= LO && nn <= HI);
        if (todo) pack_rows_class<U>(todo, c0, ro, nn, in, out_off, out, status, ang, keep_first_angle, nonfinite);
    }
}

}  // namespace fcz
