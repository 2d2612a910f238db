// fcz_compress.h -- k_compress_tiled: the compress kernel (Foldcomp::preprocess/compress/writeStream,
// reference src/foldcomp.cpp:450-606, 1038-1109).
//
// One wavefront per chain. The chain is walked in tiles of up to 64 residues:
//   1. the tile's atoms (a contiguous range of the SoA arrays, plus the next residue for the backbone
//      windows that straddle the tile edge) arrive with coalesced 16-byte loads that were issued one tile
//      earlier (register prefetch) and are parked in LDS as {x, y, z, code} records;
//   2. every residue lane reads its atoms' codes from LDS (all reads in flight at once) and records, per
//      canonical slot, the index of the first atom with that name (findFirstAtomCoords semantics,
//      reference src/sidechain.cpp:140-147); absent names point at an all-zero record, which is what the
//      reference reads for a missing atom;
//   3. all angle evaluations of the tile form ONE flat work list -- 3 backbone dihedrals + 3 backbone bond
//      angles per residue window, then one dihedral per side-chain atom -- and lanes take items
//      round-robin: every iteration has 64 busy lanes running the same code. Side-chain items are
//      numbered exactly like the FCZ side-chain byte stream, so their byte stores are consecutive.
// Backbone angles go to per-chain scratch (coalesced per angle type); after the tile loop the wave reduces
// min/max per type and quantises + packs the 8-byte words from registers.
#pragma once
#include "fcz_kernels.h"

namespace fcz {

// Optional phase timer (build with -DFCZ_PROFILE_PHASES): per-wave s_memtime deltas summed into g_phase_cycles.
#ifdef FCZ_PROFILE_PHASES
__device__ unsigned long long g_phase_cycles[16];
#define PH_DECL unsigned long long ph_t = __builtin_readcyclecounter(); unsigned long long ph_acc[10] = {0,0,0,0,0,0,0,0,0,0};
#define PH_MARK(i) { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; }
#define PH_FLUSH if (lane == 0) { for (int i_ = 0; i_ < 10; i_++) atomicAdd(&g_phase_cycles[i_], ph_acc[i_]); }
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH
#endif

#ifndef FCZ_COMPRESS_MIN_WAVES
#define FCZ_COMPRESS_MIN_WAVES 2
#endif
constexpr int CT_CAP = 768;          // staged atom records per tile (typical tile: 65 residues * 8.4 atoms = 545)
constexpr int CT_ZERO = CT_CAP;      // index of the all-zero record (missing atoms)
constexpr int CT_NV = CT_CAP / (4 * WAVE);   // float4 staging rounds per array
constexpr int CT_SB = 8;             // tiles per metadata super-block (512 residues)

struct alignas(16) compress_tile_lds {
    float4 atom[CT_CAP + 1];                       // {x, y, z, code bits}
    uint16_t idx[66][16];                          // [residue in tile][canonical slot] -> atom record index
    uint16_t scpre[66];                            // tile-local exclusive prefix of side-chain torsion counts
    uint8_t rc[68];
    uint8_t item_res[64 * 11];                     // side-chain item -> residue in tile
    float out_bb[6][WAVE];                         // backbone angles of the tile, flushed with coalesced stores
    alignas(4) uint8_t out_sc[64 * 11 + 4];        // side-chain bytes of the tile
};

__device__ __forceinline__ v3 tile_atom(const compress_tile_lds& L, uint32_t res, uint32_t slot) {
    const float4 a = L.atom[L.idx[res][slot]];
    return v3{a.x, a.y, a.z};
}

// ---- wave reductions on the DPP cross-lane network (no LDS traffic) -------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// four DPP steps leave every lane with the extremum of its row of 16; the four rows are combined through
// scalar reads
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
struct lo_hi { float lo, hi; };
__device__ __noinline__ lo_hi first_extrema(const float* __restrict__ src, uint32_t cnt, int lane) {
    const float kInf = __builtin_huge_valf();
    ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
    for (uint32_t k = lane; k < cnt; k += WAVE) { const float v = src[k]; ext_min_upd(mn, v, k); ext_max_upd(mx, v, k); }
    return lo_hi{wave_ext_min(mn), wave_ext_max(mx)};
}

// tile geometry shared by the prefetch and the consumer
struct tile_ext { uint32_t T, nres, A0, cnt; bool look; };

// last few atoms of the whole batch: element-wise loads that never run past the end of the arrays.
// Returned by value: taking the address of the prefetch registers would demote them to scratch memory.
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

__global__ __launch_bounds__(BLOCK, FCZ_COMPRESS_MIN_WAVES)
void k_compress_tiled(fcz_chain_batch in, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
                      int32_t* __restrict__ status, float* __restrict__ ang, int keep_first_angle) {
    __shared__ compress_tile_lds s_tile[WAVES_PER_BLOCK];
    __shared__ uint8_t s_slot_of[FCZ_N_RES_CODES][40];  // atom code -> canonical slot, 255 = not in residue
    __shared__ uint16_t s_prev[FCZ_N_RES_CODES][FCZ_MAX_RES_ATOMS];
    __shared__ uint8_t s_natoms[32];

    for (int i = threadIdx.x; i < FCZ_N_RES_CODES * 40; i += BLOCK) (&s_slot_of[0][0])[i] = 255;
    if (threadIdx.x < 32) s_natoms[threadIdx.x] = fcz_res_natoms[threadIdx.x < 24 ? threadIdx.x : 23];
    __syncthreads();
    for (int i = threadIdx.x; i < FCZ_N_RES_CODES * FCZ_MAX_RES_ATOMS; i += BLOCK) {
        int rc = i / FCZ_MAX_RES_ATOMS, j = i % FCZ_MAX_RES_ATOMS;
        if (j < fcz_res_natoms[rc]) s_slot_of[rc][fcz_res_atom[rc][j]] = (uint8_t)j;
        s_prev[rc][j] = fcz_res_prev[rc][j];
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= in.n_chains) return;
    compress_tile_lds& L = s_tile[wave];
    PH_DECL

    const uint32_t r0 = in.res_off[c], n = in.res_off[c + 1] - r0;
    const uint32_t title_len = in.title_off[c + 1] - in.title_off[c];
    const uint32_t thr = (uint32_t)in.anchor_threshold;
    uint8_t* rec = out + out_off[c];
    const uint32_t rec_size = (uint32_t)(out_off[c + 1] - out_off[c]);
    // header scalars: issued now, consumed at the very end
    const int32_t h_first_res = in.first_res_index[c], h_first_atom = in.first_atom_index[c];
    const char h_chain = in.chain_id[c];
    const uint32_t a_first = in.atom_off[r0], a_end = in.atom_off[r0 + n];
    const uint32_t h_rc_first = in.res_code[r0], h_rc_last = in.res_code[r0 + (n ? n - 1 : 0)];

    // ---- validation (the reference aborts on these inputs) + total side-chain torsion count ----
    int bad = (n < 2) ? FCZ_E_TOO_SHORT : (thr < 2 ? FCZ_E_INVALID_ARG : 0);
    uint32_t nsc = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += 8 * WAVE) {   // 8 independent loads in flight per round trip
        uint32_t rcs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t k = k0 + u * WAVE + lane; rcs[u] = (k < n) ? in.res_code[r0 + k] : 0xffu; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (rcs[u] == 0xffu) continue;
            if (!res_code_ok(rcs[u])) bad = bad ? bad : FCZ_E_RESIDUE;
            nsc += s_natoms[rcs[u] & 31u] - 3;
        }
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { int o = __shfl_xor(bad, d, WAVE); bad = o < bad ? o : bad; }
    if (bad) {
        for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
        if (lane == 0 && status) status[c] = bad;
        return;
    }
    nsc = wave_sum(nsc);
    PH_MARK(0)

    const uint32_t m = n - 1;
    const uint32_t n_anchor = n / thr + 2;
    const uint32_t interval = n / (n_anchor - 1);
    const rec_layout RL = make_layout(n, n_anchor, title_len, nsc);
    const size_t R = in.n_residues;
    float* a_arr = ang + r0;   // array q of this chain = a_arr + q*R : phi psi omega n_ca_c ca_c_n c_n_ca
    const float sc_min = -180.0f, sc_disc = 255.0f / (180.0f - (-180.0f));  // FixedAngleDiscretizer(255)

    if (lane == 0) L.atom[CT_ZERO] = float4{0.f, 0.f, 0.f, 0.f};

    // Metadata registers of the current super-block, rotated so that index 0 is always the current tile:
    // mo[u] = atom_off of residue (tile base + u*64 + lane), mr[u] = its residue code.
    uint32_t mo[CT_SB + 1], mr[CT_SB + 1];
    auto load_meta = [&](uint32_t sb_base) {
#pragma unroll
        for (int u = 0; u <= CT_SB; u++) { const uint32_t k = sb_base + u * WAVE + lane; mo[u] = in.atom_off[r0 + (k <= n ? k : n)]; }
#pragma unroll
        for (int u = 0; u <= CT_SB; u++) { const uint32_t k = sb_base + u * WAVE + lane; mr[u] = (k < n) ? in.res_code[r0 + k] : 23u; }
    };
    // extent of the tile that starts at residue `base` (metadata in o_lane = mo[0], o_next = mo[1])
    auto tile_extent = [&](uint32_t base, uint32_t o_lane, uint32_t o_next) -> tile_ext {
        tile_ext e;
        e.T = (n - base < (uint32_t)WAVE) ? (n - base) : (uint32_t)WAVE;
        e.A0 = __shfl(o_lane, 0, WAVE);
        const uint32_t o64 = __shfl(o_next, 0, WAVE), o65 = __shfl(o_next, 1, WAVE);
        e.look = base + e.T < n;
        const uint32_t last = e.T + (e.look ? 1u : 0u);
        uint32_t A1 = (last == 65) ? o65 : (last == 64 ? o64 : __shfl(o_lane, (int)(last & 63u), WAVE));
        if (A1 - e.A0 > (uint32_t)CT_CAP) {
            // unusually atom-rich residues (e.g. explicit hydrogens): shrink the tile. Residues [0,t) plus the
            // look-ahead residue t fit iff the ends of residues 0..t all lie within the staging capacity.
            uint32_t my_end = __shfl_down(o_lane, 1, WAVE);
            if (lane == 63) my_end = o64;
            const bool fits = (base + lane < n) && (my_end - e.A0 <= (uint32_t)CT_CAP);
            const unsigned long long fm = __ballot(fits);
            const uint32_t lead = (fm == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~fm);
            if (lead < 2) { e.T = 0; e.nres = 0; e.cnt = 0; return e; }   // not a protein chain
            e.T = lead - 1; e.look = true;
            A1 = (e.T + 1 == 64) ? o64 : __shfl(o_lane, (int)(e.T + 1), WAVE);
        }
        e.nres = e.T + (e.look ? 1u : 0u);
        e.cnt = A1 - e.A0;
        return e;
    };
    // coalesced 16-byte loads of a tile's atoms into registers
    float4 px[CT_NV], py[CT_NV], pz[CT_NV];
    uint32_t pc[CT_NV];
    auto issue_atoms = [&](const tile_ext& e) {
        const bool whole = (size_t)e.A0 + CT_CAP + 4 <= (size_t)in.n_atoms;   // 16-byte reads stay inside the arrays
#pragma unroll
        for (int u = 0; u < CT_NV; u++) {
            const uint32_t i4 = 4 * (u * WAVE + lane);
            px[u] = py[u] = pz[u] = float4{0.f, 0.f, 0.f, 0.f}; pc[u] = 0;
            if (i4 < e.cnt) {
                if (whole || (size_t)e.A0 + i4 + 4 <= (size_t)in.n_atoms) {
                    px[u] = ld_f4(in.x + e.A0 + i4);
                    py[u] = ld_f4(in.y + e.A0 + i4);
                    pz[u] = ld_f4(in.z + e.A0 + i4);
                    pc[u] = ld_u32(in.atom_code + e.A0 + i4);
                } else {
                    const atom_quad q = load_atoms_tail(in.x, in.y, in.z, in.atom_code, e.A0 + i4, e.cnt - i4);
                    px[u] = q.x; py[u] = q.y; pz[u] = q.z; pc[u] = q.c;
                }
            }
        }
    };

    uint32_t sc_base = 0;       // side-chain bytes emitted so far
    uint32_t base = 0;
    uint32_t slot_in_sb = 0;    // tiles consumed from the current super-block
    uint32_t last_cnt = 1;      // staged atoms of the final tile (for the OXT test)
    tile_ext cur; cur.T = 0; cur.nres = 0; cur.cnt = 0; cur.A0 = 0; cur.look = false;
    // Software pipeline with a single copy of every stage: pass -1 only prefetches tile 0; pass t >= 0 parks
    // tile t in LDS, prefetches tile t+1 and then computes tile t.
    for (int pass = -1;; pass++) {
        const bool live = pass >= 0;
        if (live && cur.T == 0) {  // one residue plus its successor exceed the staging capacity
            for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
            if (lane == 0 && status) status[c] = FCZ_E_INVALID_ARG;
            return;
        }
        const uint32_t T = cur.T, cnt = cur.cnt, A0 = cur.A0;
        const bool look = cur.look;
        const uint32_t o_lane = mo[0], rc_lane = mr[0];
        uint32_t o64r = 0, o65r = 0, rc64 = 23;
        if (live) {
            o64r = __shfl(mo[1], 0, WAVE) - A0; o65r = __shfl(mo[1], 1, WAVE) - A0;
            rc64 = __shfl(mr[1], 0, WAVE);          // code of residue base+64
            // ---- park the prefetched atoms in LDS as {x,y,z,code} records ----
#pragma unroll
            for (int u = 0; u < CT_NV; u++) {
                const uint32_t i4 = 4 * (u * WAVE + lane);
                if (i4 < cnt) {
                    L.atom[i4 + 0] = float4{px[u].x, py[u].x, pz[u].x, __uint_as_float(pc[u] & 0xffu)};
                    L.atom[i4 + 1] = float4{px[u].y, py[u].y, pz[u].y, __uint_as_float((pc[u] >> 8) & 0xffu)};
                    L.atom[i4 + 2] = float4{px[u].z, py[u].z, pz[u].z, __uint_as_float((pc[u] >> 16) & 0xffu)};
                    L.atom[i4 + 3] = float4{px[u].w, py[u].w, pz[u].w, __uint_as_float(pc[u] >> 24)};
                }
            }
            L.rc[lane] = (uint8_t)rc_lane;
        }
        // ---- metadata rotation + prefetch of the next tile (loads land while this tile computes) ----
        const uint32_t base_next = live ? base + T : 0u;
        const bool have_next = base_next < n;
        tile_ext nxt; nxt.T = 0; nxt.nres = 0; nxt.cnt = 0; nxt.A0 = 0; nxt.look = false;
        if (have_next) {
            if (live && T == (uint32_t)WAVE && slot_in_sb + 1 < CT_SB) {
#pragma unroll
                for (int u = 0; u < CT_SB; u++) { mo[u] = mo[u + 1]; mr[u] = mr[u + 1]; }
                slot_in_sb++;
            } else {
                // first tile, super-block exhausted, or a shrunken tile shifted the grid: (re)load metadata
                slot_in_sb = 0;
                load_meta(base_next);
            }
            nxt = tile_extent(base_next, mo[0], mo[1]);
            if (nxt.T) issue_atoms(nxt);
        }
        PH_MARK(2)
        if (!live) { cur = nxt; continue; }
        __builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in issue order: no memory fence needed

        // ---- slot index table: first atom of each canonical name ----
        {
            const uint32_t a_lo = o_lane - A0;
            uint32_t a_hi = __shfl_down(o_lane, 1, WAVE) - A0;
            if (lane == 63) a_hi = o64r;
            const uint32_t nres_t = T + (look ? 1u : 0u);
            // lanes 0..T-1 own the tile's residues; the look-ahead residue T is built by lane T when the tile is
            // short, by lane 0 (second trip, offsets from the next metadata column) when T == 64
            for (uint32_t rr = lane; rr < nres_t; rr += WAVE) {
                const bool second = rr >= (uint32_t)WAVE;
                const uint32_t rc = second ? rc64 : rc_lane;
                const uint32_t lo = second ? o64r : a_lo, hi = second ? o65r : a_hi;
                // all code reads of the residue in flight together, then all slot lookups
                uint32_t codes[16], slots[16];
#pragma unroll
                for (int j = 0; j < 16; j++) codes[j] = (lo + j < hi) ? __float_as_uint(L.atom[lo + j].w) : 255u;
#pragma unroll
                for (int j = 0; j < 16; j++) slots[j] = (codes[j] < 40u) ? s_slot_of[rc][codes[j]] : 255u;
                const uint32_t zz = (uint32_t)CT_ZERO | ((uint32_t)CT_ZERO << 16);
                uint4* row = reinterpret_cast<uint4*>(&L.idx[rr][0]);
                row[0] = uint4{zz, zz, zz, zz};
                row[1] = uint4{zz, zz, zz, zz};
                uint32_t filled = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t sl = slots[j];
                    if (sl != 255u && !((filled >> sl) & 1u)) { filled |= 1u << sl; L.idx[rr][sl] = (uint16_t)(lo + j); }
                }
                for (uint32_t i = lo + 16; i < hi; i++) {   // residues with more than 16 atoms (explicit hydrogens)
                    const uint32_t code = __float_as_uint(L.atom[i].w);
                    const uint32_t sl = code < 40u ? s_slot_of[rc][code] : 255u;
                    if (sl != 255u && !((filled >> sl) & 1u)) { filled |= 1u << sl; L.idx[rr][sl] = (uint16_t)i; }
                }
            }
        }
        // side-chain item numbering of this tile
        uint32_t my_cnt = 0;
        if ((uint32_t)lane < T) my_cnt = s_natoms[rc_lane & 31u] - 3;
        uint32_t tile_sc;
        const uint32_t my_pre = wave_excl_scan(my_cnt, lane, &tile_sc);
        if ((uint32_t)lane < T) {
            L.scpre[lane] = (uint16_t)my_pre;
            for (uint32_t j = 0; j < my_cnt; j++) L.item_res[my_pre + j] = (uint8_t)lane;
        }
        __builtin_amdgcn_wave_barrier();
        PH_MARK(3)

        // ---- anchors (reference Foldcomp::_setAnchor src/foldcomp.cpp:745-761, written :1045-1059) ----
        if ((uint32_t)lane < T) {
            const uint32_t k = base + lane;
            const uint32_t ia = k / interval;
            const bool is_a = (k % interval == 0) && (ia < n_anchor - 1);
            const bool is_last = (k == n - 1);
            if (is_a || is_last) {
                const v3 N0 = tile_atom(L, lane, 0), CA0 = tile_atom(L, lane, 1), C0 = tile_atom(L, lane, 2);
                for (int rep = 0; rep < 2; rep++) {
                    if (rep == 0 ? !is_a : !is_last) continue;
                    const uint32_t slot = rep == 0 ? ia : n_anchor - 1;
                    uint8_t* p = rec + RL.o_anchor + 36 * slot;
                    st_f32(p, N0.x); st_f32(p + 4, N0.y); st_f32(p + 8, N0.z);
                    st_f32(p + 12, CA0.x); st_f32(p + 16, CA0.y); st_f32(p + 20, CA0.z);
                    st_f32(p + 24, C0.x); st_f32(p + 28, C0.y); st_f32(p + 32, C0.z);
                    st_u32(rec + RL.o_aidx + 4 * slot, k);
                }
            }
            if (keep_first_angle && k == 0)
                a_arr[3 * R + (n - 1)] = bond_angle_deg(tile_atom(L, 0, 0), tile_atom(L, 0, 1), tile_atom(L, 0, 2));
        }
        PH_MARK(4)

        // ---- the flat work list of the tile ----
        // items [0,3W): backbone dihedrals (q = item / W: psi, omega, phi), [3W,6W): backbone bond angles
        // (ca_c_n, c_n_ca, n_ca_c), then the side-chain dihedrals. W = residue windows of the tile (k < n-1).
        // Every item is "angle between two vectors": for a dihedral the two plane normals (getTorsionFromXYZ,
        // reference src/torsion_angle.cpp:50-94), for a bond angle the two bond vectors (angle(),
        // src/float3d.h:55-65; getBondAngles src/nerf.cpp:495-508); getCosineTheta + acos is one shared path.
        const uint32_t W = (base + T <= m) ? T : (m > base ? m - base : 0);
        const uint32_t n_items = 6 * W + tile_sc;
        for (uint32_t t0 = 0; t0 < n_items; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            if (t >= n_items) continue;
            uint32_t rA, sA, rB, sB, rC, sC, rD, sD;   // (residue in tile, slot) of the 4 (3) points
            bool is_dih;
            uint32_t dest;       // backbone: arr * 64 + residue in tile; side chain: byte index in the tile
            const bool is_bb = t < 6 * W;
            if (is_bb) {
                is_dih = t < 3 * W;
                const uint32_t tt = is_dih ? t : t - 3 * W;
                const uint32_t q = (tt >= 2 * W) ? 2u : (tt >= W ? 1u : 0u);
                const uint32_t res = tt - q * W;
                const uint32_t g0 = is_dih ? q : q + 1;           // first backbone atom of the window (0=N0 .. 5=C1)
                rA = res + (g0 >= 3 ? 1u : 0u);     sA = g0 >= 3 ? g0 - 3 : g0;
                rB = res + (g0 + 1 >= 3 ? 1u : 0u); sB = g0 + 1 >= 3 ? g0 - 2 : g0 + 1;
                rC = res + (g0 + 2 >= 3 ? 1u : 0u); sC = g0 + 2 >= 3 ? g0 - 1 : g0 + 2;
                rD = res + 1u;                       sD = g0;         // atom g0+3 (dihedrals only)
                // psi, omega, phi -> arrays 1, 2, 0 (split src/foldcomp.cpp:488-492);
                // ca_c_n, c_n_ca, n_ca_c -> arrays 4, 5, 3 (split :497-505)
                const uint32_t arr = is_dih ? ((q == 0) ? 1u : (q == 1 ? 2u : 0u)) : ((q == 0) ? 4u : (q == 1 ? 5u : 3u));
                dest = arr * WAVE + res;
            } else {
                // calculateTorsionAnglesInResidue, reference src/sidechain.cpp:149-168
                is_dih = true;
                const uint32_t ts = t - 6 * W;
                const uint32_t res = L.item_res[ts];
                const uint32_t j = 3 + ts - L.scpre[res];
                const uint32_t pk = s_prev[L.rc[res]][j];
                rA = rB = rC = rD = res;
                sA = pk & 15u; sB = (pk >> 4) & 15u; sC = (pk >> 8) & 15u; sD = j;
                dest = ts;
            }
            const v3 a = tile_atom(L, rA, sA), b = tile_atom(L, rB, sB), cc = tile_atom(L, rC, sC);
            v3 va, vb, d2;
            v3 u1{0.f, 0.f, 0.f}, u2 = u1;
            if (is_dih) {
                const v3 d = tile_atom(L, rD, sD);
                const v3 d1 = vsub(b, a);
                d2 = vsub(cc, b);
                const v3 d3 = vsub(d, cc);
                u1 = vcross(d1, d2); u2 = vcross(d2, d3);
                va = u1; vb = u2;
            } else {
                va = vsub(a, b); vb = vsub(cc, b);
                d2 = va;
            }
            const float ct = vcos_theta(va, vb);
            float v = acos_deg(ct);
            if (is_dih) {
                if (v != v) v = (ct < 0.0f) ? 180.0f : 0.0f;   // isnan(acos) guard, src/torsion_angle.cpp:77-84
                const v3 w = vcross(u2, d2);
                if ((u1.x * w.x) + (u1.y * w.y) + (u1.z * w.z) < 0.0f) v = -1.0f * v;
            }
            if (is_bb) (&L.out_bb[0][0])[dest] = v;
            else L.out_sc[dest] = (uint8_t)quant_trunc(v, sc_min, sc_disc);   // src/foldcomp.cpp:532-538
        }
        // ---- flush the tile's results: 6 coalesced float stores + the side-chain bytes as (unaligned) dwords.
        //      No global store inside the item loop keeps the wave's VMEM queue short, so the wait for the
        //      prefetched next tile never has to drain stores.
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < W) {
#pragma unroll
            for (int q = 0; q < 6; q++) a_arr[(size_t)q * R + base + lane] = L.out_bb[q][lane];
        }
        {
            uint8_t* dstp = rec + RL.o_sc + sc_base;
            const uint32_t nd = tile_sc >> 2;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const uint32_t d = u * WAVE + lane;
                if (d < nd) st_u32(dstp + 4 * d, *reinterpret_cast<const uint32_t*>(&L.out_sc[4 * d]));
            }
            const uint32_t tail = tile_sc & 3u;
            if ((uint32_t)lane < tail) dstp[4 * nd + lane] = L.out_sc[4 * nd + lane];
        }
        sc_base += tile_sc;
        last_cnt = cnt;
        base = base_next;
        cur = nxt;
        PH_MARK(5)
        __builtin_amdgcn_wave_barrier();
        PH_MARK(6)
        if (!have_next) break;
    }
    __threadfence_block();

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
    auto finish_q = [&](int q, float lo, float hi, const float* src, uint32_t cntq) {
        if (__builtin_expect(lo == 0.0f || hi == 0.0f, 0)) { const lo_hi e = first_extrema(src, cntq, lane); lo = e.lo; hi = e.hi; }
        qmin[q] = lo; qdisc[q] = nbins[q] / (hi - lo); qcont[q] = (hi - lo) / nbins[q];
    };
    constexpr int U = 8;
    if (n <= (uint32_t)(U * WAVE)) {
        // everything of the chain in registers: one batch of loads, no reload for the quantisation pass
        float va[7][U];
        uint32_t rcs[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
#pragma unroll
            for (int q = 0; q < 6; q++) va[q][u] = (k < m) ? a_arr[(size_t)q * R + k] : 0.0f;
            va[6][u] = (k < n) ? in.bfac_ca[r0 + k] : 0.0f;
            rcs[u] = (k < n) ? in.res_code[r0 + k] : 0u;
        }
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
            finish_q(q, wave_min_f32(lo), wave_max_f32(hi), (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0), cntq);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            if (k < n) pack_store(k, rcs[u], va[0][u], va[1][u], va[2][u], va[3][u], va[4][u], va[5][u], va[6][u]);
        }
    } else {
        for (int q = 0; q < 7; q++) {
            float lo = kInf, hi = -kInf;
            const float* src = (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0);
            const uint32_t cntq = (q < 6) ? m : n;
            for (uint32_t k0 = 0; k0 < cntq; k0 += U * WAVE) {
                float t[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const uint32_t k = k0 + u * WAVE + lane; t[u] = (k < cntq) ? src[k] : 0.0f; }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const bool on = k0 + u * WAVE + lane < cntq;
                    lo = __builtin_fminf(lo, on ? t[u] : kInf); hi = __builtin_fmaxf(hi, on ? t[u] : -kInf);
                }
            }
            float lo_w = wave_min_f32(lo), hi_w = wave_max_f32(hi);
            if (__builtin_expect(lo_w == 0.0f || hi_w == 0.0f, 0)) { const lo_hi e = first_extrema(src, cntq, lane); lo_w = e.lo; hi_w = e.hi; }
            qmin[q] = lo_w; qdisc[q] = nbins[q] / (hi_w - lo_w); qcont[q] = (hi_w - lo_w) / nbins[q];
        }
        for (uint32_t k = lane; k < n; k += WAVE) {
            const bool w = k < m;
            pack_store(k, in.res_code[r0 + k], w ? a_arr[k] : 0.f, w ? a_arr[R + k] : 0.f, w ? a_arr[2 * R + k] : 0.f,
                       w ? a_arr[3 * R + k] : 0.f, w ? a_arr[4 * R + k] : 0.f, w ? a_arr[5 * R + k] : 0.f, in.bfac_ca[r0 + k]);
        }
    }

    PH_MARK(7)
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
#pragma unroll
        for (int q = 0; q < 6; q++) { st_f32(h + 24 + 4 * q, qmin[q]); st_f32(h + 48 + 4 * q, qcont[q]); }
        // OXT (src/foldcomp.cpp:474-482): the last atom of the span, still parked in LDS from the final tile
        const float4 la = L.atom[last_cnt - 1];
        const bool has_oxt = __float_as_uint(la.w) == FCZ_ATOM_OXT;
        uint8_t* o = rec + RL.o_oxt;
        o[0] = has_oxt ? 1 : 0;
        st_f32(o + 1, has_oxt ? la.x : 0.0f);
        st_f32(o + 5, has_oxt ? la.y : 0.0f);
        st_f32(o + 9, has_oxt ? la.z : 0.0f);
        st_f32(rec + RL.o_tmp, qmin[6]);
        st_f32(rec + RL.o_tmp + 4, qcont[6]);
        if (status) status[c] = FCZ_OK;
    }
    PH_MARK(8)
    PH_FLUSH
}

}  // namespace fcz
