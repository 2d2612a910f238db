// fcz_compress.h -- k_compress_tiled: the compress kernel, second generation.
//
// One wavefront per chain. The chain is walked in tiles of up to 64 residues; per tile
//   1. the tile's atoms (a contiguous range of the SoA arrays, plus the next residue for the backbone
//      windows that straddle the tile edge) are loaded with coalesced, lane-strided loads into LDS;
//   2. every residue lane scans its atoms *in LDS* and records, per canonical slot, the index of the
//      first atom with that name (findFirstAtomCoords semantics, reference src/sidechain.cpp:140-147;
//      missing atoms read as (0,0,0));
//   3. all angle evaluations of the tile form ONE flat work list -- 3 backbone dihedrals + 3 backbone
//      bond angles per residue window, then one dihedral per side-chain atom -- and lanes take items
//      round-robin, so every iteration has 64 busy lanes running the same code (the per-residue loop of
//      the first-generation kernel idled half the lanes on the ragged side-chain counts). Side-chain
//      items are numbered exactly like the FCZ side-chain byte stream, so their byte stores are
//      consecutive across lanes.
// Backbone angles go to per-chain scratch (coalesced per angle type); after the tile loop the wave reduces
// min/max per type (value, index) and quantises + packs the 8-byte words.
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
constexpr int CT_CAP = 768;          // staged atoms per tile (typical tile: 65 residues * 8.4 atoms = 545)
constexpr int CT_RES = 65;           // 64 residues + 1 look-ahead
constexpr uint32_t CT_NONE = 0xffffu;

struct alignas(16) compress_tile_lds {
    float x[CT_CAP], y[CT_CAP], z[CT_CAP];
    uint16_t idx[FCZ_MAX_RES_ATOMS][CT_RES + 1];   // [slot][residue in tile] -> staged atom index
    uint16_t aoff[CT_RES + 1];                     // tile-local atom offset of each residue
    uint16_t scpre[CT_RES + 1];                    // tile-local exclusive prefix of side-chain torsion counts
    alignas(16) uint8_t code[CT_CAP];
    uint8_t rc[CT_RES + 3];
    uint8_t item_res[64 * 11];                     // side-chain item -> residue in tile
};

__device__ __forceinline__ v3 tile_atom(const compress_tile_lds& L, uint32_t res, uint32_t slot) {
    const uint32_t i = L.idx[slot][res];
    if (i == CT_NONE) return v3{0.0f, 0.0f, 0.0f};
    return v3{L.x[i], L.y[i], L.z[i]};
}

__global__ __launch_bounds__(BLOCK, FCZ_COMPRESS_MIN_WAVES) void k_compress_tiled(fcz_chain_batch in, const uint64_t* __restrict__ out_off,
                                                          uint8_t* __restrict__ out, int32_t* __restrict__ status,
                                                          float* __restrict__ ang, int keep_first_angle) {
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

    uint32_t sc_base = 0;       // side-chain bytes emitted so far
    uint32_t base = 0;
    while (base < n) {
        // ---- tile metadata, one round trip: atom offsets of residues base..base+65, residue codes ----
        uint32_t T = (n - base < (uint32_t)WAVE) ? (n - base) : (uint32_t)WAVE;
        const uint32_t o_lane = in.atom_off[r0 + ((base + lane <= n) ? base + lane : n)];            // start of residue `lane`
        const uint32_t o_hi = (lane < 2) ? in.atom_off[r0 + ((base + 64 + lane <= n) ? base + 64 + lane : n)] : 0u;  // 64, 65
        uint32_t rc_lane = (base + lane < n) ? in.res_code[r0 + base + lane] : 23u;
        const uint32_t rc_hi = (lane == 0 && base + 64 < n) ? in.res_code[r0 + base + 64] : 23u;
        const uint32_t A0 = __shfl(o_lane, 0, WAVE);
        const uint32_t o64 = __shfl(o_hi, 0, WAVE), o65 = __shfl(o_hi, 1, WAVE);
        uint32_t my_end = __shfl_down(o_lane, 1, WAVE);            // end of residue `lane`
        if (lane == 63) my_end = o64;
        bool look = base + T < n;
        uint32_t A1;
        {
            const uint32_t last = T + (look ? 1u : 0u);            // atom_off index (relative) of the tile end
            A1 = (last == 65) ? o65 : (last == 64 ? o64 : __shfl(o_lane, (int)last, WAVE));
        }
        if (A1 - A0 > (uint32_t)CT_CAP) {
            // unusually atom-rich residues (e.g. explicit hydrogens): shrink the tile. Residues [0,t) plus the
            // look-ahead residue t fit iff the ends of residues 0..t all lie within the staging capacity.
            const bool fits = (base + lane < n) && (my_end - A0 <= (uint32_t)CT_CAP);
            const unsigned long long fm = __ballot(fits);
            const uint32_t lead = (fm == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~fm);
            if (lead < 2) {  // one residue plus its successor exceed the staging capacity: not a protein chain
                for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
                if (lane == 0 && status) status[c] = FCZ_E_INVALID_ARG;
                return;
            }
            T = lead - 1; look = true;
            A1 = (T + 1 == 64) ? o64 : __shfl(o_lane, (int)(T + 1), WAVE);
        }
        const uint32_t nres_t = T + (look ? 1u : 0u);
        const uint32_t cnt = A1 - A0;
        if ((uint32_t)lane <= nres_t) L.aoff[lane] = (uint16_t)(o_lane - A0);
        if (lane == 0) { L.aoff[64] = (uint16_t)(o64 - A0); L.aoff[65] = (uint16_t)(o65 - A0); L.rc[64] = (uint8_t)rc_hi; }
        L.rc[lane] = (uint8_t)rc_lane;

        PH_MARK(1)
        // ---- coalesced staging of the tile's atoms: 16-byte loads, all issued before the first use ----
        {
            constexpr int NV = CT_CAP / (4 * WAVE);   // float4 rounds (3)
            float4 vx[NV], vy[NV], vz[NV];
            uint32_t vc[NV];
            const bool whole = (size_t)A0 + CT_CAP + 4 <= (size_t)in.n_atoms;   // 16-byte reads stay inside the arrays
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i4 = 4 * (u * WAVE + lane);
                vx[u] = vy[u] = vz[u] = float4{0.f, 0.f, 0.f, 0.f}; vc[u] = 0;
                if (i4 < cnt) {
                    if (whole || (size_t)A0 + i4 + 4 <= (size_t)in.n_atoms) {
                        __builtin_memcpy(&vx[u], in.x + A0 + i4, 16);
                        __builtin_memcpy(&vy[u], in.y + A0 + i4, 16);
                        __builtin_memcpy(&vz[u], in.z + A0 + i4, 16);
                        __builtin_memcpy(&vc[u], in.atom_code + A0 + i4, 4);
                    } else {
                        // last few atoms of the whole batch: element-wise, never past the end of the arrays
                        const uint32_t a = A0 + i4;
                        const bool h1 = i4 + 1 < cnt, h2 = i4 + 2 < cnt, h3 = i4 + 3 < cnt;
                        vx[u] = float4{in.x[a], h1 ? in.x[a + 1] : 0.f, h2 ? in.x[a + 2] : 0.f, h3 ? in.x[a + 3] : 0.f};
                        vy[u] = float4{in.y[a], h1 ? in.y[a + 1] : 0.f, h2 ? in.y[a + 2] : 0.f, h3 ? in.y[a + 3] : 0.f};
                        vz[u] = float4{in.z[a], h1 ? in.z[a + 1] : 0.f, h2 ? in.z[a + 2] : 0.f, h3 ? in.z[a + 3] : 0.f};
                        vc[u] = (uint32_t)in.atom_code[a] | (h1 ? (uint32_t)in.atom_code[a + 1] << 8 : 0u) |
                                (h2 ? (uint32_t)in.atom_code[a + 2] << 16 : 0u) | (h3 ? (uint32_t)in.atom_code[a + 3] << 24 : 0u);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t i4 = 4 * (u * WAVE + lane);
                if (i4 < cnt) {
                    *reinterpret_cast<float4*>(&L.x[i4]) = vx[u];
                    *reinterpret_cast<float4*>(&L.y[i4]) = vy[u];
                    *reinterpret_cast<float4*>(&L.z[i4]) = vz[u];
                    *reinterpret_cast<uint32_t*>(&L.code[i4]) = vc[u];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        PH_MARK(2)
        // ---- slot index table: first atom of each canonical name ----
        for (uint32_t rr = lane; rr < nres_t; rr += WAVE) {
            const uint32_t rc = L.rc[rr];
#pragma unroll
            for (int s = 0; s < FCZ_MAX_RES_ATOMS; s++) L.idx[s][rr] = (uint16_t)CT_NONE;
            uint32_t filled = 0;
            const uint32_t e = L.aoff[rr + 1];
            for (uint32_t i = L.aoff[rr]; i < e; i++) {
                const uint32_t code = L.code[i];
                const uint32_t slot = code < 40 ? s_slot_of[rc][code] : 255u;
                if (slot != 255u && !((filled >> slot) & 1u)) { filled |= 1u << slot; L.idx[slot][rr] = (uint16_t)i; }
            }
        }
        // side-chain item numbering of this tile
        uint32_t my_cnt = 0;
        if ((uint32_t)lane < T) my_cnt = s_natoms[L.rc[lane] & 31u] - 3;
        uint32_t tile_sc;
        const uint32_t my_pre = wave_excl_scan(my_cnt, lane, &tile_sc);
        if ((uint32_t)lane < T) {
            L.scpre[lane] = (uint16_t)my_pre;
            for (uint32_t j = 0; j < my_cnt; j++) L.item_res[my_pre + j] = (uint8_t)lane;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();

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
        // (ca_c_n, c_n_ca, n_ca_c), then the side-chain dihedrals. W = number of residue windows of the
        // tile = residues k with k < n-1.
        const uint32_t W = (base + T <= m) ? T : (m > base ? m - base : 0);
        const uint32_t n_items = 6 * W + tile_sc;
        for (uint32_t t0 = 0; t0 < n_items; t0 += WAVE) {
            const uint32_t t = t0 + lane;
            if (t >= n_items) continue;
            if (t < 3 * W) {
                // getTorsionFromXYZ window, reference src/torsion_angle.cpp:46-96; split src/foldcomp.cpp:488-492
                const uint32_t q = (t >= 2 * W) ? 2u : (t >= W ? 1u : 0u);
                const uint32_t res = t - q * W;
                v3 P[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t g = q + p;
                    P[p] = tile_atom(L, res + (g >= 3 ? 1u : 0u), g >= 3 ? g - 3 : g);
                }
                const float v = dihedral_deg(P[0], P[1], P[2], P[3]);
                const uint32_t arr = (q == 0) ? 1u : (q == 1 ? 2u : 0u);      // psi, omega, phi
                a_arr[arr * R + base + res] = v;
            } else if (t < 6 * W) {
                // getBondAngles, reference src/nerf.cpp:495-508; split src/foldcomp.cpp:497-505
                const uint32_t tt = t - 3 * W;
                const uint32_t q = (tt >= 2 * W) ? 2u : (tt >= W ? 1u : 0u);
                const uint32_t res = tt - q * W;
                v3 P[3];
#pragma unroll
                for (int p = 0; p < 3; p++) {
                    const uint32_t g = q + 1 + p;
                    P[p] = tile_atom(L, res + (g >= 3 ? 1u : 0u), g >= 3 ? g - 3 : g);
                }
                const float v = bond_angle_deg(P[0], P[1], P[2]);
                const uint32_t arr = (q == 0) ? 4u : (q == 1 ? 5u : 3u);      // ca_c_n, c_n_ca, n_ca_c
                a_arr[arr * R + base + res] = v;
            } else {
                // calculateTorsionAnglesInResidue, reference src/sidechain.cpp:149-168; truncating quantiser
                // src/foldcomp.cpp:532-538
                const uint32_t ts = t - 6 * W;
                const uint32_t res = L.item_res[ts];
                const uint32_t rc = L.rc[res];
                const uint32_t j = 3 + ts - L.scpre[res];
                const uint32_t pk = s_prev[rc][j];
                const float v = dihedral_deg(tile_atom(L, res, pk & 15), tile_atom(L, res, (pk >> 4) & 15),
                                             tile_atom(L, res, (pk >> 8) & 15), tile_atom(L, res, j));
                rec[RL.o_sc + sc_base + ts] = (uint8_t)quant_trunc(v, sc_min, sc_disc);
            }
        }
        sc_base += tile_sc;
        base += T;
        PH_MARK(5)
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        PH_MARK(6)
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
            ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
            const uint32_t cntq = (q < 6) ? m : n;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t k = u * WAVE + lane;
                if (k < cntq) { ext_min_upd(mn, va[q][u], k); ext_max_upd(mx, va[q][u], k); }
            }
            const float lo = wave_ext_min(mn), hi = wave_ext_max(mx);
            qmin[q] = lo; qdisc[q] = nbins[q] / (hi - lo); qcont[q] = (hi - lo) / nbins[q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            if (k < n) {
                pack_store(k, rcs[u], va[0][u], va[1][u], va[2][u], va[3][u], va[4][u], va[5][u], va[6][u]);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 7; q++) {
            ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
            const float* src = (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0);
            const uint32_t cntq = (q < 6) ? m : n;
            for (uint32_t k0 = 0; k0 < cntq; k0 += U * WAVE) {
                float t[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const uint32_t k = k0 + u * WAVE + lane; t[u] = (k < cntq) ? src[k] : 0.0f; }
#pragma unroll
                for (int u = 0; u < U; u++) { const uint32_t k = k0 + u * WAVE + lane; if (k < cntq) { ext_min_upd(mn, t[u], k); ext_max_upd(mx, t[u], k); } }
            }
            const float lo = wave_ext_min(mn), hi = wave_ext_max(mx);
            qmin[q] = lo; qdisc[q] = nbins[q] / (hi - lo); qcont[q] = (hi - lo) / nbins[q];
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
        const uint32_t a_first = in.atom_off[r0], a_end = in.atom_off[r0 + n];
        rec[0] = 'F'; rec[1] = 'C'; rec[2] = 'M'; rec[3] = 'P';
        uint8_t* h = rec + 4;
        st_u16(h + 0, n);
        st_u16(h + 2, a_end - a_first);
        st_u16(h + 4, (uint32_t)in.first_res_index[c]);
        st_u16(h + 6, (uint32_t)in.first_atom_index[c]);
        h[8] = (uint8_t)n_anchor;
        h[9] = (uint8_t)in.chain_id[c];
        h[10] = 0; h[11] = 0;  // struct padding: the reference leaves it uninitialised
        st_u32(h + 12, nsc);
        h[16] = (uint8_t)fcz_res1[in.res_code[r0]];
        h[17] = (uint8_t)fcz_res1[in.res_code[r0 + n - 1]];
        h[18] = 0; h[19] = 0;
        st_u32(h + 20, title_len);
#pragma unroll
        for (int q = 0; q < 6; q++) { st_f32(h + 24 + 4 * q, qmin[q]); st_f32(h + 48 + 4 * q, qcont[q]); }
        const bool has_oxt = in.atom_code[a_end - 1] == FCZ_ATOM_OXT;   // src/foldcomp.cpp:474-482
        uint8_t* o = rec + RL.o_oxt;
        o[0] = has_oxt ? 1 : 0;
        st_f32(o + 1, has_oxt ? in.x[a_end - 1] : 0.0f);
        st_f32(o + 5, has_oxt ? in.y[a_end - 1] : 0.0f);
        st_f32(o + 9, has_oxt ? in.z[a_end - 1] : 0.0f);
        st_f32(rec + RL.o_tmp, qmin[6]);
        st_f32(rec + RL.o_tmp + 4, qcont[6]);
        if (status) status[c] = FCZ_OK;
    }
    PH_MARK(8)
    PH_FLUSH
}

}  // namespace fcz
