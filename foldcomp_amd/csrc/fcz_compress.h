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

constexpr int CT_CAP = 768;          // staged atoms per tile (typical tile: 65 residues * 8.4 atoms = 545)
constexpr int CT_RES = 65;           // 64 residues + 1 look-ahead
constexpr uint32_t CT_NONE = 0xffffu;

struct compress_tile_lds {
    float x[CT_CAP], y[CT_CAP], z[CT_CAP];
    uint16_t idx[FCZ_MAX_RES_ATOMS][CT_RES + 1];   // [slot][residue in tile] -> staged atom index
    uint16_t aoff[CT_RES + 1];                     // tile-local atom offset of each residue
    uint16_t scpre[CT_RES + 1];                    // tile-local exclusive prefix of side-chain torsion counts
    uint8_t code[CT_CAP];
    uint8_t rc[CT_RES + 3];
    uint8_t item_res[64 * 11];                     // side-chain item -> residue in tile
};

__device__ __forceinline__ v3 tile_atom(const compress_tile_lds& L, uint32_t res, uint32_t slot) {
    const uint32_t i = L.idx[slot][res];
    if (i == CT_NONE) return v3{0.0f, 0.0f, 0.0f};
    return v3{L.x[i], L.y[i], L.z[i]};
}

__global__ __launch_bounds__(BLOCK, 3) void k_compress_tiled(fcz_chain_batch in, const uint64_t* __restrict__ out_off,
                                                          uint8_t* __restrict__ out, int32_t* __restrict__ status,
                                                          float* __restrict__ ang, int keep_first_angle) {
    __shared__ compress_tile_lds s_tile[WAVES_PER_BLOCK];
    __shared__ uint8_t s_slot_of[FCZ_N_RES_CODES][40];  // atom code -> canonical slot, 255 = not in residue

    for (int i = threadIdx.x; i < FCZ_N_RES_CODES * 40; i += BLOCK) (&s_slot_of[0][0])[i] = 255;
    __syncthreads();
    for (int i = threadIdx.x; i < FCZ_N_RES_CODES * FCZ_MAX_RES_ATOMS; i += BLOCK) {
        int rc = i / FCZ_MAX_RES_ATOMS, j = i % FCZ_MAX_RES_ATOMS;
        if (j < fcz_res_natoms[rc]) s_slot_of[rc][fcz_res_atom[rc][j]] = (uint8_t)j;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= in.n_chains) return;
    compress_tile_lds& L = s_tile[wave];

    const uint32_t r0 = in.res_off[c], n = in.res_off[c + 1] - r0;
    const uint32_t title_len = in.title_off[c + 1] - in.title_off[c];
    const uint32_t thr = (uint32_t)in.anchor_threshold;
    uint8_t* rec = out + out_off[c];
    const uint32_t rec_size = (uint32_t)(out_off[c + 1] - out_off[c]);

    // ---- validation (the reference aborts on these inputs) + total side-chain torsion count ----
    int bad = (n < 2) ? FCZ_E_TOO_SHORT : (thr < 2 ? FCZ_E_INVALID_ARG : 0);
    uint32_t nsc = 0;
    for (uint32_t k = lane; k < n; k += WAVE) {
        const uint32_t rc = in.res_code[r0 + k];
        if (!res_code_ok(rc)) bad = bad ? bad : FCZ_E_RESIDUE;
        nsc += fcz_res_natoms[rc < 24 ? rc : 23] - 3;
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { int o = __shfl_xor(bad, d, WAVE); bad = o < bad ? o : bad; }
    if (bad) {
        for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
        if (lane == 0 && status) status[c] = bad;
        return;
    }
    nsc = wave_sum(nsc);

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
        // ---- tile extent: T residues (+1 look-ahead) whose atoms fit the staging buffer ----
        uint32_t T = (n - base < (uint32_t)WAVE) ? (n - base) : (uint32_t)WAVE;
        const uint32_t A0 = in.atom_off[r0 + base];
        bool look = base + T < n;
        uint32_t A1 = in.atom_off[r0 + base + T + (look ? 1u : 0u)];
        if (A1 - A0 > (uint32_t)CT_CAP) {
            // unusually atom-rich residues (e.g. explicit hydrogens): shrink the tile. Lane l holds the end of
            // residue l; residues [0,t) plus look-ahead residue t fit iff lanes 0..t all fit.
            const uint32_t my_end = (base + lane + 1 <= n) ? in.atom_off[r0 + base + lane + 1] : 0xffffffffu;
            const bool fits = (my_end != 0xffffffffu) && (my_end - A0 <= (uint32_t)CT_CAP);
            const unsigned long long fm = __ballot(fits);
            const uint32_t lead = (fm == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~fm);
            if (lead < 2) {  // one residue plus its successor exceed the staging capacity: not a protein chain
                for (uint32_t i = lane; i < rec_size; i += WAVE) rec[i] = 0;
                if (lane == 0 && status) status[c] = FCZ_E_INVALID_ARG;
                return;
            }
            T = lead - 1; look = true;
            A1 = in.atom_off[r0 + base + T + 1];
        }
        const uint32_t nres_t = T + (look ? 1u : 0u);
        const uint32_t cnt = A1 - A0;

        // ---- per-residue metadata ----
        for (uint32_t rr = lane; rr <= nres_t; rr += WAVE) L.aoff[rr] = (uint16_t)(in.atom_off[r0 + base + rr] - A0);
        for (uint32_t rr = lane; rr < nres_t; rr += WAVE) L.rc[rr] = in.res_code[r0 + base + rr];
        // ---- coalesced staging of the tile's atoms ----
        for (uint32_t i = lane; i < cnt; i += WAVE) {
            L.x[i] = in.x[A0 + i]; L.y[i] = in.y[A0 + i]; L.z[i] = in.z[A0 + i]; L.code[i] = in.atom_code[A0 + i];
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
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
        if ((uint32_t)lane < T) my_cnt = fcz_res_natoms[L.rc[lane]] - 3;
        uint32_t tile_sc;
        const uint32_t my_pre = wave_excl_scan(my_cnt, lane, &tile_sc);
        if ((uint32_t)lane < T) {
            L.scpre[lane] = (uint16_t)my_pre;
            for (uint32_t j = 0; j < my_cnt; j++) L.item_res[my_pre + j] = (uint8_t)lane;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();

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
                    P[p] = (g < 3) ? tile_atom(L, res, g) : tile_atom(L, res + 1, g - 3);
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
                    P[p] = (g < 3) ? tile_atom(L, res, g) : tile_atom(L, res + 1, g - 3);
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
                const uint32_t pk = fcz_res_prev[rc][j];
                const float v = dihedral_deg(tile_atom(L, res, pk & 15), tile_atom(L, res, (pk >> 4) & 15),
                                             tile_atom(L, res, (pk >> 8) & 15), tile_atom(L, res, j));
                rec[RL.o_sc + sc_base + ts] = (uint8_t)quant_trunc(v, sc_min, sc_disc);
            }
        }
        sc_base += tile_sc;
        base += T;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    __threadfence_block();

    // ---- per-chain quantiser parameters: Discretizer::Discretizer src/discretizer.cpp:22-33 ----
    const float kInf = __builtin_huge_valf();
    float qmin[7], qdisc[7], qcont[7];
    const float nbins[7] = {4095.0f, 4095.0f, 2047.0f, 255.0f, 255.0f, 255.0f, 255.0f};
#pragma unroll
    for (int q = 0; q < 7; q++) {
        ext mn{kInf, 0xffffffffu}, mx{-kInf, 0xffffffffu};
        const float* src = (q < 6) ? (a_arr + (size_t)q * R) : (in.bfac_ca + r0);
        const uint32_t cntq = (q < 6) ? m : n;
        for (uint32_t k = lane; k < cntq; k += WAVE) {
            const float v = src[k];
            ext_min_upd(mn, v, k); ext_max_upd(mx, v, k);
        }
        const float lo = wave_ext_min(mn), hi = wave_ext_max(mx);
        qmin[q] = lo;
        qdisc[q] = nbins[q] / (hi - lo);
        qcont[q] = (hi - lo) / nbins[q];
    }

    // ---- packed words (src/foldcomp.cpp:582-601, convertBackboneChainToBytes :33-52) + B-factors ----
    for (uint32_t k = lane; k < n; k += WAVE) {
        uint32_t res = in.res_code[r0 + k], om = 0, ps = 0, ph = 0, b1 = 0, b2 = 0, b3 = 0;
        if (k < m) {
            ph = quant_round(a_arr[0 * R + k], qmin[0], qdisc[0]) & 0xfffu;
            ps = quant_round(a_arr[1 * R + k], qmin[1], qdisc[1]) & 0xfffu;
            om = quant_round(a_arr[2 * R + k], qmin[2], qdisc[2]) & 0x7ffu;
            b3 = quant_round(a_arr[3 * R + k], qmin[3], qdisc[3]) & 0xffu;
            b1 = quant_round(a_arr[4 * R + k], qmin[4], qdisc[4]) & 0xffu;
            b2 = quant_round(a_arr[5 * R + k], qmin[5], qdisc[5]) & 0xffu;
        }
        uint8_t* w = rec + RL.o_words + 8 * k;
        w[0] = (uint8_t)(((res & 0x1fu) << 3) | (om >> 8));
        w[1] = (uint8_t)(om & 0xffu);
        w[2] = (uint8_t)(ps >> 4);
        w[3] = (uint8_t)(((ps & 0xfu) << 4) | (ph >> 8));
        w[4] = (uint8_t)(ph & 0xffu);
        w[5] = (uint8_t)b1; w[6] = (uint8_t)b2; w[7] = (uint8_t)b3;
        rec[RL.o_tbytes + k] = (uint8_t)quant_round(in.bfac_ca[r0 + k], qmin[6], qdisc[6]);
    }

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
}

}  // namespace fcz
