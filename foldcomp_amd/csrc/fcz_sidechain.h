// fcz_sidechain.h -- side-chain reconstruction of the decompress path (Nerf::reconstructAminoAcid, reference
// src/nerf.cpp:106-155; the per-residue part of Foldcomp::decompress, src/foldcomp.cpp:860-900).
//
// Residues are independent once the blended backbone exists, so this stage ignores chain boundaries: the R
// residues of the batch form one flat array, a block takes 256 consecutive residues (whatever chains they belong
// to) and every atom that has to be placed is one work item.
//
//   k_res_index      one wavefront per chain: resolves everything that hangs off the chain's header into flat
//                    per-residue arrays (first output atom, residue code, torsion bytes) and emits the outputs
//                    that need no geometry (B-factors, residue codes, OXT). Entries beyond RI_ROWS (= 16 x
//                    FCZ_INDEX_ROWS_MAX_ROUNDS = 64) residues.
//   k_res_index_rows the same for entries of up to RI_ROWS residues, four to a wavefront (one entry per 16-lane group, 1 / 2 / 4
//                    rounds of 16 residues by length class).
//   k_sidechain      persistent blocks over 256-residue tiles.
//                    phase 0 (thread = residue): load the residue's index entry and backbone atoms (prefetched one
//                      tile ahead); emit N, CA, C; place O (every residue's first item); publish the residue
//                      in LDS; append its remaining atoms to per-depth work lists.
//                    phase 1 (thread = work item): depth d = length of the predecessor chain of an atom inside
//                      its residue (CB: 2 ... TRP CH2: 7). Lists are processed in depth order with a block barrier
//                      in between, so every predecessor is in the LDS slot store when an item runs, and every
//                      round has full wavefronts: 6.25 wave-rounds of place_atom per wavefront instead of the 11 a
//                      residue-per-lane loop needs (the longest residue of 64 is almost always a TRP/ARG/TYR).
//                    Atoms live in LDS in OUTPUT order (one buffer is both the predecessor store and the
//                      write-back staging), so the tile leaves as fully coalesced stores: scattered 4-byte stores
//                      (one L2 request per lane) cost 20 % of the kernel before.
#pragma once
#include "fcz_kernels.h"

namespace fcz {

// One wavefront per chain: everything of the record that is addressed through the chain's header is resolved
// here, so k_sidechain sees plain per-residue arrays (one level of coalesced loads, nothing chain-dependent).
//   res_aoff[r]   index of the residue's first output atom (n_res + 1 entries, the last = total atoms)
//   res_rc[r]     residue code (first residue: header.firstResidue, src/foldcomp.cpp:863)
//   res_sc[q][r]  the residue's (<= 11) side-chain torsion bytes as three dwords (bytes past the residue's own are
//                 whatever follows in the record and are never used)
// plus the per-residue outputs that need nothing else: B-factor (src/foldcomp.cpp:884-892), residue code, and the
// chain's OXT atom (:893-900).
#ifndef FCZ_INDEX_ROWS_MAX_ROUNDS
#define FCZ_INDEX_ROWS_MAX_ROUNDS 4    // (8 = entries of up to 128 residues: measured slower here, the rows kernel loses two wavefronts per SIMD)
#endif
constexpr uint32_t RI_ROWS = 16u * FCZ_INDEX_ROWS_MAX_ROUNDS;   // entries of up to this many residues are k_res_index_rows' (four to a wavefront)
__global__ __launch_bounds__(BLOCK) void k_res_index(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off,
                                                     uint32_t n_entries, const uint32_t* __restrict__ res_off,
                                                     const uint32_t* __restrict__ atom_off, uint32_t n_res,
                                                     uint32_t* __restrict__ res_aoff, uint8_t* __restrict__ res_rc,
                                                     uint32_t* __restrict__ res_sc, fcz_atoms_out out, const uint8_t* __restrict__ codes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= n_entries) return;
    const uint32_t r0 = res_off[c], n = res_off[c + 1] - r0;
    if (n <= RI_ROWS) return;   // skipped entry (n = 0), or k_res_index_rows' (several short chains to a wavefront)
    const uint8_t* e = blob + off[c];
    const entry_view v = view_entry(e);
    const uint32_t abase = atom_off[c];
    const float tmin = ld_f32(e + v.L.o_tmp), tcf = ld_f32(e + v.L.o_tmp + 4);
    // residue codes: the byte array the sizes pass left (k_entry_sizes: first residue already from the header, codes clamped) --
    // consecutive bytes instead of one byte out of every 8-byte word (a whole extra pass over the words array)
    const uint8_t* rcs = codes + (off[c] >> 3);
    const uint8_t* scb = e + v.L.o_sc;
    uint32_t run = 0;
    auto emit = [&](uint32_t k, uint32_t rc, uint32_t tq, uint32_t ex, uint32_t q0, uint32_t q1, uint32_t q2) {
        const size_t r = (size_t)r0 + k;
        res_aoff[r] = abase + ex;
        res_rc[r] = (uint8_t)rc;
        res_sc[r] = q0; res_sc[(size_t)n_res + r] = q1; res_sc[2 * (size_t)n_res + r] = q2;
        out.bfac_res[r] = dequant(tq, tmin, tcf);
        if (out.res_code) out.res_code[r] = (uint8_t)rc;
    };
    // Torsion bytes of a residue: atoms before it minus 3 per residue = torsion bytes before it. The dwords read at most 11
    // bytes past the residue's last torsion byte: still inside the record (8-byte B-factor header + n bytes follow).
    // a chain of up to 384 residues: every load of a dependency level is in flight at once (two memory round trips per chain
    // instead of two per 64 residues). The rounds are a compile-time count chosen by the chain's length (1, 2, 4 or 6 rounds of 64
    // residues; chains of up to 64 are k_res_index_rows'): a 100-residue chain does not pay the loads, scans and stores of four empty
    // rounds.
    auto rounds = [&](auto U_) {
        constexpr int U = decltype(U_)::value;
        uint32_t wb[U], tq[U], rc[U], na[U], ex[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane, kc = k < n ? k : n - 1;
            wb[u] = rcs[kc];
            tq[u] = e[v.L.o_tbytes + kc];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            uint32_t r = wb[u];
            if (r >= 24) r = 23;
            rc[u] = r;
            na[u] = k < n ? (uint32_t)fcz_res_natoms[r] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t tot;
            ex[u] = run + wave_excl_scan(na[u], lane, &tot);
            run += tot;
        }
        uint32_t q0[U], q1[U], q2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            const bool act = k < n;
            const uint8_t* sp = scb + (act ? ex[u] - 3 * k : 0u);
            q0[u] = ld_u32(sp); q1[u] = ld_u32(sp + 4);
            q2[u] = ld_u32((act && na[u] > 11) ? sp + 8 : sp);   // only TRP / TYR own a third dword
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t k = u * WAVE + lane;
            if (k < n) emit(k, rc[u], tq[u], ex[u], q0[u], q1[u], na[u] > 11 ? q2[u] : 0u);
        }
    };
    if (n <= 2u * WAVE) rounds(std::integral_constant<int, 2>{});
    else if (n <= 4u * WAVE) rounds(std::integral_constant<int, 4>{});
    else if (n <= 6u * WAVE) rounds(std::integral_constant<int, 6>{});
    else {
        for (uint32_t base = 0; base < n; base += WAVE) {
            const uint32_t k = base + lane;
            const bool act = k < n;
            uint32_t na = 0, rc = 23, tq = 0;
            if (act) {
                rc = rcs[k];
                tq = e[v.L.o_tbytes + k];
                if (rc >= 24) rc = 23;
                na = fcz_res_natoms[rc];
            }
            uint32_t tot;
            const uint32_t ex = run + wave_excl_scan(na, lane, &tot);
            run += tot;
            if (!act) continue;
            const uint8_t* sp = scb + (ex - 3 * k);
            emit(k, rc, tq, ex, ld_u32(sp), ld_u32(sp + 4), (na > 11) ? ld_u32(sp + 8) : 0u);
        }
    }
    if (lane == 0) {
        const bool oxt = e[v.L.o_oxt] != 0;
        if (oxt) {
            const uint32_t a = abase + run;
            const v3 o = ld_v3(e + v.L.o_oxt + 1);
            out.x[a] = o.x; out.y[a] = o.y; out.z[a] = o.z;
            if (out.atom_code) out.atom_code[a] = FCZ_ATOM_OXT;
        }
        if (r0 + n == n_res) res_aoff[n_res] = abase + run + (oxt ? 1u : 0u);   // closes the array: total atoms
    }
}

// Entries of 1 .. 64 residues FOUR to a wavefront: one entry per 16-lane group, in 1, 2 or 4 rounds of 16 residues by length class
// (lane `sub` of a group holds residues sub, sub + 16, ...); a persistent grid over chunks of 16 consecutive entries. What k_res_index
// does per wavefront -- header, layout, the B-factor parameters, the OXT -- is done once per group here; a 16-residue chain filled a
// quarter of the lanes before (1.3 ms per 2 M of them).
template <int G, int U>
__device__ __forceinline__ void res_index_rows(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, const uint32_t c, const bool live,
                                               const uint32_t r0, const uint32_t n, const uint32_t* __restrict__ atom_off, uint32_t n_res,
                                               uint32_t* __restrict__ res_aoff, uint8_t* __restrict__ res_rc, uint32_t* __restrict__ res_sc,
                                               const fcz_atoms_out& out, const uint8_t* __restrict__ codes) {
    const uint32_t sub = (uint32_t)(threadIdx.x & (G - 1));
    // (a group without an entry runs along on a live group's entry -- same addresses, nothing written)
    const uint8_t* e = blob + off[c];
    const entry_view v = view_entry(e);
    const uint32_t abase = atom_off[c];
    const float tmin = ld_f32(e + v.L.o_tmp), tcf = ld_f32(e + v.L.o_tmp + 4);
    const uint8_t* rcs = codes + (off[c] >> 3);
    const uint8_t* scb = e + v.L.o_sc;
    uint32_t rc[U], tq[U], na[U], ex[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub, kc = k < n ? k : n - 1;
        rc[u] = rcs[kc];
        tq[u] = e[v.L.o_tbytes + kc];
    }
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        if (rc[u] >= 24) rc[u] = 23;
        na[u] = k < n ? (uint32_t)fcz_res_natoms[rc[u]] : 0u;
        uint32_t inc = na[u];
#pragma unroll
        for (int d = 1; d < G; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, d, G); if (sub >= (uint32_t)d) inc += t; }
        ex[u] = run + inc - na[u];
        run += (uint32_t)__shfl((int)inc, G - 1, G);
    }
    uint32_t q0[U], q1[U], q2[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        const bool act = k < n;
        const uint8_t* sp = scb + (act ? ex[u] - 3 * k : 0u);
        q0[u] = ld_u32(sp); q1[u] = ld_u32(sp + 4); q2[u] = ld_u32((act && na[u] > 11) ? sp + 8 : sp);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t k = (uint32_t)u * G + sub;
        if (live && k < n) {
            const size_t r = (size_t)r0 + k;
            res_aoff[r] = abase + ex[u];
            res_rc[r] = (uint8_t)rc[u];
            res_sc[r] = q0[u]; res_sc[(size_t)n_res + r] = q1[u]; res_sc[2 * (size_t)n_res + r] = na[u] > 11 ? q2[u] : 0u;
            out.bfac_res[r] = dequant(tq[u], tmin, tcf);
            if (out.res_code) out.res_code[r] = (uint8_t)rc[u];
        }
    }
    if (live && sub == 0) {
        const bool oxt = e[v.L.o_oxt] != 0;
        if (oxt) {
            const uint32_t a = abase + run;
            const v3 o = ld_v3(e + v.L.o_oxt + 1);
            out.x[a] = o.x; out.y[a] = o.y; out.z[a] = o.z;
            if (out.atom_code) out.atom_code[a] = FCZ_ATOM_OXT;
        }
        if (r0 + n == n_res) res_aoff[n_res] = abase + run + (oxt ? 1u : 0u);
    }
}

constexpr int RI_CHUNK = 16;
template <int U>
__device__ __forceinline__ void res_index_class(unsigned long long todo, const uint32_t c0, const uint32_t ro, const uint32_t nn,
                                                const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, const uint32_t* __restrict__ atom_off,
                                                uint32_t n_res, uint32_t* __restrict__ res_aoff, uint8_t* __restrict__ res_rc, uint32_t* __restrict__ res_sc,
                                                const fcz_atoms_out& out, const uint8_t* __restrict__ codes) {
    const uint32_t g16 = (uint32_t)(threadIdx.x & 63) >> 4;
    while (todo) {
        const int l0 = __builtin_ctzll(todo);
        uint32_t cc = c0 + (uint32_t)l0, rr0 = (uint32_t)__builtin_amdgcn_readlane((int)ro, l0), rn = (uint32_t)__builtin_amdgcn_readlane((int)nn, l0);
        bool live = g16 == 0;
        todo &= todo - 1;
#pragma unroll
        for (uint32_t g = 1; g < 4; g++) {
            if (!todo) break;
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t r0g = (uint32_t)__builtin_amdgcn_readlane((int)ro, l), ng = (uint32_t)__builtin_amdgcn_readlane((int)nn, l);
            if (g16 == g) { cc = c0 + (uint32_t)l; rr0 = r0g; rn = ng; live = true; }
        }
        res_index_rows<16, U>(blob, off, cc, live, rr0, rn, atom_off, n_res, res_aoff, res_rc, res_sc, out, codes);
    }
}
__global__ __launch_bounds__(BLOCK) void k_res_index_rows(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off,
                                                          uint32_t n_entries, const uint32_t* __restrict__ res_off,
                                                          const uint32_t* __restrict__ atom_off, uint32_t n_res,
                                                          uint32_t* __restrict__ res_aoff, uint8_t* __restrict__ res_rc,
                                                          uint32_t* __restrict__ res_sc, fcz_atoms_out out, const uint8_t* __restrict__ codes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    const uint32_t n_chunks = (n_entries + RI_CHUNK - 1) / RI_CHUNK;
    for (uint32_t ch = blockIdx.x * WAVES_PER_BLOCK + wave; ch < n_chunks; ch += n_waves) {
        const uint32_t c0 = ch * RI_CHUNK;
        const uint32_t ci = c0 + (uint32_t)lane;
        const uint32_t ro = res_off[ci <= n_entries ? ci : n_entries];
        const uint32_t nn = (uint32_t)__shfl_down((int)ro, 1, WAVE) - ro;
        const bool mine = lane < RI_CHUNK && ci < n_entries;
        const unsigned long long t1 = __ballot(mine && nn >= 1u && nn <= 16u), t2 = __ballot(mine && nn > 16u && nn <= 32u),
                                 t4 = __ballot(mine && nn > 32u && nn <= 64u);
        if (t1) res_index_class<1>(t1, c0, ro, nn, blob, off, atom_off, n_res, res_aoff, res_rc, res_sc, out, codes);
        if (t2) res_index_class<2>(t2, c0, ro, nn, blob, off, atom_off, n_res, res_aoff, res_rc, res_sc, out, codes);
        if (t4) res_index_class<4>(t4, c0, ro, nn, blob, off, atom_off, n_res, res_aoff, res_rc, res_sc, out, codes);
#if FCZ_INDEX_ROWS_MAX_ROUNDS >= 8
        const unsigned long long t8 = __ballot(mine && nn > 64u && nn <= 128u);
        if (t8) res_index_class<8>(t8, c0, ro, nn, blob, off, atom_off, n_res, res_aoff, res_rc, res_sc, out, codes);
#endif
    }
}

constexpr int SC_TILE = BLOCK;             // residues per tile
constexpr int SC_CAP = 2304;               // staged output atoms per tile (a typical tile holds 256 * 8.36 = 2140 +- 43; what
                                           // does not fit goes to the 128-residue launch). With the other sizes below the block
                                           // needs < 40 KB of LDS: four blocks per CU
constexpr int SC_LIST = 1408;              // work items per tile: 10 per TRP, and 14 x + 3 (256 - x) <= SC_CAP bounds x at 139
static_assert(11 * SC_LIST >= 10 * (SC_CAP - 3 * SC_TILE), "item list too short for the atom-richest tile that still fits SC_CAP");
constexpr int SC_DEPTHS = 6;               // list depths 2..7 (depth 1 = O, placed by the owning thread)
constexpr int SC_FIELD = 10;               // bits per depth in the packed per-depth counters (<= 3 * 256 items)
constexpr int SC_MAX_ITEMS = 10;           // listed atoms per residue (TRP: 14 - 3 backbone - O)
constexpr int SC_GEOM_CODES = 20;          // residue codes that own side-chain geometry (the 20 standard residues)

// ideal geometry of atom `slot` of a residue type, with every trig value precomputed, plus where its predecessors
// and the atom itself sit in the residue's output order
struct alignas(16) sc_geom {
    float d2x;        // -1 * L * cos(bond angle)
    float blen;       // L
    float sb;         // sin(bond angle)
    uint32_t meta;    // out pos of prev0 | prev1 << 4 | prev2 << 8 | own << 12 | atom code << 16
};

struct sidechain_lds {
    float stage[3][SC_CAP];                            // x, y, z of the pass's atoms in OUTPUT order: predecessor store
                                                       // and write-back buffer in one
    uint32_t sc[3][SC_TILE];                           // torsion bytes of each residue (three dwords)
    uint16_t apos[SC_TILE];                            // pass-local position of each residue's first atom
    uint16_t list[SC_LIST];                            // work items: residue in tile | slot << 8, grouped by depth
    uint8_t rc[SC_TILE];
    unsigned long long wave_tot[WAVES_PER_BLOCK];
    uint32_t dstart[SC_DEPTHS + 2];                    // list range of each depth in the current pass
    // tables (the residue code differs per lane, so these are LDS lookups rather than scalar loads)
    float tor_cos[256], tor_sin[256];                  // every sinf/cosf of a torsion byte
    sc_geom geom[SC_GEOM_CODES][FCZ_MAX_RES_ATOMS - 3];   // [code][slot - 3]: O and the side-chain atoms
    uint8_t opos[FCZ_N_RES_CODES][4];                  // output position of N, CA, C, O
    uint8_t items[FCZ_N_RES_CODES][SC_DEPTHS * 4];     // [depth-2][q]: slots (>= 4) of the residue's atoms at that depth, 0 = none
    uint8_t natoms[FCZ_N_RES_CODES];
    unsigned long long dcnt[FCZ_N_RES_CODES];          // items per depth, SC_FIELD bits each
};

#ifndef FCZ_SIDECHAIN_MIN_BLOCKS
#define FCZ_SIDECHAIN_MIN_BLOCKS 4
#endif

#ifdef FCZ_SC_TIMING
// measurement aid (not built into the product): wavefront-cycles between the phase boundaries of k_sidechain
__device__ unsigned long long g_sc_timing[12];
#define SC_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define SC_STAMP(i)
#endif
// res_aoff has n_res + 1 entries (the last one = total atoms). FAST: placements in plain float arithmetic
// (FCZ_NUMERICS_FAST, place_atom_d2_fast); the tables (torsion bytes, ideal geometry) are the exact ones either way.
// tile_res = residues per tile: 256 (tile_list == nullptr: every tile of the batch), or 128 for the tiles a 256-residue launch
// could not stage (more than SC_CAP atoms: only when nearly every residue is a TRP/ARG/TYR) and put on punt_list as two halves.
template <bool FAST>
__global__ __launch_bounds__(BLOCK, FCZ_SIDECHAIN_MIN_BLOCKS)
void k_sidechain(uint32_t n_res, uint32_t n_tiles, uint32_t tile_res, const uint32_t* __restrict__ tile_list, const uint32_t* __restrict__ tile_count,
                 uint32_t* __restrict__ punt_list, uint32_t* __restrict__ punt_count,
                 const uint32_t* __restrict__ res_aoff, const uint8_t* __restrict__ res_rc,
                 const uint32_t* __restrict__ res_sc, const v3* __restrict__ bb, int alt_order, fcz_atoms_out out) {
    __shared__ sidechain_lds S;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // ---- tables, once per (persistent) block ----
    {
        const float cont = (180.0f - (-180.0f)) / 255.0f;   // FixedAngleDiscretizer(255), src/discretizer.h:89-106
        const float ta = deg2rad(dequant((uint32_t)t, -180.0f, cont));
        S.tor_cos[t] = cosf_glibc(ta);
        S.tor_sin[t] = sinf_glibc(ta);
        if (t < FCZ_N_RES_CODES) {
            const int rc = t, na = fcz_res_natoms[rc];
            S.natoms[rc] = (uint8_t)na;
            // canonical slot -> output position (the `-a` order permutes atoms inside a residue)
            uint32_t opos[FCZ_MAX_RES_ATOMS];
            for (int j = 0; j < FCZ_MAX_RES_ATOMS; j++) opos[j] = (uint32_t)j;
            if (alt_order) for (int j = 0; j < na; j++) opos[fcz_res_alt_slot[rc][j]] = (uint32_t)j;
            for (int j = 0; j < 4; j++) S.opos[rc][j] = (uint8_t)opos[j];
            int depth[FCZ_MAX_RES_ATOMS];
            for (int j = 0; j < FCZ_MAX_RES_ATOMS; j++) depth[j] = 0;
            for (int j = 3; j < na; j++) {
                const uint32_t pk = fcz_res_prev[rc][j];
                int d = depth[pk & 15]; const int d1 = depth[(pk >> 4) & 15], d2 = depth[(pk >> 8) & 15];
                d = d1 > d ? d1 : d; d = d2 > d ? d2 : d;
                depth[j] = d + 1;
                if (rc < SC_GEOM_CODES) {
                    const float L = __uint_as_float(fcz_res_blen_bits[rc][j]);
                    const float ba = deg2rad(__uint_as_float(fcz_res_bang_bits[rc][j]));
                    sc_geom g;
                    g.d2x = -1.0f * L * cosf_glibc(ba);
                    g.blen = L;
                    g.sb = sinf_glibc(ba);
                    g.meta = opos[pk & 15] | (opos[(pk >> 4) & 15] << 4) | (opos[(pk >> 8) & 15] << 8) | (opos[j] << 12) |
                             ((uint32_t)fcz_res_atom[rc][j] << 16);
                    S.geom[rc][j - 3] = g;
                }
            }
            unsigned long long cnt = 0;
            for (int d = 2; d < 2 + SC_DEPTHS; d++) {
                int w = 0;
                for (int q = 0; q < 4; q++) S.items[rc][4 * (d - 2) + q] = 0;
                for (int j = 4; j < na; j++)
                    if (depth[j] == d) { S.items[rc][4 * (d - 2) + w++] = (uint8_t)j; cnt += 1ull << (SC_FIELD * (d - 2)); }
            }
            S.dcnt[rc] = cnt;
        }
    }
    __syncthreads();

    // one tile of per-residue inputs, loaded a whole tile ahead of its use. Unconditional loads from clamped
    // indices: a conditional merge would make the compiler wait for the data right here instead of a tile later.
    struct res_in { uint32_t a, a_next, rc, q0, q1, q2, a_first, a_end; v3 b0, b1, b2; };
    const uint32_t n_units = tile_list ? *tile_count : n_tiles;
    auto unit_tile = [&](uint32_t u) -> uint32_t { return tile_list ? tile_list[u < n_units ? u : 0u] : u; };
    // buffer loads: the base of every array at the tile's first residue lives in SGPRs, the lane's offset is 4 t (12 t for bb)
    // and the range check returns zero past the end of the batch: no per-load 64-bit address arithmetic, no clamping
    auto load_res = [&](uint32_t tile) -> res_in {
        const size_t r_lo = (size_t)tile * tile_res;
        const size_t rf = r_lo < (size_t)n_res ? r_lo : (size_t)n_res;
        const size_t re = r_lo + tile_res < (size_t)n_res ? r_lo + tile_res : (size_t)n_res;
        const uint32_t left = (uint32_t)((size_t)n_res - rf);                     // residues from the tile's first one to the end
        const uint32_t rows = left < tile_res ? left : tile_res;                  // residues of this tile
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(res_aoff) + rf, 0, (int)((rows + 1u) * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(res_rc) + rf, 0, (int)rows, 0x00020000);
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(res_sc) + rf, 0, (int)(rows * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(res_sc) + (size_t)n_res + rf, 0, (int)(rows * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(res_sc) + 2 * (size_t)n_res + rf, 0, (int)(rows * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<v3*>(bb) + 3 * rf, 0, (int)(rows * 36u), 0x00020000);
        res_in in;
        in.a = __builtin_amdgcn_raw_buffer_load_b32(ra, 4 * t, 0, 0);
        in.a_next = __builtin_amdgcn_raw_buffer_load_b32(ra, 4 * t + 4, 0, 0);
        in.rc = __builtin_amdgcn_raw_buffer_load_b8(rr, t, 0, 0);
        in.q0 = __builtin_amdgcn_raw_buffer_load_b32(r0, 4 * t, 0, 0);
        in.q1 = __builtin_amdgcn_raw_buffer_load_b32(r1, 4 * t, 0, 0);
        in.q2 = __builtin_amdgcn_raw_buffer_load_b32(r2, 4 * t, 0, 0);
        in.a_first = res_aoff[rf]; in.a_end = res_aoff[re];
        typedef uint32_t u3v __attribute__((ext_vector_type(3)));
        const u3v w0 = __builtin_amdgcn_raw_buffer_load_b96(rb, 36 * t, 0, 0), w1 = __builtin_amdgcn_raw_buffer_load_b96(rb, 36 * t + 12, 0, 0),
                  w2 = __builtin_amdgcn_raw_buffer_load_b96(rb, 36 * t + 24, 0, 0);
        in.b0 = v3{__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
        in.b1 = v3{__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z)};
        in.b2 = v3{__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z)};
        return in;
    };
    // Stores and loads share one in-order counter (vmcnt). The wait for a tile's prefetched inputs at the top of the loop can
    // leave the previous tile's write-back in flight only if the number of stores issued since the prefetch is a compile-time
    // constant on EVERY path into the loop top: SC_WB buffer stores per thread and tile (the range check of the buffer
    // resource drops lanes past the end; a null range drops everything), also before the first tile and for a punted tile.
    // With a data-dependent store count every wavefront sat at the loop top until its write-back had been acknowledged by
    // the L2: 12 % of the kernel (tools/dbg/sc_timing.py).
    constexpr int SC_WB = 3 * (SC_CAP / BLOCK);
    constexpr uint32_t SC_SKIP = 0xffffffffu;      // staged x of an atom that is not this kernel's to store (the chain's OXT)
    constexpr int SC_SKIP_OFF = 0x7ffffff0;        // its store offset: past every range (a tile is < 10 KB), and offset + 4 does not wrap
    auto null_stores = [&]() {
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(out.x, 0, 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < SC_WB; u++) __builtin_amdgcn_raw_buffer_store_b32(0u, r0, 64 * u, 0, 0);   // apart: neither merged nor widened
    };
    if (blockIdx.x >= n_units) return;
    res_in nxt = load_res(unit_tile(blockIdx.x));
    null_stores();
#ifdef FCZ_SC_TIMING
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const uint32_t tile = unit_tile(unit);
        const uint32_t r = tile * tile_res + (uint32_t)t;
        const res_in cur = nxt;
        nxt = load_res(unit_tile(unit + gridDim.x < n_units ? unit + gridDim.x : unit));
        const uint32_t A0 = cur.a_first, A1 = cur.a_end;
        // the work-item list: a residue's N, CA, C are never items and of the rest at most 10 in 11 are (TRP: 14 atoms, 10 items),
        // so items <= 10 / 11 (atoms - 3 rows). For a full tile that fits SC_CAP the static_assert above bounds it; the batch's
        // LAST tile has fewer rows, and when nearly all of them are TRP / TYR / ARG (157 TRP: 2 199 atoms fit, 1 570 items do not)
        // its items overran the list (round 6: found by the differential fuzz, the tile's side chains came out as garbage)
        const uint32_t rows_here = (size_t)tile * tile_res < (size_t)n_res ? ((size_t)n_res - (size_t)tile * tile_res < tile_res ? (uint32_t)((size_t)n_res - (size_t)tile * tile_res) : tile_res) : 0u;
        if (A1 - A0 > (uint32_t)SC_CAP || 10u * (A1 - A0 - 3u * rows_here) > 11u * (uint32_t)SC_LIST) {
            // does not fit the staging buffer or the item list: both halves go to the 128-residue launch (never happens there:
            // 128 * 14 + OXT atoms, 128 * 10 items)
            if (t == 0 && punt_list) { const uint32_t k = atomicAdd(punt_count, 2u); punt_list[k] = 2u * tile; punt_list[k + 1] = 2u * tile + 1u; }
            null_stores();
            continue;
        }
        {
            const bool act = r < n_res && (uint32_t)t < tile_res;
            uint32_t rc = 23, na = 0;
            SC_STAMP(0)
            if (act) {
                const uint32_t ap = cur.a - A0;
                const v3 b0 = cur.b0, b1 = cur.b1, b2 = cur.b2;
                rc = cur.rc;
                na = S.natoms[rc];
                S.sc[0][t] = cur.q0; S.sc[1][t] = cur.q1; S.sc[2][t] = cur.q2;
                S.apos[t] = (uint16_t)ap;
                S.rc[t] = (uint8_t)rc;
                const uint32_t op4 = *reinterpret_cast<const uint32_t*>(&S.opos[rc][0]);
                const uint32_t p0 = ap + (op4 & 0xffu), p1 = ap + ((op4 >> 8) & 0xffu), p2 = ap + ((op4 >> 16) & 0xffu);
                S.stage[0][p0] = b0.x; S.stage[1][p0] = b0.y; S.stage[2][p0] = b0.z;
                S.stage[0][p1] = b1.x; S.stage[1][p1] = b1.y; S.stage[2][p1] = b1.z;
                S.stage[0][p2] = b2.x; S.stage[1][p2] = b2.y; S.stage[2][p2] = b2.z;
                if (out.atom_code) {
                    out.atom_code[A0 + p0] = fcz_res_atom[rc][0]; out.atom_code[A0 + p1] = fcz_res_atom[rc][1];
                    out.atom_code[A0 + p2] = fcz_res_atom[rc][2];
                }
                if (na > 3) {
                    // O: slot 3 of every residue type, predecessors N, CA, C (src/amino_acid.h; fcz_res_prev[*][3] == 0x210)
                    const sc_geom G = S.geom[rc][0];
                    const uint32_t q = cur.q0 & 0xffu;
                    v3 d2;
                    d2.x = G.d2x;
                    d2.y = G.blen * S.tor_cos[q] * G.sb;
                    d2.z = G.blen * S.tor_sin[q] * G.sb;
                    const v3 p = FAST ? place_atom_d2_fast(b0, b1, b2, d2) : place_atom_d2(b0, b1, b2, d2);
                    const uint32_t po = ap + (op4 >> 24);
                    S.stage[0][po] = p.x; S.stage[1][po] = p.y; S.stage[2][po] = p.z;
                    if (out.atom_code) out.atom_code[A0 + po] = (uint8_t)(G.meta >> 16);
                }
                // the chain's OXT (already written by k_res_index) lies inside this tile's output range: the write-back skips it
                if (cur.a_next - cur.a - na == 1) S.stage[0][ap + na] = __uint_as_float(SC_SKIP);
            }
            SC_STAMP(1)
            // ---- per-depth work lists: block-wide exclusive scan of the packed per-depth counts ----
            // (six 10-bit fields: the two 30-bit halves never carry into each other, so they scan as two dwords on the DPP network)
            const unsigned long long mine = act ? S.dcnt[rc] : 0ull;
            const uint32_t m_lo = (uint32_t)(mine & 0x3fffffffull), m_hi = (uint32_t)(mine >> 30);
            uint32_t t_lo, t_hi;
            const uint32_t e_lo = wave_excl_scan_dpp(m_lo, &t_lo), e_hi = wave_excl_scan_dpp(m_hi, &t_hi);
            const unsigned long long inc = (((unsigned long long)(e_hi + m_hi)) << 30) | (unsigned long long)(e_lo + m_lo);
            if (lane == WAVE - 1) S.wave_tot[wave] = inc;
            __syncthreads();
            SC_STAMP(2)
            unsigned long long pre = inc - mine, total = 0;
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; w++) { const unsigned long long v = S.wave_tot[w]; if (w < wave) pre += v; total += v; }
            // list ranges per depth (uniform) and this thread's insert position per depth; all shifts are static
            uint32_t ds = 0;
            const uint32_t* my_items = reinterpret_cast<const uint32_t*>(&S.items[rc][0]);
#pragma unroll
            for (int d = 0; d < SC_DEPTHS; d++) {
                if (t == 0) S.dstart[d] = ds;
                const uint32_t base = ds + (uint32_t)((pre >> (SC_FIELD * d)) & ((1u << SC_FIELD) - 1u));
                ds += (uint32_t)((total >> (SC_FIELD * d)) & ((1u << SC_FIELD) - 1u));
                if (na > 4) {
                    const uint32_t pack = my_items[d];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const uint32_t slot = (pack >> (8 * q)) & 0xffu;
                        if (slot) S.list[base + q] = (uint16_t)((uint32_t)t | (slot << 8));
                    }
                }
            }
            if (t == 0) S.dstart[SC_DEPTHS] = ds;
            SC_STAMP(3)
            __syncthreads();
            SC_STAMP(4)
            // ---- items, depth by depth ----
#pragma unroll 1
            for (int d = 0; d < SC_DEPTHS; d++) {
                const uint32_t dlo = S.dstart[d], dhi = S.dstart[d + 1];
                for (uint32_t i = dlo + (uint32_t)t; i < dhi; i += BLOCK) {
                    const uint32_t ent = S.list[i];
                    const uint32_t rl = ent & 255u, j = ent >> 8;
                    const sc_geom G = S.geom[S.rc[rl]][j - 3];
                    const uint32_t ap = S.apos[rl];
                    const uint32_t jj = j - 3;
                    const uint32_t q = (S.sc[jj >> 2][rl] >> (8 * (jj & 3u))) & 0xffu;
                    v3 d2;
                    d2.x = G.d2x;
                    d2.y = G.blen * S.tor_cos[q] * G.sb;
                    d2.z = G.blen * S.tor_sin[q] * G.sb;
                    const uint32_t ia = ap + (G.meta & 15u), ib = ap + ((G.meta >> 4) & 15u), ic = ap + ((G.meta >> 8) & 15u);
                    const v3 pa{S.stage[0][ia], S.stage[1][ia], S.stage[2][ia]};
                    const v3 pb{S.stage[0][ib], S.stage[1][ib], S.stage[2][ib]};
                    const v3 pc{S.stage[0][ic], S.stage[1][ic], S.stage[2][ic]};
                    const v3 p = FAST ? place_atom_d2_fast(pa, pb, pc, d2) : place_atom_d2(pa, pb, pc, d2);
                    const uint32_t po = ap + ((G.meta >> 12) & 15u);
                    S.stage[0][po] = p.x; S.stage[1][po] = p.y; S.stage[2][po] = p.z;
                    if (out.atom_code) out.atom_code[A0 + po] = (uint8_t)(G.meta >> 16);
                }
                SC_STAMP(5)
                __syncthreads();
                SC_STAMP(6)
            }
            // ---- write-back: the tile's atoms are one contiguous range of the output arrays ----
            {
                const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A1 - A0) * 4u));
                const size_t a0u = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)A0);
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(out.x + a0u, 0, (int)bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(out.y + a0u, 0, (int)bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(out.z + a0u, 0, (int)bytes, 0x00020000);
#pragma unroll
                for (int u = 0; u < SC_CAP / BLOCK; u++) {
                    const uint32_t i = (uint32_t)u * BLOCK + (uint32_t)t;
                    const uint32_t xb = __float_as_uint(S.stage[0][i]);
                    const int off = xb == SC_SKIP ? SC_SKIP_OFF : (int)(4u * i);     // out of every range: dropped
                    __builtin_amdgcn_raw_buffer_store_b32(xb, rx, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S.stage[1][i]), ry, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S.stage[2][i]), rz, off, 0, 0);
                }
            }
            SC_STAMP(7)
            __syncthreads();
            SC_STAMP(8)
        }
    }
#ifdef FCZ_SC_TIMING
    if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g_sc_timing[i], tacc[i]);
#endif
}

}  // namespace fcz
