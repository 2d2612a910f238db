// fcz_pdb.h -- PDB text of decompressed chains on the device (writeAtomCoordinatesToPDB, reference
// src/atom_coordinate.cpp:220-291; number formatting fast_ftoa<T,P> :185-218; TITLE wrap at 70 columns; TER record).
// One third of the reference's decompress CPU time and 81 bytes per atom of output: an HBM-bound byte kernel.
//
//   k_pdb_sizes    one wavefront per chain: exact text size (title lines + atom lines + TER); a line is 81 bytes unless
//                  a number overflows its column, in which case printf widens the field -- counted exactly.
//   k_pdb_format   one wavefront per chain, tiles of 64 residues: residues -> per-residue atom counts (wave scan), then
//                  lane = atom: the line is formatted byte by byte into an LDS staging buffer at its exact offset and the
//                  tile's bytes leave as coalesced dword stores.
// Everything the text needs besides the atoms comes from the FCZ header of the entry (title, chain id, residue and atom
// numbering, the OXT quirk: its residue number is header.nResidue and its residue name header.lastResidue,
// Foldcomp::read src/foldcomp.cpp:960-963).
#pragma once
#include "fcz_kernels.h"

namespace fcz {

// number of characters of printf("%d", v)
__device__ __forceinline__ uint32_t dec_len(int v) {
    uint32_t a = v < 0 ? (uint32_t)(-(long long)v) : (uint32_t)v;
    uint32_t d = 1;
    d += a >= 10u; d += a >= 100u; d += a >= 1000u; d += a >= 10000u; d += a >= 100000u; d += a >= 1000000u;
    d += a >= 10000000u; d += a >= 100000000u; d += a >= 1000000000u;
    return d + (v < 0 ? 1u : 0u);
}

// fast_ftoa<T,P> (reference src/atom_coordinate.cpp:185-218): r = n +- 0.5/T in float, I = (int)r, D = (int)((r - I) * T)
struct ftoa_parts { uint32_t I, D; bool neg; };
__device__ __forceinline__ ftoa_parts fast_ftoa_parts(float v, float T) {
    const float half = 0.5f / T;
    ftoa_parts p;
    p.neg = v < 0.0f;
    const float r = v + (p.neg ? -half : half);
    const int I = (int)r;
    const int D = (int)((r - (float)I) * T);
    p.I = (uint32_t)(I < 0 ? -I : I);
    p.D = (uint32_t)(D < 0 ? -D : D);
    return p;
}
__device__ __forceinline__ uint32_t ftoa_len(const ftoa_parts& p, uint32_t P) { return dec_len((int)p.I) + 1u + P + (p.neg ? 1u : 0u); }

// per-chain facts from the FCZ header
struct pdb_chain {
    uint32_t n, first_res, first_atom, title_len, o_title;
    uint8_t chain, last_letter;
};
__device__ __forceinline__ pdb_chain pdb_chain_of(const uint8_t* e) {
    pdb_chain c;
    c.n = ld_u16(e + 4);
    c.first_res = ld_u16(e + 8);
    c.first_atom = ld_u16(e + 10);
    c.chain = e[13];
    c.last_letter = e[21];
    c.title_len = ld_u32(e + 24);
    c.o_title = 76 + 4 * (uint32_t)e[12];
    return c;
}
// TITLE records: "TITLE     " + 70 characters, continuation lines "TITLE  " + "% 3d" of the line number
__device__ __forceinline__ uint32_t title_cont_width(uint32_t k) { const uint32_t w = dec_len((int)k) + 1u; return w < 3u ? 3u : w; }
__device__ __forceinline__ uint32_t title_line_bytes(uint32_t line, uint32_t title_len) {   // line = 0, 1, ...
    const uint32_t chars = title_len - 70u * line < 70u ? title_len - 70u * line : 70u;
    return (line == 0 ? 10u : 7u + title_cont_width(line + 1)) + chars + 1u;
}

// bytes of one ATOM line beyond the 81 of the fixed-column case
__device__ __forceinline__ uint32_t atom_line_extra(int serial, int resnum, float x, float y, float z, float b) {
    uint32_t ex = 0, l;
    l = dec_len(serial); ex += l > 5u ? l - 5u : 0u;
    l = dec_len(resnum); ex += l > 4u ? l - 4u : 0u;
    l = ftoa_len(fast_ftoa_parts(x, 1000.0f), 3); ex += l > 8u ? l - 8u : 0u;
    l = ftoa_len(fast_ftoa_parts(y, 1000.0f), 3); ex += l > 8u ? l - 8u : 0u;
    l = ftoa_len(fast_ftoa_parts(z, 1000.0f), 3); ex += l > 8u ? l - 8u : 0u;
    l = ftoa_len(fast_ftoa_parts(b, 100.0f), 2); ex += l > 6u ? l - 6u : 0u;
    return ex;
}

// residue / serial numbering of output atom i of a chain whose last atom may be the OXT
struct pdb_atom_ctx { uint32_t n_atoms; bool has_oxt; };

__global__ __launch_bounds__(BLOCK) void k_pdb_sizes(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries,
                                                     const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ atom_off,
                                                     fcz_atoms_out at, uint64_t* __restrict__ text_size) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= n_entries) return;
    const uint32_t r0 = res_off[c], n = res_off[c + 1] - r0;
    if (n == 0) { if (lane == 0) text_size[c] = 0; return; }
    const uint8_t* e = blob + off[c];
    const pdb_chain H = pdb_chain_of(e);
    const uint32_t a0 = atom_off[c], n_atoms = atom_off[c + 1] - a0;
    unsigned long long bytes = 0;
    // title
    const uint32_t n_lines = (H.title_len + 69u) / 70u;
    for (uint32_t l = lane; l < n_lines; l += WAVE) bytes += title_line_bytes(l, H.title_len);
    // atoms: residue by residue (lane = residue), every atom of a residue shares resnum and B-factor
    uint32_t run = 0;
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        uint32_t na = 0;
        if (k < n) { const uint32_t rc = at.res_code[r0 + k]; na = fcz_res_natoms[rc < 24 ? rc : 23]; }
        uint32_t tot;
        const uint32_t ex = run + wave_excl_scan(na, lane, &tot);
        run += tot;
        if (k < n) {
            const float b = at.bfac_res[r0 + k];
            for (uint32_t j = 0; j < na; j++) {
                const uint32_t i = ex + j;
                bytes += 81u + atom_line_extra((int)(H.first_atom + i), (int)(H.first_res + k), at.x[a0 + i], at.y[a0 + i], at.z[a0 + i], b);
            }
        }
    }
    const bool has_oxt = n_atoms == run + 1;
    if (lane == 0) {
        int last_resnum = (int)(H.first_res + n - 1);
        if (has_oxt) {
            const uint32_t i = run;
            last_resnum = (int)H.n;
            bytes += 81u + atom_line_extra((int)(H.first_atom + i), last_resnum, at.x[a0 + i], at.y[a0 + i], at.z[a0 + i], at.bfac_res[r0 + n - 1]);
        }
        // "TER   %5d      %3s %s%4d\n"
        uint32_t l = dec_len((int)(H.first_atom + n_atoms)); bytes += 27u + (l > 5u ? l - 5u : 0u);
        l = dec_len(last_resnum); bytes += (l > 4u ? l - 4u : 0u);
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) bytes += __shfl_xor(bytes, d, WAVE);
    if (lane == 0) text_size[c] = bytes;
}

// ---- byte emitter into the wave's LDS staging buffer ----
struct line_writer {
    uint8_t* buf; uint32_t pos;
    __device__ __forceinline__ void ch(uint32_t c) { buf[pos++] = (uint8_t)c; }
    __device__ __forceinline__ void spaces(uint32_t k) { for (uint32_t i = 0; i < k; i++) buf[pos++] = ' '; }
    __device__ __forceinline__ void str(const char* s, uint32_t k) { for (uint32_t i = 0; i < k; i++) buf[pos++] = (uint8_t)s[i]; }
    // printf("%*d"): right-aligned in `width`, wider if needed
    __device__ __forceinline__ void dec(int v, uint32_t width) {
        const uint32_t l = dec_len(v);
        if (l < width) spaces(width - l);
        uint32_t a = v < 0 ? (uint32_t)(-(long long)v) : (uint32_t)v;
        if (v < 0) ch('-');
        const uint32_t nd = l - (v < 0 ? 1u : 0u);
        uint32_t p = pos + nd;
        for (uint32_t i = 0; i < nd; i++) { buf[--p] = (uint8_t)('0' + a % 10u); a /= 10u; }
        pos += nd;
    }
    // "%*s" of fast_ftoa<T,P>(v)
    __device__ __forceinline__ void num(float v, float T, uint32_t P, uint32_t width) {
        const ftoa_parts f = fast_ftoa_parts(v, T);
        const uint32_t l = ftoa_len(f, P);
        if (l < width) spaces(width - l);
        if (f.neg) ch('-');
        const uint32_t nd = dec_len((int)f.I);
        uint32_t a = f.I, p = pos + nd;
        for (uint32_t i = 0; i < nd; i++) { buf[--p] = (uint8_t)('0' + a % 10u); a /= 10u; }
        pos += nd;
        ch('.');
        uint32_t d = f.D; p = pos + P;
        for (uint32_t i = 0; i < P; i++) { buf[--p] = (uint8_t)('0' + d % 10u); d /= 10u; }
        pos += P;
    }
};

constexpr int PDB_STAGE = 64 * 128;   // bytes of staging per wavefront: 64 lines of at most 128 bytes

__device__ __forceinline__ uint32_t pdb_atom_line_len(int serial, int resnum, float x, float y, float z, float b) {
    return 81u + atom_line_extra(serial, resnum, x, y, z, b);
}

// "ATOM  %5d %s %3s %s%4d    %8s%8s%8s  1.00%6s          %2s  \n"
__device__ __forceinline__ void pdb_atom_line(line_writer& w, int serial, uint32_t atom_code, const char* res3, uint32_t chain, int resnum,
                                              float x, float y, float z, float b) {
    w.str("ATOM  ", 6);
    w.dec(serial, 5);
    w.ch(' ');
    const char* nm = fcz_atom_name[atom_code < FCZ_N_ATOM_CODES ? atom_code : 0];
    const uint32_t nl = nm[1] == 0 ? 1u : (nm[2] == 0 ? 2u : (nm[3] == 0 ? 3u : 4u));
    if (nl < 4) { w.ch(' '); w.str(nm, nl); w.spaces(3 - nl); } else w.str(nm, 4);
    w.ch(' ');
    w.str(res3, 3);
    w.ch(' ');
    w.ch(chain);
    w.dec(resnum, 4);
    w.spaces(4);
    w.num(x, 1000.0f, 3, 8); w.num(y, 1000.0f, 3, 8); w.num(z, 1000.0f, 3, 8);
    w.str("  1.00", 6);
    w.num(b, 100.0f, 2, 6);
    w.spaces(10);
    w.ch(' '); w.ch((uint32_t)nm[0]);
    w.ch(' '); w.ch(' '); w.ch('\n');
}

__global__ __launch_bounds__(BLOCK) void k_pdb_format(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries,
                                                      const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ atom_off,
                                                      fcz_atoms_out at, int alt_order, const uint64_t* __restrict__ text_off,
                                                      uint8_t* __restrict__ text) {
    __shared__ alignas(16) uint8_t s_stage[WAVES_PER_BLOCK][PDB_STAGE];
    __shared__ uint16_t s_res_first[WAVES_PER_BLOCK][WAVE + 1];   // first atom (tile-local) of each residue of the tile
    __shared__ uint8_t s_atom_res[WAVES_PER_BLOCK][WAVE * FCZ_MAX_RES_ATOMS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (c >= n_entries) return;
    const uint32_t r0 = res_off[c], n = res_off[c + 1] - r0;
    if (n == 0) return;
    const uint8_t* e = blob + off[c];
    const pdb_chain H = pdb_chain_of(e);
    const uint32_t a0 = atom_off[c], n_atoms = atom_off[c + 1] - a0;
    uint8_t* dst = text + text_off[c];
    uint8_t* stage = &s_stage[wave][0];
    unsigned long long written = 0;   // bytes of this chain already stored (uniform)

    // copies the first `bytes` staged bytes to the output; dst + written may be unaligned: head bytes, dwords, tail bytes
    auto flush = [&](uint32_t bytes) {
        __builtin_amdgcn_wave_barrier();
        uint8_t* d = dst + written;
        const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) < bytes ? (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) : bytes;
        if ((uint32_t)lane < head) d[lane] = stage[lane];
        const uint32_t nd = (bytes - head) >> 2;
        for (uint32_t i = lane; i < nd; i += WAVE) {
            const uint8_t* s = stage + head + 4 * i;
            const uint32_t v = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
            *reinterpret_cast<uint32_t*>(d + head + 4 * i) = v;
        }
        const uint32_t tail0 = head + 4 * nd;
        if (tail0 + lane < bytes) d[tail0 + lane] = stage[tail0 + lane];
        written += bytes;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- TITLE records: lane = line, 64 lines per round ----
    const uint32_t n_lines = (H.title_len + 69u) / 70u;
    for (uint32_t lb = 0; lb < n_lines; lb += WAVE) {
        const uint32_t l = lb + lane;
        const uint32_t len = l < n_lines ? title_line_bytes(l, H.title_len) : 0u;
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(len, lane, &tot);
        if (l < n_lines) {
            line_writer w{stage, ex};
            if (l == 0) w.str("TITLE     ", 10);
            else { w.str("TITLE  ", 7); w.dec((int)(l + 1), title_cont_width(l + 1)); }
            const uint32_t chars = H.title_len - 70u * l < 70u ? H.title_len - 70u * l : 70u;
            const uint8_t* t = e + H.o_title + 70u * l;
            for (uint32_t i = 0; i < chars; i++) w.ch(t[i]);
            w.ch('\n');
        }
        flush(tot);
    }

    // ---- ATOM records ----
    uint32_t run = 0;          // atoms of the residues before the tile
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        uint32_t na = 0, rc = 23;
        if (k < n) { rc = at.res_code[r0 + k]; rc = rc < 24 ? rc : 23; na = fcz_res_natoms[rc]; }
        uint32_t tile_atoms;
        const uint32_t ex = wave_excl_scan(na, lane, &tile_atoms);
        s_res_first[wave][lane] = (uint16_t)ex;
        for (uint32_t j = 0; j < na; j++) s_atom_res[wave][ex + j] = (uint8_t)lane;
        __builtin_amdgcn_wave_barrier();
        // the OXT closes the chain: one more line after the last residue's atoms
        const bool last_tile = base + WAVE >= n;
        const bool has_oxt = last_tile && n_atoms == run + tile_atoms + 1;
        const uint32_t lines = tile_atoms + (has_oxt ? 1u : 0u);
        for (uint32_t ab = 0; ab < lines; ab += WAVE) {
            const uint32_t ai = ab + lane;          // atom of the tile
            const bool on = ai < lines;
            const bool oxt = on && ai == tile_atoms;
            uint32_t rl = 0, j = 0;
            if (on && !oxt) { rl = s_atom_res[wave][ai]; j = ai - s_res_first[wave][rl]; }
            const uint32_t kk = oxt ? n - 1 : base + rl;
            uint32_t rcl = 23; float b = 0.f, x = 0.f, y = 0.f, z = 0.f;
            if (on) {
                rcl = at.res_code[r0 + kk]; rcl = rcl < 24 ? rcl : 23;
                b = at.bfac_res[r0 + kk];
                const uint32_t g = a0 + run + ai;
                x = at.x[g]; y = at.y[g]; z = at.z[g];
            }
            const int serial = (int)(H.first_atom + run + ai);
            const int resnum = oxt ? (int)H.n : (int)(H.first_res + kk);
            const uint32_t len = on ? pdb_atom_line_len(serial, resnum, x, y, z, b) : 0u;
            uint32_t tot;
            const uint32_t ofs = wave_excl_scan(len, lane, &tot);
            if (on) {
                uint32_t acode = FCZ_ATOM_OXT;
                const char* r3 = fcz_res3[rcl];
                if (oxt) {
                    int li = 23;
                    for (int q = 0; q < 24; q++) if ((uint8_t)fcz_res1[q] == H.last_letter) { li = q; break; }
                    r3 = fcz_res3[li];
                } else {
                    const uint32_t slot = alt_order ? fcz_res_alt_slot[rcl][j] : j;
                    acode = fcz_res_atom[rcl][slot];
                }
                line_writer w{stage, ofs};
                pdb_atom_line(w, serial, acode, r3, H.chain, resnum, x, y, z, b);
            }
            flush(tot);
        }
        run += tile_atoms;
        __builtin_amdgcn_wave_barrier();
    }
    // ---- TER: "TER   %5d      %3s %s%4d\n" with the residue of the last atom ----
    {
        const bool has_oxt = n_atoms == run + 1;
        uint32_t len = 0;
        if (lane == 0) {
            uint32_t rcl = at.res_code[r0 + n - 1]; rcl = rcl < 24 ? rcl : 23;
            const char* r3 = fcz_res3[rcl];
            int resnum = (int)(H.first_res + n - 1);
            if (has_oxt) {
                int li = 23;
                for (int q = 0; q < 24; q++) if ((uint8_t)fcz_res1[q] == H.last_letter) { li = q; break; }
                r3 = fcz_res3[li];
                resnum = (int)H.n;
            }
            line_writer w{stage, 0};
            w.str("TER   ", 6); w.dec((int)(H.first_atom + n_atoms), 5); w.spaces(6); w.str(r3, 3); w.ch(' '); w.ch(H.chain);
            w.dec(resnum, 4); w.ch('\n');
            len = w.pos;
        }
        len = __shfl(len, 0, WAVE);
        flush(len);
    }
}

}  // namespace fcz
