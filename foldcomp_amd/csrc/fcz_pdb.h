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
#include <type_traits>

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

// fast_ftoa<T,P> (reference src/atom_coordinate.cpp:185-218): r = n +- 0.5/T in float, I = (int)r, D = (int)((r - I) * T).
// odd: r is a NaN, infinite or beyond the int range. (int) of such a float is INT_MIN on x86-64 (cvttss2si's "integer indefinite"),
// for I and then for D as well, std::abs leaves it INT_MIN, and itoa_pos_only (:172-183) ends after ONE character for a negative
// number: '0' + INT_MIN % 10 = '('. The number reads "(.00(" ("-(.00(" for a negative infinity); what a decoded NaN looks like in
// the reference's text (a record whose quantiser parameters are NaN).
struct ftoa_parts { uint32_t I, D; bool neg, odd; };
__device__ __forceinline__ ftoa_parts fast_ftoa_parts(float v, float T) {
    const float half = 0.5f / T;
    ftoa_parts p;
    p.neg = v < 0.0f;
    const float r = v + (p.neg ? -half : half);
    p.odd = !(__builtin_fabsf(r) < 2147483648.0f);
    const int I = p.odd ? 0 : (int)r;
    const int D = p.odd ? 0 : (int)((r - (float)I) * T);
    p.I = (uint32_t)(I < 0 ? -I : I);
    p.D = (uint32_t)(D < 0 ? -D : D);
    return p;
}
__device__ __forceinline__ uint32_t ftoa_len(const ftoa_parts& p, uint32_t P) { return (p.odd ? 1u : dec_len((int)p.I)) + 1u + P + (p.neg ? 1u : 0u); }

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
                                                     fcz_atoms_out at, uint32_t pad, uint64_t* __restrict__ text_size) {
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
    // atoms. A line's extra bytes are a sum of per-field overflows, so the fields go where they read contiguously: what hangs on the
    // residue (residue number, B-factor: shared by its atoms) with lane = residue, what hangs on the atom (serial number, x, y, z)
    // with lane = atom over the chain's flat atom range (round 4: the per-residue walk over its atoms read every coordinate plane
    // at a 33-byte lane stride: 3.4x the bytes by counter)
    auto over = [](uint32_t l, uint32_t w) { return l > w ? l - w : 0u; };
    uint32_t run = 0;
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        uint32_t na = 0;
        if (k < n) { const uint32_t rc = at.res_code[r0 + k]; na = fcz_res_natoms[rc < 24 ? rc : 23]; }
        if (k < n) bytes += (unsigned long long)na * (81u + over(dec_len((int)(H.first_res + k)), 4u) + over(ftoa_len(fast_ftoa_parts(at.bfac_res[r0 + k], 100.0f), 2), 6u));
        run += wave_sum(na);
    }
    for (uint32_t i0 = 0; i0 < run; i0 += WAVE) {
        const uint32_t i = i0 + lane;
        if (i < run) {
            bytes += over(dec_len((int)(H.first_atom + i)), 5u) + over(ftoa_len(fast_ftoa_parts(at.x[a0 + i], 1000.0f), 3), 8u) +
                     over(ftoa_len(fast_ftoa_parts(at.y[a0 + i], 1000.0f), 3), 8u) + over(ftoa_len(fast_ftoa_parts(at.z[a0 + i], 1000.0f), 3), 8u);
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
    if (lane == 0) text_size[c] = bytes + pad;      // pad: the terminator a database entry carries (left zero by the caller)
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
        if (f.odd) { ch('('); ch('.'); for (uint32_t i = 1; i < P; i++) ch('0'); ch('('); return; }
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

constexpr int PDB_STAGE = 64 * 128 + 16;   // bytes of staging per wavefront: 64 lines of at most 128 bytes (+ alignment pad)

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

// ---- fixed-column fast path: the 81-byte record assembled in registers at compile-time byte positions ----------
struct line81 {
    uint32_t w[21];
    template <int POS> __device__ __forceinline__ void put(uint32_t c) { w[POS >> 2] |= c << (8 * (POS & 3)); }
};
// "ATOM  " ... "  1.00" ... "\n" with zero bytes in the columns the fields fill: 6-10 serial, 12-15 name, 17-19 residue,
// 21 chain, 22-25 residue number, 30-53 x y z, 60-65 B-factor, 77 element
__host__ __device__ constexpr uint32_t line81_char(int p) {
    return p == 0 ? 'A' : p == 1 ? 'T' : p == 2 ? 'O' : p == 3 ? 'M' : p == 56 ? '1' : p == 57 ? '.' : (p == 58 || p == 59) ? '0' : p == 80 ? '\n' :
           ((p >= 6 && p <= 10) || (p >= 12 && p <= 15) || (p >= 17 && p <= 19) || (p >= 21 && p <= 25) || (p >= 30 && p <= 53) ||
            (p >= 60 && p <= 65) || p == 77 || p > 80) ? 0u : ' ';
}
__host__ __device__ constexpr uint32_t line81_template(int i) {
    return line81_char(4 * i) | (line81_char(4 * i + 1) << 8) | (line81_char(4 * i + 2) << 16) | (line81_char(4 * i + 3) << 24);
}
// printf("%*d") of a value known to fit: digits right to left, then the sign, then blanks
template <int COL, int W>
__device__ __forceinline__ void put_int(line81& L, uint32_t a, bool neg) {
    bool sign = neg;
    auto step = [&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t q = a / 10u, d = a - 10u * q;
        uint32_t c;
        if (i == 0 || a != 0u) c = '0' + d;
        else if (sign) { c = '-'; sign = false; }
        else c = ' ';
        L.put<COL + W - 1 - i>(c);
        a = q;
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (W > 1) step(std::integral_constant<int, 1>{});
    if constexpr (W > 2) step(std::integral_constant<int, 2>{});
    if constexpr (W > 3) step(std::integral_constant<int, 3>{});
    if constexpr (W > 4) step(std::integral_constant<int, 4>{});
}
// "%*s" of fast_ftoa<T,P>(v) known to fit: WI columns for sign + integer digits, '.', P decimals
template <int COL, int WI, int P>
__device__ __forceinline__ void put_num(line81& L, const ftoa_parts& f) {
    if (__builtin_expect(f.odd, 0)) {
        L.put<COL + WI - 1>('('); L.put<COL + WI - 2>(f.neg ? '-' : ' ');
        if constexpr (WI > 2) L.put<COL + WI - 3>(' ');
        if constexpr (WI > 3) L.put<COL + WI - 4>(' ');
        L.put<COL + WI>('.');
        if constexpr (P > 1) L.put<COL + WI + 1>('0');
        if constexpr (P > 2) L.put<COL + WI + 2>('0');
        L.put<COL + WI + P>('(');
        return;
    }
    put_int<COL, WI>(L, f.I, f.neg);
    L.put<COL + WI>('.');
    uint32_t d = f.D;
    {
        const uint32_t q = d / 10u; L.put<COL + WI + P>('0' + (d - 10u * q)); d = q;
    }
    if constexpr (P > 1) { const uint32_t q = d / 10u; L.put<COL + WI + P - 1>('0' + (d - 10u * q)); d = q; }
    if constexpr (P > 2) { const uint32_t q = d / 10u; L.put<COL + WI + P - 2>('0' + (d - 10u * q)); d = q; }
}
// the record of one atom whose every number fits its column (line length 81), stored at `dst` (LDS, any alignment)
__device__ __forceinline__ void pdb_atom_line_fixed(uint8_t* dst, int serial, uint32_t atom_code, uint32_t res3_bits, uint32_t chain,
                                                    int resnum, const ftoa_parts& fx, const ftoa_parts& fy, const ftoa_parts& fz,
                                                    const ftoa_parts& fb) {
    line81 L;
#pragma unroll
    for (int i = 0; i < 21; i++) L.w[i] = line81_template(i);   // fixed text in place, zero bytes where fields are OR-ed in
    put_int<6, 5>(L, (uint32_t)(serial < 0 ? -serial : serial), serial < 0);
    // atom name: 4 columns; names shorter than 4 start in the second column
    const char* nm = fcz_atom_name[atom_code < FCZ_N_ATOM_CODES ? atom_code : 0];
    const uint32_t n0 = (uint8_t)nm[0], n1 = (uint8_t)nm[1], n2 = n1 ? (uint8_t)nm[2] : 0u, n3 = n2 ? (uint8_t)nm[3] : 0u;
    const bool four = n3 != 0;
    L.put<12>(four ? n0 : ' '); L.put<13>(four ? n1 : n0); L.put<14>(four ? n2 : (n1 ? n1 : ' ')); L.put<15>(four ? n3 : (n2 ? n2 : ' '));
    L.put<17>(res3_bits & 0xffu); L.put<18>((res3_bits >> 8) & 0xffu); L.put<19>((res3_bits >> 16) & 0xffu);
    L.put<21>(chain);
    put_int<22, 4>(L, (uint32_t)(resnum < 0 ? -resnum : resnum), resnum < 0);
    put_num<30, 4, 3>(L, fx);
    put_num<38, 4, 3>(L, fy);
    put_num<46, 4, 3>(L, fz);
    put_num<60, 3, 2>(L, fb);
    L.put<77>(n0);
#pragma unroll
    for (int p = 0; p < 81; p++) dst[p] = (uint8_t)(L.w[p >> 2] >> (8 * (p & 3)));
}

__global__ __launch_bounds__(BLOCK) void k_pdb_format(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries,
                                                      const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ atom_off,
                                                      fcz_atoms_out at, int alt_order, const uint64_t* __restrict__ text_off,
                                                      uint8_t* __restrict__ text) {
    __shared__ alignas(16) uint8_t s_stage[WAVES_PER_BLOCK][PDB_STAGE];
    __shared__ uint16_t s_res_first[WAVES_PER_BLOCK][WAVE + 1];   // first atom (tile-local) of each residue of the tile
    __shared__ uint8_t s_atom_res[WAVES_PER_BLOCK][WAVE * FCZ_MAX_RES_ATOMS];
    __shared__ uint8_t s_res_rc[WAVES_PER_BLOCK][WAVE];
    __shared__ float s_res_bf[WAVES_PER_BLOCK][WAVE];
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

    // The staged bytes start `pad` bytes into the staging buffer, pad = alignment of their destination, so that aligned
    // dwords of the stage are aligned dwords of the output: partial head dword, whole dwords, partial tail dword.
    auto pad_now = [&]() -> uint32_t { return (uint32_t)((uintptr_t)(dst + written) & 3u); };
    auto flush = [&](uint32_t bytes) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t pad = pad_now();
        uint8_t* d0 = dst + written - pad;                 // 4-byte aligned
        const uint32_t end = pad + bytes;
        const uint32_t first = pad ? 1u : 0u, full = end >> 2;
        if (pad && (uint32_t)lane >= pad && (uint32_t)lane < (end < 4u ? end : 4u)) d0[lane] = stage[lane];
        for (uint32_t i = first + (uint32_t)lane; i < full; i += WAVE)
            *reinterpret_cast<uint32_t*>(d0 + 4 * i) = *reinterpret_cast<const uint32_t*>(stage + 4 * i);
        const uint32_t tail0 = (full < first ? first : full) * 4u;
        if (tail0 + (uint32_t)lane < end) d0[tail0 + lane] = stage[tail0 + lane];
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
            line_writer w{stage, pad_now() + ex};
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
    // residue name of the OXT record: header.lastResidue (one letter) -> three letters
    int oxt_rc = 23;
    for (int q = 0; q < 24; q++) if ((uint8_t)fcz_res1[q] == H.last_letter) { oxt_rc = q; break; }
    uint32_t run = 0;          // atoms of the residues before the tile
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t k = base + lane;
        uint32_t na = 0, rc = 23; float bres = 0.f;
        if (k < n) { rc = at.res_code[r0 + k]; bres = at.bfac_res[r0 + k]; rc = rc < 24 ? rc : 23; na = fcz_res_natoms[rc]; }
        uint32_t tile_atoms;
        const uint32_t ex = wave_excl_scan(na, lane, &tile_atoms);
        s_res_first[wave][lane] = (uint16_t)ex;
        s_res_rc[wave][lane] = (uint8_t)rc;
        s_res_bf[wave][lane] = bres;
        for (uint32_t j = 0; j < na; j++) s_atom_res[wave][ex + j] = (uint8_t)lane;
        __builtin_amdgcn_wave_barrier();
        // the OXT closes the chain: one more line after the last residue's atoms
        const bool last_tile = base + WAVE >= n;
        const bool has_oxt = last_tile && n_atoms == run + tile_atoms + 1;
        const uint32_t lines = tile_atoms + (has_oxt ? 1u : 0u);
        // coordinates of the first round; every round loads the next one's before it formats its own
        float nx = 0.f, ny = 0.f, nz = 0.f;
        { const uint32_t g = a0 + run + (lane < (int)lines ? (uint32_t)lane : 0u); if (lines) { nx = at.x[g]; ny = at.y[g]; nz = at.z[g]; } }
        for (uint32_t ab = 0; ab < lines; ab += WAVE) {
            const uint32_t ai = ab + lane;          // atom of the tile
            const bool on = ai < lines;
            const bool oxt = on && ai == tile_atoms;
            const float x = nx, y = ny, z = nz;
            {
                const uint32_t an = ai + WAVE;
                const uint32_t g = a0 + run + (an < lines ? an : ab);   // clamped: unconditional loads
                nx = at.x[g]; ny = at.y[g]; nz = at.z[g];
            }
            uint32_t rl = 0, j = 0;
            if (on && !oxt) { rl = s_atom_res[wave][ai]; j = ai - s_res_first[wave][rl]; }
            if (oxt) rl = (n - 1) - base;
            const uint32_t rcl = s_res_rc[wave][rl];
            const float b = s_res_bf[wave][rl];
            const int serial = (int)(H.first_atom + run + ai);
            const int resnum = oxt ? (int)H.n : (int)(H.first_res + base + rl);
            // does every number fit its column? (then the line is exactly 81 bytes)
            const ftoa_parts fx = fast_ftoa_parts(x, 1000.0f), fy = fast_ftoa_parts(y, 1000.0f), fz = fast_ftoa_parts(z, 1000.0f),
                             fb = fast_ftoa_parts(b, 100.0f);
            const bool fits = serial <= 99999 && serial >= -9999 && resnum <= 9999 && resnum >= -999 &&
                              fx.I <= (fx.neg ? 999u : 9999u) && fy.I <= (fy.neg ? 999u : 9999u) && fz.I <= (fz.neg ? 999u : 9999u) &&
                              fb.I <= (fb.neg ? 99u : 999u);
            const uint32_t n_on = lines - ab < (uint32_t)WAVE ? lines - ab : (uint32_t)WAVE;
            const bool fixed = !__any(on && !fits);
            uint32_t acode = FCZ_ATOM_OXT, r3rc = (uint32_t)oxt_rc;
            if (!oxt) { const uint32_t slot = alt_order ? fcz_res_alt_slot[rcl][j] : j; acode = fcz_res_atom[rcl][slot]; r3rc = rcl; }
            const char* r3 = fcz_res3[r3rc];
            uint32_t tot = 81u * n_on;
            if (fixed) {
                if (on) {
                    const uint32_t r3b = (uint32_t)(uint8_t)r3[0] | ((uint32_t)(uint8_t)r3[1] << 8) | ((uint32_t)(uint8_t)r3[2] << 16);
                    pdb_atom_line_fixed(stage + pad_now() + 81u * (uint32_t)lane, serial, acode, r3b, H.chain, resnum, fx, fy, fz, fb);
                }
            } else {
                const uint32_t len = on ? pdb_atom_line_len(serial, resnum, x, y, z, b) : 0u;
                const uint32_t ofs = wave_excl_scan(len, lane, &tot);
                if (on) {
                    line_writer w{stage, pad_now() + ofs};
                    pdb_atom_line(w, serial, acode, r3, H.chain, resnum, x, y, z, b);
                }
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
            if (has_oxt) { r3 = fcz_res3[oxt_rc]; resnum = (int)H.n; }
            line_writer w{stage, pad_now()};
            w.str("TER   ", 6); w.dec((int)(H.first_atom + n_atoms), 5); w.spaces(6); w.str(r3, 3); w.ch(' '); w.ch(H.chain);
            w.dec(resnum, 4); w.ch('\n');
            len = w.pos - pad_now();
        }
        len = __shfl(len, 0, WAVE);
        flush(len);
    }
}

}  // namespace fcz
