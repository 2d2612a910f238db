// fcz_ingest.h -- structure ingest on the device: PDB text -> fcz_chain_batch.
//
// What the reference does per input file before Foldcomp::compress sees an atom span, restated for a wavefront per file:
//   StructureReader (gemmi) -> atom table                 reference src/structure_reader.cpp:31-61; the fixed-column ATOM /
//                                                         HETATM record as the Python binding reads it, foldcomp/foldcomp.cxx:259-278
//   removeAlternativePosition                             src/atom_coordinate.cpp:362-370
//   identifyChains                                        src/atom_coordinate.cpp:469-497
//   identifyDiscontinousResInd                            src/atom_coordinate.cpp:506-530
//   per fragment: splitAtomByResidue, residue codes, the CA B-factor, N/CA/C present and in order
//                                                         src/atom_coordinate.cpp:304-328, src/foldcomp.cpp:450-559, src/main.cpp:455-508
//   title: _entry.id = HEADER id code, else the TITLE records  (gemmi; src/main.cpp:465)
//
//   k_ingest_parse   wavefront = file. The file is walked in 4 KB chunks: every lane finds the line ends in its 64 bytes, a
//                    wave scan numbers them, then lane = line: record type, fixed-column fields, numbers by the exact
//                    integer-digits x reciprocal scheme the host parser proves (host/foldcomp_hip.cpp fixed_field), names as packed
//                    integers -> codes through an LDS hash. An atom whose name equals that of the record before it is dropped
//                    (that IS removeAlternativePosition: see keep rule below). Kept atoms are appended to the file's slice of
//                    the scratch atom table. Anything outside the fixed layout marks the FILE for the host parser
//                    (FCZ_INGEST_HOST_*): the device never guesses.
//   k_ingest_frags   wavefront = file: chain cuts, gap cuts, residue boundaries, per-residue validation -> fragment list,
//                    per-residue scratch (first atom, code, CA B-factor), per-file totals of what was accepted.
//   k_ingest_fill    wavefront = file: accepted fragments copied to their place in the batch arrays (offsets from scans over
//                    the per-file totals).
#pragma once
#include "fcz_kernels.h"

namespace fcz {

// per-file status beyond FCZ_OK: the file is left to the host parser (nothing of it enters the batch)
constexpr int32_t FCZ_INGEST_HOST_FIELD = 1;      // a number / integer field outside the fixed layout, or a short ATOM record
constexpr int32_t FCZ_INGEST_HOST_TITLE = 2;      // title longer than the device buffer
constexpr int32_t FCZ_INGEST_HOST_FRAGS = 3;      // more fragments than the per-file list holds
constexpr int32_t FCZ_INGEST_NO_ATOMS = 4;        // "[Error] No atoms found in the input file" (src/main.cpp:459-462)

constexpr int IG_CHUNK = 4096;                    // bytes per wavefront step (64 lanes x 64 bytes)
constexpr int IG_LINES = 1024;                    // line ends one chunk can hold in LDS before it is split (81-byte lines: 51)
constexpr int IG_TITLE_CAP = 512;                 // bytes of title per file
constexpr int IG_MAX_FRAGS = 32;                  // fragments per file

// fragment refusal reasons (what Batch::prepare of the host throws), chain_meta of a refused fragment carries them << 24
constexpr uint32_t IG_REF_NONE = 0, IG_REF_RESNAME = 1, IG_REF_BACKBONE = 2, IG_REF_TOO_LONG = 3, IG_REF_SKIP_DISC = 4, IG_REF_BACKBONE_TWICE = 5, IG_REF_LAST_NAME = 6;
constexpr uint32_t IG_META_MULTI_CHAIN = 1u << 16, IG_META_MULTI_FRAG = 1u << 17;

struct ingest_scratch {       // per-atom table of the kept atoms, file f at [abase[f], abase[f] + n_kept[f])
    uint32_t* name;           // packed atom name
    uint32_t* resn;           // packed residue name
    int32_t* serial;          // atom serial number
    int32_t* resseq;          // residue sequence number
    float *x, *y, *z, *b;
    uint32_t* chain;          // chain name: up to four characters packed (low byte = the first: the FCZ header's chain character)
    uint8_t* acode;           // codec atom code (255 = other)
    int8_t* rcode;            // codec residue code of the atom's residue name (-1 = not one the codec takes)
    // per residue, file f at [abase[f] ...): filled by k_ingest_frags
    uint32_t* r_first;        // first atom (file-local index) of the residue
    float* r_bfac;            // B-factor of its CA atom
    uint8_t* r_code;
};
struct ingest_frag { uint32_t a, b, r0, nres, meta; };   // atoms [a, b) and residues [r0, r0 + nres) file-local; meta: chain | ordinal << 8 | flags | reason << 24

__device__ __forceinline__ bool ig_is_space(uint32_t c) { return c == ' ' || (c - 9u) < 5u; }   // isspace of the C locale
__device__ __forceinline__ uint32_t ig_prev_lane(unsigned long long mask, int lane) {            // highest set bit below `lane`, 64 if none
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    return below ? 63u - (uint32_t)__builtin_clzll(below) : 64u;
}

// a stripped field of up to four characters as a packed name (host: field_pack)
__device__ __forceinline__ uint32_t ig_pack4(uint32_t w, int n) {   // w = the n (<= 4) field bytes, little endian
    // branch-free: blanks counted from both ends by selects, the rest shifted down and masked
    const uint32_t c0 = w & 0xffu, c1 = (w >> 8) & 0xffu, c2 = (w >> 16) & 0xffu, c3 = (w >> 24) & 0xffu;
    const bool s0 = ig_is_space(c0) | (n < 1), s1 = ig_is_space(c1) | (n < 2), s2 = ig_is_space(c2) | (n < 3), s3 = ig_is_space(c3) | (n < 4);
    const uint32_t a = !s0 ? 0u : !s1 ? 1u : !s2 ? 2u : !s3 ? 3u : 4u;          // first character that is not a blank
    const uint32_t e = !s3 ? 4u : !s2 ? 3u : !s1 ? 2u : !s0 ? 1u : 0u;          // one past the last
    const uint32_t k = e > a ? e - a : 0u;                                       // characters kept (0 .. 4)
    const uint32_t sh = (a & 3u) * 8u;
    const uint32_t m = k >= 4u ? 0xffffffffu : ((1u << (8u * k)) - 1u);
    return k ? ((w >> sh) & m) : 0u;
}
// fixed_field<W, F> of the host parser on W bytes: right-aligned, F decimals, digits / leading spaces / one '-'
template <int W, int F>
__device__ __forceinline__ bool ig_fixed(const uint8_t* f, float* out) {
    constexpr int IP = W - F - 1;
    if (f[IP] != '.') return false;
    uint32_t frac = 0;
#pragma unroll
    for (int i = 0; i < F; i++) { const uint32_t c = (uint32_t)f[IP + 1 + i] - '0'; if (c > 9u) return false; frac = frac * 10u + c; }
    int i = 0;
    while (i < IP && f[i] == ' ') i++;
    bool neg = false;
    if (i < IP && f[i] == '-') { neg = true; i++; }
    if (i >= IP) return false;
    uint32_t ip = 0;
    for (; i < IP; i++) { const uint32_t c = (uint32_t)f[i] - '0'; if (c > 9u) return false; ip = ip * 10u + c; }
    constexpr uint32_t P10 = F == 3 ? 1000u : (F == 2 ? 100u : 10u);
    constexpr double INV = F == 3 ? 0.001 : (F == 2 ? 0.01 : 0.1);
    const float v = (float)((double)(ip * P10 + frac) * INV);     // == (float)strtod(field) for this layout (proved exhaustively on the host)
    *out = neg ? -v : v;
    return true;
}
// field_int of the host parser on n bytes: spaces around, optional sign, digits only
__device__ __forceinline__ bool ig_int(const uint8_t* f, int n, int32_t* out) {
    int a = 0, e = n;
    while (a < e && f[a] == ' ') a++;
    while (e > a && (f[e - 1] == ' ' || f[e - 1] == '\r')) e--;
    bool neg = false;
    if (a < e && (f[a] == '-' || f[a] == '+')) { neg = f[a] == '-'; a++; }
    if (a >= e) return false;
    long long v = 0;
    for (; a < e; a++) { const uint32_t c = (uint32_t)f[a] - '0'; if (c > 9u) return false; v = v * 10 + (long long)c; }
    *out = (int32_t)(neg ? -v : v);
    return true;
}

// ---- the same field readers on a line held in registers (17 dwords = columns 1-68), every index a compile-time constant:
//      no dependent byte loads (the per-byte global loads of the readers above, each waiting for the compare before it, were
//      what the parse kernel spent its time on) ----
constexpr int IG_LINE_DW = 20;                   // columns 1-80
struct ig_line { uint32_t w[IG_LINE_DW]; };
template <int I> __device__ __forceinline__ uint32_t ig_b(const ig_line& L) { return (L.w[I >> 2] >> (8 * (I & 3))) & 0xffu; }
// right-aligned integer part of a fixed field, characters [A, A + N): spaces, one optional '-', at least one digit
template <int A, int N>
__device__ __forceinline__ bool ig_intpart(const ig_line& L, uint32_t* val, bool* neg) {
    // right to left: digits (at least one), then at most one '-', then blanks. Plain boolean arithmetic: a wavefront's lanes
    // hold lines of every shape, and every `if` here was a branch with its exec-mask bookkeeping
    const uint32_t c0 = ig_b<A + N - 1>(L), c1 = ig_b<A + (N >= 2 ? N - 2 : 0)>(L), c2 = ig_b<A + (N >= 3 ? N - 3 : 0)>(L), c3 = ig_b<A + (N >= 4 ? N - 4 : 0)>(L);
    const uint32_t d0 = c0 - '0', d1 = c1 - '0', d2 = c2 - '0', d3 = c3 - '0';
    const bool g0 = d0 <= 9u, g1 = d1 <= 9u, g2 = d2 <= 9u, g3 = d3 <= 9u;
    const bool r1 = g0, r2 = r1 & g1, r3 = r2 & g2, r4 = r3 & g3;                 // the run of digits reaches 1, 2, 3, 4 characters
    uint32_t v = d0;
    if (N >= 2) v += r2 ? d1 * 10u : 0u;
    if (N >= 3) v += r3 ? d2 * 100u : 0u;
    if (N >= 4) v += r4 ? d3 * 1000u : 0u;
    bool ok = g0, ng = false;
    auto left = [&](bool in_run, bool first, uint32_t c) {          // character left of the rightmost one
        const bool is_neg = c == '-', is_sp = c == ' ';
        ok = ok & (in_run | (first ? (is_neg | is_sp) : is_sp));
        ng = ng | (first & is_neg);
    };
    if (N >= 2) left(r2, r1 & !g1, c1);
    if (N >= 3) left(r3, r2 & !g2, c2);
    if (N >= 4) left(r4, r3 & !g3, c3);
    *val = v; *neg = ng;
    return ok;
}
// fixed_field<8, 3> at column A (coordinates) / fixed_field<6, 2> (B-factor)
template <int A>
__device__ __forceinline__ bool ig_fixed83(const ig_line& L, float* out) {
    const uint32_t d0 = ig_b<A + 5>(L) - '0', d1 = ig_b<A + 6>(L) - '0', d2 = ig_b<A + 7>(L) - '0';
    uint32_t ip; bool neg;
    const bool ok = ig_intpart<A, 4>(L, &ip, &neg) & (ig_b<A + 4>(L) == '.') & (d0 <= 9u) & (d1 <= 9u) & (d2 <= 9u);
    const float v = (float)((double)(ip * 1000u + d0 * 100u + d1 * 10u + d2) * 0.001);
    *out = neg ? -v : v;
    return ok;
}
template <int A>
__device__ __forceinline__ bool ig_fixed62(const ig_line& L, float* out) {
    const uint32_t d0 = ig_b<A + 4>(L) - '0', d1 = ig_b<A + 5>(L) - '0';
    uint32_t ip; bool neg;
    const bool ok = ig_intpart<A, 3>(L, &ip, &neg) & (ig_b<A + 3>(L) == '.') & (d0 <= 9u) & (d1 <= 9u);
    const float v = (float)((double)(ip * 100u + d0 * 10u + d1) * 0.01);
    *out = neg ? -v : v;
    return ok;
}
// field_int on characters [A, A + N), N <= 5: spaces on both sides, optional sign, digits
template <int A, int N>
__device__ __forceinline__ bool ig_int_reg(const ig_line& L, int32_t* out) {
    // blanks, then the number -- an optional sign and digits, at least one --, then blanks or CRs. Boolean arithmetic over the
    // N <= 5 characters (a wavefront's lanes hold fields of every shape): lead[i] = everything up to i is a blank, tail[i] =
    // everything from i on is a blank or a CR, the number is what lies between
    const uint32_t c[5] = {ig_b<A>(L), ig_b<A + (N > 1 ? 1 : 0)>(L), ig_b<A + (N > 2 ? 2 : 0)>(L), ig_b<A + (N > 3 ? 3 : 0)>(L), ig_b<A + (N > 4 ? 4 : 0)>(L)};
    bool lead[5], tail[5];
    bool run = true;
#pragma unroll
    for (int i = 0; i < N; i++) { run = run & (c[i] == ' '); lead[i] = run; }
    run = true;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) { run = run & ((c[i] == ' ') | (c[i] == '\r')); tail[i] = run; }
    bool ok = true, neg = false;
    int32_t v = 0; uint32_t nd = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t d = c[i] - '0';
        const bool dig = d <= 9u, core = !lead[i] & !tail[i], first = core & (i == 0 ? true : lead[i > 0 ? i - 1 : 0]);
        const bool sign = first & ((c[i] == '-') | (c[i] == '+'));
        ok = ok & (!core | sign | dig);
        neg = neg | (sign & (c[i] == '-'));
        const bool take = core & dig;
        v = take ? v * 10 + (int32_t)d : v;
        nd += take ? 1u : 0u;
    }
    *out = neg ? -v : v;
    return ok & (nd > 0u);
}
template <int A, int N>
__device__ __forceinline__ uint32_t ig_pack_reg(const ig_line& L) {      // field_pack of N <= 4 characters at column A
    const uint32_t w = ig_b<A>(L) | (ig_b<A + (N > 1 ? 1 : 0)>(L) << 8) | (ig_b<A + (N > 2 ? 2 : 0)>(L) << 16) | (ig_b<A + (N > 3 ? 3 : 0)>(L) << 24);
    return ig_pack4(w, N);
}

constexpr int IG_BACK = 256;                      // bytes of the previous chunk kept in front of the staged one (a line that ends in a
                                                  // chunk starts at most this far back, or is read from global memory)
struct ingest_lds {
    alignas(16) uint8_t buf[IG_BACK + IG_CHUNK + 80];   // the staged text: every global byte is loaded once, coalesced
    uint32_t line_end[IG_LINES];          // chunk-relative position of every '\n' of the chunk, in order
    uint32_t akey[64], rkey[32];          // open-addressed name -> code tables (packed name, 0 = empty)
    uint8_t aval[64], rval[32];
};
__device__ __forceinline__ uint32_t ig_hash(uint32_t k) { return (k * 0x9E3779B1u) >> 24; }

#ifdef FCZ_IG_TIMING
// measurement aid (not built into the product): wavefront-cycles in the parts of k_ingest_parse
__device__ unsigned long long g_ig_timing[8];
#define IG_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define IG_STAMP(i)
#endif
// ---- k_ingest_parse --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_ingest_parse(const uint8_t* __restrict__ text, const uint64_t* __restrict__ file_off, uint32_t n_files,
                                                       uint64_t text_bytes, const uint64_t* __restrict__ abase, ingest_scratch T,
                                                       uint8_t* __restrict__ titles, uint32_t* __restrict__ title_len,
                                                       uint32_t* __restrict__ n_kept, int32_t* __restrict__ file_status) {
    __shared__ ingest_lds S;
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
    // name -> code tables (the host's NameCodes): open addressing, 64 / 32 slots
    S.akey[lane] = 0; if (lane < 32) S.rkey[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        for (int i = 0; i < FCZ_N_ATOM_CODES; i++) {
            uint32_t k; __builtin_memcpy(&k, fcz_atom_name[i], 4);
            uint32_t h = ig_hash(k) & 63u; while (S.akey[h]) h = (h + 1) & 63u;
            S.akey[h] = k; S.aval[h] = (uint8_t)i;
        }
        for (int i = 0; i < FCZ_N_RES_CODES; i++) {
            uint32_t k; __builtin_memcpy(&k, fcz_res3[i], 4);
            uint32_t h = ig_hash(k) & 31u; while (S.rkey[h]) h = (h + 1) & 31u;
            S.rkey[h] = k; S.rval[h] = (uint8_t)i;
        }
    }
    __builtin_amdgcn_wave_barrier();
    auto atom_code_of = [&](uint32_t k) -> uint32_t {
        for (uint32_t h = ig_hash(k) & 63u; S.akey[h]; h = (h + 1) & 63u) if (S.akey[h] == k) return S.aval[h];
        return (uint32_t)FCZ_ATOM_OTHER;
    };
    auto res_code_of = [&](uint32_t k) -> int {
        for (uint32_t h = ig_hash(k) & 31u; S.rkey[h]; h = (h + 1) & 31u) if (S.rkey[h] == k) { const int c = S.rval[h]; return (c < 20 || c == 23) ? c : -1; }
        return -1;
    };

#ifdef FCZ_IG_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    const uint64_t f0 = file_off[f], f1 = file_off[f + 1];
    const uint8_t* base = text + f0;
    const uint64_t flen = f1 - f0;
    const uint64_t A0 = abase[f];
    uint8_t* tbuf = titles + (size_t)f * IG_TITLE_CAP;
    uint32_t kept = 0;                     // atoms written so far (uniform)
    uint32_t last_name = 0; bool have_last = false;       // the last ATOM / HETATM record seen (uniform): its atom name,
    int32_t last_seq = 0; uint32_t last_rn = 0, last_seg = 0, last_icch = 0;   // its residue (number, name, segment, insertion code | chain << 8)
    uint32_t tlen = 0, hdr_id = 0, hdr_n = 0;             // title state (uniform): TITLE text so far, the last HEADER id code
    bool ended = false;                                   // an END record was read
    // MODEL / ENDMDL records (uniform): where the reader stands -- 0 nothing read, 1 atoms without a MODEL record, 2 a model is open,
    // 3 the last model was closed --, whether the open model has atoms, the last MODEL number (-2: none yet, -1: not a plain number),
    // and whether a MODEL / ENDMDL record came after the last ATOM / HETATM record (the next one starts a new chain whatever its name)
    int mstate = 0; bool open_has_atoms = false; int32_t last_mnum = -2; bool boundary_since_rec = false;
    bool last_line_rec = false;                           // the last line read was an ATOM / HETATM record (an ANISOU may follow it)
    int32_t status = FCZ_OK;
    uint64_t line_start = 0;               // file-relative start of the line that is open at the chunk's beginning (uniform)

    // one line [ls, le) (file-relative, le excludes the newline) per lane; `on` = this lane has a line
    // One line [ls, le) (file-relative, le excludes the line end) per lane; `on` = this lane has a line; lo = offset of its first
    // byte in S.buf, or -1 when it is not staged (it started more than IG_BACK before the chunk); has_nl = a line end follows it.
    // The rules are gemmi's (lib/gemmi/pdb.hpp:262-365, restated in foldcomp_amd/structure.py parse_pdb_gemmi): records are
    // matched on four letters case-insensitively, END stops the reading, `len` below is the reader's line length (line end
    // included, at most 120). What the fixed-column fast path cannot promise to read as that reader would -- a field outside the
    // fixed layout, a two-character chain name, an ANISOU record that does not follow its atom, models that are not numbered upwards, a residue whose lines are apart
    // (the reader regroups them)
    // -- marks the file for the host, which implements every rule.
    auto do_lines = [&](bool on, uint64_t ls, uint64_t le, int lo, bool has_nl) {
        if (ended) return;
        bool cryst_bad = false;
        const uint32_t raw = on ? (uint32_t)(le - ls) : 0u;
        const uint32_t glen = raw + (has_nl ? 1u : 0u) < 120u ? raw + (has_nl ? 1u : 0u) : 120u;       // gemmi's len
        const uint8_t* p = base + ls;
        // the staged copy serves every read of the usual line (no global load the parse has to wait for)
        const bool staged = on && lo >= 0 && (uint32_t)lo + 80u <= (uint32_t)sizeof(S.buf) && (uint32_t)lo + raw <= (uint32_t)sizeof(S.buf);
        uint32_t len = raw;
        if (on && len && (staged ? S.buf[lo + len - 1] : p[len - 1]) == '\r') len--;
        ig_line L;
        if (staged) {
#pragma unroll
            for (int i = 0; i < IG_LINE_DW; i++) __builtin_memcpy(&L.w[i], &S.buf[lo + 4 * i], 4);
        } else {
#pragma unroll
            for (int i = 0; i < IG_LINE_DW; i++) L.w[i] = 0;
            if (on && raw >= 4) L.w[0] = ld_u32(p);
            if (on && raw >= 8) L.w[1] = ld_u32(p + 4);
        }
        // the record name as the reader compares it: four characters, case folded, nothing beyond the line (a short last line)
        const uint32_t nvis = raw + (has_nl ? 1u : 0u);
        const uint32_t w0 = on ? (nvis >= 4 ? L.w[0] : (L.w[0] & ((1u << (8 * nvis)) - 1u))) : 0u;
        const uint32_t up4 = w0 & ~0x20202020u;
        const bool rec = on && (up4 == (0x4d4f5441u & ~0x20202020u) || up4 == (0x41544548u & ~0x20202020u));     // ATOM, HETA(TM)
        // a step of nothing but ATOM / HETATM lines (all but two or three steps of a file) skips what the other records need
        const bool any_other = __any(on && !rec);
        bool is_end = false, foreign = false, is_model = false, is_endm = false, is_anis = false; int32_t mnum = -1;
        if (any_other) {
            is_end = on && !rec && (up4 & 0x00ffffffu) == 0x00444e45u && ((up4 >> 24) & 0xf0u) == 0u;          // END, not ENDMDL
            // records this path does not model: data_ / {"data_ (another format)
            foreign = on && !rec && ((up4 == (0x61746164u & ~0x20202020u) && (L.w[1] & 0xffu) == '_') ||
                                                (up4 == (0x6164227bu & ~0x20202020u) && ((L.w[1] & 0x00ffffffu) & ~0x00202020u) == (0x005f6174u & ~0x00202020u)));   // data_, {"data_
            // ANISOU: the reader attaches it to the last atom read of the current residue and fails the file when there is none or when
            // that atom already has one with a non-zero U11 (lib/gemmi/pdb.hpp:262-365). A record that stands directly behind an ATOM /
            // HETATM line -- where every file of the archive has it -- finds that fresh atom: nothing to decide. Any other place: the host's.
            is_anis = on && !rec && up4 == (0x53494e41u & ~0x20202020u);
            // MODEL / ENDMDL: one MODEL record before the first atom and ENDMDL records after the last one (the single-model file every
            // predicted structure is) change nothing. Ensembles (NMR entries of the archive): MODEL n / atoms / ENDMDL, again and again.
            // The reader (lib/gemmi/pdb.hpp:262-365) keeps the models in file order, names them by the number in columns 11-14, fails
            // a MODEL record that finds atoms in an open model or a name that already has atoms, names atoms that follow an ENDMDL
            // without a MODEL record by the count of models, and starts a new chain at every MODEL / ENDMDL whatever the chain's name.
            // Here: every MODEL after the first carries a plain number larger than the one before it (no name comes twice), stands
            // where no model is open, and atoms stand inside a model or before any -- then file order is the reader's order and the
            // only thing a model boundary changes is that the residue-order rule below does not look across it. Anything else: the host's.
            is_model = on && !rec && up4 == (0x45444f4du & ~0x20202020u);
            is_endm = on && !rec && up4 == (0x4d444e45u & ~0x20202020u);
            if (is_model && staged && len >= 14u) {
                const uint32_t c[4] = {ig_b<10>(L), ig_b<11>(L), ig_b<12>(L), ig_b<13>(L)};
                int i = 0; while (i < 4 && c[i] == ' ') i++;
                bool ok = i < 4; int32_t v = 0;
                for (; i < 4; i++) { ok = ok && (c[i] - '0' < 10u); v = v * 10 + (int32_t)(c[i] - '0'); }
                mnum = ok ? v : -1;
            }
            // CRYST1: the reader fails a file whose cell has a gamma and an alpha or beta of exactly zero (UnitCell::set). Decided here
            // only for fields that start (after blanks) with a digit 1-9 -- certainly not zero; anything else goes to the host
            if (on && !rec && up4 == (0x53595243u & ~0x20202020u) && glen > 54u) {
                bool sure = staged;
                if (staged) {
#pragma unroll
                    for (int f0 = 33; f0 <= 40; f0 += 7) {
                        uint32_t first = ' ';
#pragma unroll
                        for (int q = 6; q >= 0; q--) { const uint32_t c = (uint32_t)S.buf[lo + f0 + q]; if (c != ' ') first = c; }
                        if (!(first - '1' < 9u)) sure = false;
                    }
                }
                if (!sure) cryst_bad = true;
            }
        }
        const unsigned long long m_end = __ballot(is_end);
        const int end_lane = m_end ? __builtin_ctzll(m_end) : 64;
        const bool live = on && lane < end_lane;                 // what follows an END record is not read
        if (__any(live && (foreign || cryst_bad))) status = FCZ_INGEST_HOST_FIELD;
        {
            const unsigned long long m_live = __ballot(live), m_recs = __ballot(rec && live);
            if (any_other) {
                const bool prev_rec = lane ? ((m_recs >> (lane - 1)) & 1ull) != 0 : last_line_rec;
                if (__any(live && is_anis && !prev_rec)) status = FCZ_INGEST_HOST_FIELD;
            }
            if (m_live) last_line_rec = ((m_recs >> (63 - __builtin_clzll(m_live))) & 1ull) != 0;
        }
        const unsigned long long m_model = __ballot(live && is_model), m_endm = __ballot(live && is_endm);
        // ---- title: the last HEADER record's id code (columns 63-66, right-trimmed), else the TITLE records' text concatenated ----
        if (any_other) {
            const bool is_title = live && !rec && up4 == (0x4c544954u & ~0x20202020u) && glen > 10;
            bool is_hdr = false; uint32_t hid = 0, hn = 0;
            if (live && !rec && up4 == (0x44414548u & ~0x20202020u) && glen > 66) {
                if (staged) {
                    uint32_t c[4] = {ig_b<62>(L), ig_b<63>(L), ig_b<64>(L), ig_b<65>(L)};
                    hn = 4; while (hn && (c[hn - 1] == ' ' || c[hn - 1] == '\r' || c[hn - 1] == '\n' || c[hn - 1] == '\t')) hn--;
                    for (uint32_t i = 0; i < hn; i++) hid |= c[i] << (8 * i);
                    is_hdr = hn != 0;
                } else status = FCZ_INGEST_HOST_TITLE;
            }
            unsigned long long m_t = __ballot(is_title || is_hdr);
            while (m_t) {
                const int l = __builtin_ctzll(m_t); m_t &= m_t - 1;
                const bool hdr = __shfl((int)is_hdr, l, WAVE) != 0;
                if (hdr) { hdr_id = (uint32_t)__shfl((int)hid, l, WAVE); hdr_n = (uint32_t)__shfl((int)hn, l, WAVE); continue; }
                uint32_t add = 0;
                if (lane == l) {
                    int e = (int)glen - 1;                        // the reader drops the line's last character (its line end)
                    auto at_ = [&](int i) -> uint32_t { return staged ? (uint32_t)S.buf[lo + i] : (uint32_t)p[i]; };
                    while (e > 10) { const uint32_t c = at_(e - 1); if (c == ' ' || c == '\r' || c == '\n' || c == '\t') e--; else break; }
                    uint32_t at = tlen;
                    for (int i = 10; i < e; i++, at++) if (at < (uint32_t)IG_TITLE_CAP) tbuf[at] = (uint8_t)at_(i);
                    add = at;
                }
                tlen = (uint32_t)__shfl((int)add, l, WAVE);
            }
        }
        // ---- ATOM / HETATM records ----
        const bool arec = rec && live;
        uint32_t an = 0, rn = 0, seg = 0, icode = 0; int32_t serial = 0, resseq = 0; float x = 0.f, y = 0.f, z = 0.f, bf = 0.f; uint32_t ch = ' ';
        bool bad = false;
        if (arec) {
            // the coordinate columns must all be there (the reader fails a file with a shorter record: the host reports it)
            if (!staged || len < 54 || glen < 55) bad = true;
            else {
                an = ig_pack_reg<12, 4>(L);
                rn = ig_pack_reg<17, 3>(L);
                ch = ig_b<21>(L);
                bad = bad | (ig_b<20>(L) != ' ') | ((ch != ' ') & ig_is_space(ch));          // a two-character chain name
                icode = ig_b<26>(L);
                const bool i1 = ig_int_reg<6, 5>(L, &serial), i2 = ig_int_reg<22, 4>(L, &resseq);
                const bool f1 = ig_fixed83<30>(L, &x), f2 = ig_fixed83<38>(L, &y), f3 = ig_fixed83<46>(L, &z);
                bad = bad | !(i1 & i2 & f1 & f2 & f3);
                // B-factor: 20 when the line ends before column 65 (the reader's default); the fixed layout; six blanks = 0
                {
                    float bfx;
                    const bool fb = ig_fixed62<60>(L, &bfx) & (len >= 66);
                    const bool blank = (len >= 66) & ig_is_space(ig_b<60>(L)) & ig_is_space(ig_b<61>(L)) & ig_is_space(ig_b<62>(L)) & ig_is_space(ig_b<63>(L)) &
                                       ig_is_space(ig_b<64>(L)) & ig_is_space(ig_b<65>(L));
                    bf = glen <= 64 ? 20.0f : (fb ? bfx : 0.f);
                    bad = bad | ((glen > 64) & !fb & !blank);
                }
                // charge (columns 79-80, read_charge lib/gemmi/pdb.hpp:85-98): a digit there needs a sign (or nothing) beside it,
                // the reader fails the file otherwise
                {
                    uint32_t dg = raw > 78u ? ig_b<78>(L) : (uint32_t)'\n', sg = raw > 79u ? ig_b<79>(L) : (((raw == 79u) & has_nl) ? (uint32_t)'\n' : 0u);
                    const bool swap = sg - '0' < 10u;
                    const uint32_t dg2 = swap ? sg : dg, sg2 = swap ? dg : sg;
                    const bool sign_ok = (sg2 == '+') | (sg2 == '-') | (sg2 == 0u) | ig_is_space(sg2);
                    bad = bad | ((glen > 78) & !((dg == ' ') & (sg == ' ')) & (dg2 - '0' < 10u) & !sign_ok);
                }
                // segment id (columns 73-76) is part of the residue's identity when the line reaches it: its characters up to the
                // line end (or a CR / NUL), stripped
                {
                    const uint32_t c0 = ig_b<72>(L), c1 = ig_b<73>(L), c2 = ig_b<74>(L), c3 = ig_b<75>(L);
                    auto term = [](uint32_t c) { return (c == '\r') | (c == 0u); };
                    const bool v0 = (glen > 72) & (raw > 72u) & !term(c0), v1 = v0 & (raw > 73u) & !term(c1), v2 = v1 & (raw > 74u) & !term(c2), v3 = v2 & (raw > 75u) & !term(c3);
                    const uint32_t e0 = v0 ? c0 : ' ', e1 = v1 ? c1 : ' ', e2 = v2 ? c2 : ' ', e3 = v3 ? c3 : ' ';
                    seg = ig_pack4(e0 | (e1 << 8) | (e2 << 16) | (e3 << 24), 4);
                }
            }
        }
        // the reader gathers the atoms of a residue (number, insertion code, name, segment) wherever their lines stand inside a run
        // of lines with one chain name. File order is its order as long as every new residue of a run has a larger (number, insertion
        // code) than the one before it; anything else goes to the host, which regroups.
        const unsigned long long m_rec = __ballot(arec);
        const uint32_t pl = ig_prev_lane(m_rec, lane);
        const unsigned long long m_ev = m_model | m_endm;
        bool brk = pl < 64u ? false : boundary_since_rec;     // a MODEL / ENDMDL record between this record and the one before it
        if (m_ev) {
            const unsigned long long below = (1ull << lane) - 1ull, evb = m_ev & below;
            // where the reader stands when it comes to this lane's line
            int st = mstate;
            if (evb) st = ((m_model >> (63 - __builtin_clzll(evb))) & 1ull) ? 2 : 3;
            else if (st == 0 && (m_rec & below)) st = 1;
            const unsigned long long mb = m_model & below;
            const int32_t s_mnum = __shfl(mnum, mb ? 63 - __builtin_clzll(mb) : 0, WAVE);      // (outside every condition)
            const int32_t prev_mnum = mb ? s_mnum : last_mnum;
            if (arec && st == 3) bad = true;                                          // atoms behind an ENDMDL: a model named by the count
            if (live && is_model) {
                if (!(st == 0 || st == 3)) bad = true;                                // atoms without a model before it, or a model still open
                if (prev_mnum != -2 && !(prev_mnum >= 0 && mnum > prev_mnum)) bad = true;   // (the first MODEL record may say anything ...
                if (prev_mnum == -2 && (have_last || (m_rec & below))) bad = true;          //  ... but stands before every atom: atoms without
            }                                                                                //  one are the model "1" already)
            brk = pl < 64u ? (evb & ~((2ull << pl) - 1ull)) != 0 : (boundary_since_rec || evb != 0);
        } else if (arec && mstate == 3) bad = true;
        {
            const int src = pl < 64u ? (int)pl : 0;
            // (the shuffles stand outside every condition: a lane that sits one out would hand its neighbour nothing)
            const int32_t s_seq = __shfl(resseq, src, WAVE);
            const uint32_t s_rn = (uint32_t)__shfl((int)rn, src, WAVE), s_seg = (uint32_t)__shfl((int)seg, src, WAVE);
            const uint32_t s_ic = (uint32_t)__shfl((int)(icode | (ch << 8)), src, WAVE);
            const int32_t p_seq = pl < 64u ? s_seq : last_seq;
            const uint32_t p_rn = pl < 64u ? s_rn : last_rn;
            const uint32_t p_seg = pl < 64u ? s_seg : last_seg;
            const uint32_t p_ic = pl < 64u ? s_ic : last_icch;
            const bool has_p = pl < 64u ? true : have_last;
            if (arec && !bad && has_p && !brk && (p_ic >> 8) == ch) {
                const bool same_res = p_seq == resseq && p_rn == rn && p_seg == seg && (p_ic & 0xffu) == icode;
                if (!same_res && !(resseq > p_seq || (resseq == p_seq && icode > (p_ic & 0xffu)))) bad = true;
            }
        }
        if (__any(bad)) status = FCZ_INGEST_HOST_FIELD;
        // keep rule. removeAlternativePosition drops an atom whose name equals the name of the last atom KEPT; that is the
        // same as "equals the name of the record right before it": if that record was kept it is the comparison itself, if it
        // was dropped it carried the kept atom's name.
        const uint32_t pname = (uint32_t)__shfl((int)an, pl < 64u ? (int)pl : 0, WAVE);
        const bool has_prev = pl < 64u ? true : have_last;
        const uint32_t prev_name = pl < 64u ? pname : last_name;
        const bool keep = arec && !bad && !(has_prev && prev_name == an);   // (a bad record marks the file for the host: nothing of it is used)
        const unsigned long long m_keep = __ballot(keep);
        if (keep) {
            const size_t o = (size_t)A0 + kept + (uint32_t)__builtin_popcountll(m_keep & ((1ull << lane) - 1ull));
            T.name[o] = an; T.resn[o] = rn; T.serial[o] = serial; T.resseq[o] = resseq;
            T.x[o] = x; T.y[o] = y; T.z[o] = z; T.b[o] = bf; T.chain[o] = (uint32_t)ch & 0xffu;
            T.acode[o] = (uint8_t)atom_code_of(an);
            T.rcode[o] = (int8_t)res_code_of(rn);
        }
        kept += (uint32_t)__builtin_popcountll(m_keep);
        if (m_ev) {
            const int t = 63 - __builtin_clzll(m_ev);
            mstate = ((m_model >> t) & 1ull) ? 2 : 3;
            open_has_atoms = ((m_rec >> t) >> 1) != 0;
            if (m_model) last_mnum = __shfl(mnum, 63 - __builtin_clzll(m_model), WAVE);
            boundary_since_rec = ((m_ev >> (m_rec ? 63 - __builtin_clzll(m_rec) : 0)) >> (m_rec ? 1 : 0)) != 0;
        } else if (m_rec) { if (mstate == 0) mstate = 1; open_has_atoms = true; boundary_since_rec = false; }
        if (m_rec) {
            const int hl = 63 - __builtin_clzll(m_rec);
            last_name = (uint32_t)__shfl((int)an, hl, WAVE); have_last = true;
            last_seq = __shfl(resseq, hl, WAVE); last_rn = (uint32_t)__shfl((int)rn, hl, WAVE); last_seg = (uint32_t)__shfl((int)seg, hl, WAVE);
            last_icch = (uint32_t)__shfl((int)(icode | (ch << 8)), hl, WAVE);
        }
        if (m_end) ended = true;
    };

    uint64_t c0_staged = 0;
    // a chunk's dwords, dword (d * 64 + lane) in pre[d]: 256 contiguous bytes per load instruction; zero past the end of the file
    uint32_t pre[16];
    auto load_chunk = [&](uint64_t cc) {
#pragma unroll
        for (int d = 0; d < 16; d++) {
            const uint64_t q = cc + 4ull * (uint64_t)(d * WAVE + lane);
            uint32_t v = 0;
            if (q + 4 <= flen) v = ld_u32(base + q);
            else for (int b = 0; b < 4; b++) if (q + b < flen) v |= (uint32_t)base[q + b] << (8 * b);
            pre[d] = v;
        }
    };
    load_chunk(0);
    IG_STAMP(0)
    // (a file that is marked for the host is not read any further: nothing of it is used)
    for (uint64_t c0 = 0; c0 < flen && !ended && status == FCZ_OK; c0 += IG_CHUNK) {
        // ---- stage the chunk: the tail of the previous window moves to the front, then the chunk's 16 coalesced dwords per lane,
        //      which were requested while the previous chunk was being parsed ----
        {
            uint32_t tail;
            __builtin_memcpy(&tail, &S.buf[IG_CHUNK + 4 * lane], 4);          // = window bytes [IG_BACK + IG_CHUNK - 256 ...)
            __builtin_amdgcn_wave_barrier();
            __builtin_memcpy(&S.buf[4 * lane], &tail, 4);
#pragma unroll
            for (int d = 0; d < 16; d++) __builtin_memcpy(&S.buf[IG_BACK + 4 * (d * WAVE + lane)], &pre[d], 4);
            __builtin_amdgcn_wave_barrier();
            c0_staged = c0;
            load_chunk(c0 + IG_CHUNK);                                         // the next chunk is on its way
        }
        IG_STAMP(1)
        // ---- line ends of the chunk: every lane looks at its 64 staged bytes ----
        const uint64_t my = c0 + 64ull * (uint64_t)lane;     // file-relative start of this lane's 64 bytes
        // one bit per byte: the 0x80 flags of a dword's four bytes are gathered into a nibble by one multiplication
        // ((m >> 7) has bits 0, 8, 16, 24; times 2^28 + 2^21 + 2^14 + 2^7 they land on bits 28..31 without a carry)
        uint32_t nl_lo = 0, nl_hi = 0, z_lo = 0, z_hi = 0;
        {
            const uint4* src = reinterpret_cast<const uint4*>(&S.buf[IG_BACK + 64 * lane]);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const uint4 v4 = src[q4];
                const uint32_t vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = 4 * q4 + k;
                    const uint32_t xz = vv[k] ^ 0x0a0a0a0au;
                    const uint32_t m = ~(((xz & 0x7f7f7f7fu) + 0x7f7f7f7fu) | xz | 0x7f7f7f7fu);     // 0x80 in every byte that is '\n'
                    // a NUL inside the file ends the reader's line there (C strings): not this path's business
                    const uint32_t zz = ~(((vv[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | vv[k] | 0x7f7f7f7fu);
                    const uint32_t nn = ((m >> 7) * 0x10204080u) >> 28, nz = ((zz >> 7) * 0x10204080u) >> 28;
                    if (d < 8) { nl_lo |= nn << (4 * d); z_lo |= nz << (4 * d); } else { nl_hi |= nn << (4 * (d - 8)); z_hi |= nz << (4 * (d - 8)); }
                }
            }
        }
        // bytes past the end of the file were staged as zero: neither line ends nor NULs of the file
        const uint32_t vb = my >= flen ? 0u : (flen - my >= 64u ? 64u : (uint32_t)(flen - my));
        const unsigned long long vmask = vb >= 64u ? ~0ull : ((1ull << vb) - 1ull);
        const unsigned long long nlm = (((unsigned long long)nl_hi << 32) | nl_lo) & vmask;
        const uint32_t nul = ((((unsigned long long)z_hi << 32) | z_lo) & vmask) != 0ull ? 1u : 0u;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(nlm);
        if (__any(nul != 0u)) status = FCZ_INGEST_HOST_FIELD;
        uint32_t total;
        uint32_t ord = wave_excl_scan_dpp(cnt, &total);
        IG_STAMP(2)
        // chunks with more line ends than the table holds (blank-line runs) go through it in rounds
        for (uint32_t r0 = 0; r0 < total || r0 == 0; r0 += IG_LINES) {
            uint32_t o = ord;
            for (unsigned long long m = nlm; m; m &= m - 1) {        // a lane's 64 bytes hold one line end or two (81-byte records), rarely more
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                if (o >= r0 && o < r0 + (uint32_t)IG_LINES) S.line_end[o - r0] = 64u * (uint32_t)lane + bit;
                o++;
            }
            __builtin_amdgcn_wave_barrier();
            IG_STAMP(3)
            const uint32_t n_here = total - r0 < (uint32_t)IG_LINES ? total - r0 : (uint32_t)IG_LINES;
            for (uint32_t k0 = 0; k0 < n_here; k0 += WAVE) {
                const uint32_t k = k0 + (uint32_t)lane;
                const bool on = k < n_here;
                const uint64_t le = on ? c0 + S.line_end[k] : 0;
                const uint64_t ls = on ? (k == 0 ? line_start : c0 + S.line_end[k - 1] + 1) : 0;
                const long long rel = (long long)ls - (long long)c0;                 // >= -IG_BACK: the line's start is staged
                do_lines(on, ls, le, (on && rel >= -(long long)IG_BACK) ? (int)(IG_BACK + rel) : -1, true);
            }
            if (n_here) line_start = c0 + S.line_end[n_here - 1] + 1;
            __builtin_amdgcn_wave_barrier();
            IG_STAMP(4)
            if (total == 0) break;
        }
    }
    if (line_start < flen && !ended) {     // a last line without a line end
        const long long rel = (long long)line_start - (long long)c0_staged;
        do_lines(lane == 0, line_start, flen, rel >= -(long long)IG_BACK ? (int)(IG_BACK + rel) : -1, false);
    }
    if (lane == 0) {
        // title = the HEADER id code, else the concatenated TITLE text (as the reader leaves it: nothing stripped)
        if (tlen > (uint32_t)IG_TITLE_CAP) { if (status == FCZ_OK) status = FCZ_INGEST_HOST_TITLE; tlen = 0; }
        if (hdr_n) { for (uint32_t i = 0; i < hdr_n; i++) tbuf[i] = (uint8_t)(hdr_id >> (8 * i)); tlen = hdr_n; }
        title_len[f] = tlen;
        if (status == FCZ_OK && kept == 0) status = FCZ_INGEST_NO_ATOMS;
        n_kept[f] = status == FCZ_OK ? kept : 0u;
        file_status[f] = status;
    }
#ifdef FCZ_IG_TIMING
    IG_STAMP(5)
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_ig_timing[i], tacc[i]);
#endif
}

// ---- k_ingest_frags -----------------------------------------------------------------------------------------------------
// Chain cuts (identifyChains): walking the atoms, where the chain id changes -- if the atom there is an N a new chain starts;
// otherwise the chain ends there, the next one starts at the following N (atoms in between belong to nobody) and the walk
// resumes after it; no N any more: no further cuts. Gap cuts (identifyDiscontinousResInd): inside a chain, from its first N;
// a new fragment starts at an N whose residue number exceeds the previous N's by more than one.
// Residues of a fragment (splitAtomByResidue as the hosts restate it): a new residue starts where the residue number changes,
// except at the fragment's last atom.
struct ingest_totals { uint32_t chains, residues, atoms, title_bytes; };
struct ingest_counts { uint32_t *chains, *residues, *atoms, *title_bytes; };   // per file, scanned into the batch offsets
__device__ __forceinline__ uint32_t ig_ld_coherent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(WAVE) void k_ingest_frags(uint32_t n_files, const uint64_t* __restrict__ abase, ingest_scratch T,
                                                       const uint32_t* __restrict__ n_kept, int32_t* __restrict__ file_status,
                                                       const uint32_t* __restrict__ title_len, const char* __restrict__ names,
                                                       const uint32_t* __restrict__ name_off, const uint32_t* __restrict__ stem_len,
                                                       const uint8_t* __restrict__ titles, int anchor_threshold, int skip_discontinuous,
                                                       ingest_frag* __restrict__ frags, uint32_t* __restrict__ n_frags,
                                                       ingest_counts totals, uint32_t* __restrict__ use_stem) {
    __shared__ uint32_t s_cut[2 * IG_MAX_FRAGS + 2];     // chain ranges as (a, b) pairs
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
    const uint32_t n = n_kept[f];
    ingest_totals tot{0, 0, 0, 0};
    uint32_t nf = 0;
    if (file_status[f] != FCZ_OK || n == 0) {
        if (lane == 0) { n_frags[f] = 0; totals.chains[f] = 0; totals.residues[f] = 0; totals.atoms[f] = 0; totals.title_bytes[f] = 0; use_stem[f] = 0; }
        return;
    }
    const size_t A0 = (size_t)abase[f];
    const uint32_t PKN = 'N';
    constexpr int FR_U = 4;                               // steps of 64 atoms whose loads are in flight together
    // ---- chain ranges ----
    uint32_t n_ch = 0; bool overflow = false;
    {
        uint32_t start = 0, resume = 1; bool stopped = false;
        auto push = [&](uint32_t a, uint32_t b) { if (n_ch < (uint32_t)IG_MAX_FRAGS) { if (lane == 0) { s_cut[2 * n_ch] = a; s_cut[2 * n_ch + 1] = b; } n_ch++; } else overflow = true; };
        // (this kernel walks a file's atoms three times in steps of 64 and every step waits for its loads: four steps' loads are
        // issued together)
        for (uint32_t ib = 0; ib < n && !stopped; ib += FR_U * WAVE) {
          uint32_t cc[FR_U], cp[FR_U];
#pragma unroll
          for (int u = 0; u < FR_U; u++) {
              const uint32_t i = ib + (uint32_t)(u * WAVE + lane);
              const bool in = i < n && i >= 1;
              cc[u] = in ? (uint32_t)T.chain[A0 + i] : 0u; cp[u] = in ? (uint32_t)T.chain[A0 + i - 1] : 0u;
          }
#pragma unroll
          for (int u = 0; u < FR_U; u++) {
            const uint32_t i0 = ib + (uint32_t)(u * WAVE);
            if (i0 >= n || stopped) break;
            unsigned long long m = __ballot(cc[u] != cp[u]);
            while (m && !stopped) {
                const int l = __builtin_ctzll(m); m &= m - 1;
                const uint32_t ci = i0 + (uint32_t)l;
                if (ci < resume) continue;
                if (T.name[A0 + ci] == PKN) { push(start, ci); start = ci; resume = ci + 1; continue; }
                // the next N at or after ci
                uint32_t j = n;
                for (uint32_t j0 = ci & ~63u; j0 < n && j == n; j0 += WAVE) {
                    const uint32_t q = j0 + (uint32_t)lane;
                    const unsigned long long mn = __ballot(q < n && q >= ci && T.name[A0 + q] == PKN);
                    if (mn) j = j0 + (uint32_t)__builtin_ctzll(mn);
                }
                if (j == n) { stopped = true; break; }
                push(start, ci); start = j; resume = j + 1;
            }
          }
        }
        push(start, n);
    }
    __builtin_amdgcn_wave_barrier();
    const bool multi_chain = n_ch > 1;
    // ---- fragments of every chain; residues and their validation ----
    ingest_frag* my_frags = frags + (size_t)f * IG_MAX_FRAGS;
    uint32_t r_next = 0;                                   // next free slot of the file's residue scratch
    for (uint32_t c = 0; c < n_ch && !overflow; c++) {
        const uint32_t ca = s_cut[2 * c], cb = s_cut[2 * c + 1];
        const uint32_t chain_char = T.chain[A0 + ca] & 0xffu;
        // gap cuts: positions (all N atoms) where a new fragment starts
        uint32_t fstart[IG_MAX_FRAGS + 1]; uint32_t nfr = 0;    // uniform values, small
        {
            bool have_n = false; int32_t prev_seq = 0; uint32_t cur = 0;
            for (uint32_t ib = ca; ib < cb; ib += FR_U * WAVE) {
              uint32_t nm[FR_U]; int32_t sq[FR_U];
#pragma unroll
              for (int u = 0; u < FR_U; u++) {
                  const uint32_t i = ib + (uint32_t)(u * WAVE + lane);
                  nm[u] = i < cb ? T.name[A0 + i] : 0u; sq[u] = i < cb ? T.resseq[A0 + i] : 0;
              }
#pragma unroll
              for (int u = 0; u < FR_U; u++) {
                const uint32_t i0 = ib + (uint32_t)(u * WAVE);
                if (i0 >= cb) break;
                const uint32_t i = i0 + (uint32_t)lane;
                const bool isn = i < cb && nm[u] == PKN;
                const int32_t seq = isn ? sq[u] : 0;
                const unsigned long long mn = __ballot(isn);
                if (!mn) continue;
                const uint32_t pl = ig_prev_lane(mn, lane);
                const int32_t pseq = __shfl(seq, pl < 64u ? (int)pl : 0, WAVE);
                const bool hp = pl < 64u ? true : have_n;
                const int32_t ps = pl < 64u ? pseq : prev_seq;
                const bool first_n = isn && !hp;
                const bool cut = isn && hp && (seq - ps > 1);
                unsigned long long mc = __ballot(cut || first_n);
                while (mc) {
                    const int l = __builtin_ctzll(mc); mc &= mc - 1;
                    const uint32_t pos = i0 + (uint32_t)l;
                    if (nfr < (uint32_t)IG_MAX_FRAGS) fstart[nfr++] = pos; else overflow = true;
                    cur = pos;
                }
                (void)cur;
                const int hl = 63 - __builtin_clzll(mn);
                prev_seq = __shfl(seq, hl, WAVE); have_n = true;
              }
            }
        }
        if (nfr == 0 || overflow) continue;                // a chain without an N atom has no fragment
        fstart[nfr] = cb;
        const bool multi_frag = nfr > 1;
        for (uint32_t j = 0; j < nfr; j++) {
            const uint32_t a = fstart[j], b = fstart[j + 1];
            uint32_t reason = (skip_discontinuous && multi_frag) ? IG_REF_SKIP_DISC : IG_REF_NONE;
            // residue starts: atom a, and every atom in (a, b - 1) whose residue number differs from its predecessor's
            uint32_t nres = 0;
            for (uint32_t ib = a; ib < b; ib += FR_U * WAVE) {
                int32_t rq[FR_U], rp[FR_U];
#pragma unroll
                for (int u = 0; u < FR_U; u++) {
                    const uint32_t i = ib + (uint32_t)(u * WAVE + lane);
                    const bool in = i < b && i > a;
                    rq[u] = in ? T.resseq[A0 + i] : 0; rp[u] = in ? T.resseq[A0 + i - 1] : 0;
                }
#pragma unroll
                for (int u = 0; u < FR_U; u++) {
                    const uint32_t i = ib + (uint32_t)(u * WAVE + lane);
                    const bool st = i < b && (i == a || (i != b - 1 && rq[u] != rp[u]));
                    const unsigned long long ms = __ballot(st);
                    if (st) T.r_first[A0 + r_next + nres + (uint32_t)__builtin_popcountll(ms & ((1ull << lane) - 1ull))] = i;
                    nres += (uint32_t)__builtin_popcountll(ms);
                }
            }
            __threadfence();                                 // the residue starts are read back by other lanes below
            // the FCZ header holds nResidue in 16 bits and nAnchor in 8 (src/foldcomp.h:120-125): refused, not wrapped
            if (nres > 65535u || (anchor_threshold > 0 && nres / (uint32_t)anchor_threshold + 2u > 255u)) { if (!reason) reason = IG_REF_TOO_LONG; }
            // lane = residue: code of the residue name, first N / CA / C in order, the CA B-factor
            // (a second N, CA or C in a residue: the reference counts residues on the flat list of those atoms, it would shift all
            // that follows; the last atom's residue name: it is header.lastResidue, src/foldcomp.cpp:469 -- both refused, see the
            // hosts' build_batch / Batch::prepare)
            bool bad_name = false, bad_bb = false, bad_twice = false;
            for (uint32_t q0 = 0; q0 < nres; q0 += WAVE) {
                const uint32_t q = q0 + (uint32_t)lane;
                if (q < nres) {
                    const uint32_t s0 = ig_ld_coherent(&T.r_first[A0 + r_next + q]);
                    const uint32_t s1 = q + 1 < nres ? ig_ld_coherent(&T.r_first[A0 + r_next + q + 1]) : b;
                    const int code = T.rcode[A0 + s0];
                    if (code < 0) bad_name = true;
                    int pos[3] = {-1, -1, -1}; uint32_t cnt = 0;
                    for (uint32_t i = s0; i < s1; i++) { const uint32_t ac = T.acode[A0 + i]; if (ac < 3u) { cnt++; if (pos[ac] < 0) pos[ac] = (int)i; } }
                    if (pos[0] < 0 || pos[1] < 0 || pos[2] < 0 || !(pos[0] < pos[1] && pos[1] < pos[2])) bad_bb = true;
                    else if (cnt != 3u) bad_twice = true;
                    T.r_code[A0 + r_next + q] = (uint8_t)(code < 0 ? 23 : code);
                    T.r_bfac[A0 + r_next + q] = pos[1] >= 0 ? T.b[A0 + (uint32_t)pos[1]] : 0.f;
                }
            }
            // the host reports the first problem it meets walking the residues; a fragment with either is refused all the same
            if (!reason && __any(bad_name)) reason = IG_REF_RESNAME;
            if (!reason && __any(bad_bb)) reason = IG_REF_BACKBONE;
            if (!reason && __any(bad_twice)) reason = IG_REF_BACKBONE_TWICE;
            if (!reason && nres) {
                const uint32_t s_last = ig_ld_coherent(&T.r_first[A0 + r_next + nres - 1]);
                if (T.resn[A0 + b - 1] != T.resn[A0 + s_last]) reason = IG_REF_LAST_NAME;
            }
            if (nf < (uint32_t)IG_MAX_FRAGS) {
                if (lane == 0) {
                    ingest_frag fr;
                    fr.a = a; fr.b = b; fr.r0 = r_next; fr.nres = nres;
                    fr.meta = chain_char | (j << 8) | (multi_chain ? IG_META_MULTI_CHAIN : 0u) | (multi_frag ? IG_META_MULTI_FRAG : 0u) | (reason << 24);
                    my_frags[nf] = fr;
                }
                nf++;
                if (!reason) { tot.chains++; tot.residues += nres; tot.atoms += b - a; }
            } else overflow = true;
            r_next += nres;
        }
    }
    if (overflow) { nf = 0; tot = ingest_totals{0, 0, 0, 0}; }
    if (lane == 0) {
        if (overflow) file_status[f] = FCZ_INGEST_HOST_FRAGS;
        // the title every record of this file carries: the parsed one, or the file's stem when there is none or it equals
        // the file's base name (src/main.cpp:465)
        const uint32_t tl = title_len[f], nb = name_off[f + 1] - name_off[f];
        bool stem = tl == 0;
        if (!stem && tl == nb) { stem = true; for (uint32_t i = 0; i < tl; i++) if (titles[(size_t)f * IG_TITLE_CAP + i] != (uint8_t)names[name_off[f] + i]) { stem = false; break; } }
        use_stem[f] = stem ? 1u : 0u;
        tot.title_bytes = tot.chains * (stem ? stem_len[f] : tl);
        n_frags[f] = nf;
        totals.chains[f] = tot.chains; totals.residues[f] = tot.residues; totals.atoms[f] = tot.atoms; totals.title_bytes[f] = tot.title_bytes;
    }
}

// ---- k_ingest_fill ------------------------------------------------------------------------------------------------------
struct ingest_out {
    uint32_t *res_off, *atom_off, *title_off; float *x, *y, *z, *bfac_ca; uint8_t *atom_code, *res_code; int32_t *first_res, *first_atom;
    char *chain_id, *titles; uint32_t *chain_file, *chain_meta, *chain_name4;
};
__global__ __launch_bounds__(WAVE) void k_ingest_fill(uint32_t n_files, const uint64_t* __restrict__ abase, ingest_scratch T,
                                                      const ingest_frag* __restrict__ frags, const uint32_t* __restrict__ n_frags,
                                                      const uint32_t* __restrict__ c_off, const uint32_t* __restrict__ r_off,
                                                      const uint32_t* __restrict__ a_off, const uint32_t* __restrict__ t_off,
                                                      const uint32_t* __restrict__ title_len, const uint32_t* __restrict__ use_stem,
                                                      const char* __restrict__ names, const uint32_t* __restrict__ name_off,
                                                      const uint32_t* __restrict__ stem_len, const uint8_t* __restrict__ titles, ingest_out O,
                                                      uint32_t* __restrict__ refused, uint32_t* __restrict__ n_refused) {
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
    const uint32_t nf = n_frags[f];
    const size_t A0 = (size_t)abase[f];
    uint32_t c = c_off[f], r = r_off[f], a = a_off[f], t = t_off[f];
    const bool stem = use_stem[f] != 0;
    const uint32_t tl = stem ? stem_len[f] : title_len[f];
    const uint8_t* tsrc = stem ? (const uint8_t*)names + name_off[f] : titles + (size_t)f * IG_TITLE_CAP;
    for (uint32_t k = 0; k < nf; k++) {
        const ingest_frag fr = frags[(size_t)f * IG_MAX_FRAGS + k];
        if (fr.meta >> 24) {                               // refused: reported (file, meta with the reason), not part of the batch
            if (lane == 0) { const uint32_t q = atomicAdd(n_refused, 1u); refused[2 * (size_t)q] = f; refused[2 * (size_t)q + 1] = fr.meta; }
            continue;
        }
        const uint32_t na = fr.b - fr.a;
        for (uint32_t i = lane; i < na; i += WAVE) {
            const size_t s = A0 + fr.a + i;
            O.x[a + i] = T.x[s]; O.y[a + i] = T.y[s]; O.z[a + i] = T.z[s]; O.atom_code[a + i] = T.acode[s];
        }
        for (uint32_t q = lane; q < fr.nres; q += WAVE) {
            const size_t s = A0 + fr.r0 + q;
            O.atom_off[r + q] = a + (T.r_first[s] - fr.a);
            O.res_code[r + q] = T.r_code[s];
            O.bfac_ca[r + q] = T.r_bfac[s];
        }
        for (uint32_t i = lane; i < tl; i += WAVE) O.titles[t + i] = (char)tsrc[i];
        if (lane == 0) {
            O.res_off[c] = r; O.title_off[c] = t;
            O.first_res[c] = T.resseq[A0 + fr.a]; O.first_atom[c] = T.serial[A0 + fr.a];
            O.chain_id[c] = (char)(T.chain[A0 + fr.a] & 0xffu); O.chain_name4[c] = T.chain[A0 + fr.a];
            O.chain_file[c] = f; O.chain_meta[c] = fr.meta;
        }
        c++; r += fr.nres; a += na; t += tl;
    }
    if (f == n_files - 1 && lane == 0) { O.res_off[c] = r; O.title_off[c] = t; O.atom_off[r] = a; }   // the closing entries
}

// scratch capacity of a file's atom table: an ATOM record the parser accepts has at least 54 characters
__global__ void k_ingest_caps(const uint64_t* __restrict__ file_off, uint32_t n_files, uint64_t* __restrict__ cap) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_files) cap[f] = (file_off[f + 1] - file_off[f]) / 54u + 1u;
}

}  // namespace fcz
