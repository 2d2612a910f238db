// fcz_ingest_cif.h -- structure ingest on the device, mmCIF text (StructureReader reads PDB and mmCIF
// alike, reference src/structure_reader.cpp:31-97, and test/test.cif.gz is one of the reference's two regression pins).
//
//   k_ingest_parse_cif   wavefront = file, run after k_ingest_parse on the files that kernel left to the host because they start
//                        with `data_`. The file is walked in the same 4 KB chunks (coalesced loads a chunk ahead, staged in LDS,
//                        line ends by the 64-byte masks); then lane = line:
//     1. every line is lexed by its own lane (CIF 1.1 tokens: blanks separate, a quote opens a string that its twin followed by
//        a blank closes, `#` starts a comment, `;` in column 1 opens / closes a text field) into token bounds + a line class;
//     2. the wavefront walks the step's line classes IN ORDER through the grammar's automaton (data_ -> items; item = tag value |
//        loop_ tag+ value+; lib/gemmi/cif.hpp:37-148 as host/foldcomp_hip.cpp cif_items restates it): pairs complete, loops
//        have tags and whole rows, no tag twice (hashes in LDS), the `_atom_site` loop's column map (23 names, any case),
//        `_entry.id` (the title), `_cell.angle_*` (the zero-angle rule); it marks the lines that are `_atom_site` rows;
//     3. every marked row leaves as a 32-byte ROW RECORD in the file's slice of the scratch table (the eight dword columns, entry =
//        row ordinal): where its line starts in the file, and the bounds of the fourteen fields the reader looks at (by token
//        ordinal through the column map; an absent optional column has no characters).
//   k_ingest_rows_cif    wavefront = file, lane = row record: the fields' characters straight from the text, decimals as
//                        (double)digits / 10^f (exact operands, IEEE division = strtod's result for <= 15 digits = gemmi's
//                        fast_float), names packed, codes through the LDS hash, removeAlternativePosition by the keep rule of the
//                        PDB kernel, kept atoms written over the records (a row's atom never lands behind its own record).
//                        A kernel of its own because of the register file: the lexer and the automaton hold ~230 VGPRs and two
//                        dozen more of spilled scalars, which leaves two wavefronts per SIMD; the readers in the same kernel
//                        cost either the second wavefront or scratch (profiles/r4_ab_cif_readers.txt). Here they run at the PDB
//                        kernel's register count with every row of the step in flight.
//   The same k_ingest_frags / k_ingest_fill then build the batch.
//
// The device never guesses. It takes the shape every predicted-structure file has -- one data_ block, items one per line (or
// a tag line followed by its value line / text field), loop rows of whole lines, `_atom_site` rows of exactly one line each
// without quotes except around the atom name ("O5'"), chain names of up to four characters, integer residue numbers with an optional
// one-character insertion code, models one after the other under rising plain numbers, residues in rising (number, insertion code)
// order inside a chain run (round 6: the PDB archive's shape beside AFDB's) -- and hands EVERYTHING else back to the host reader (save_ frames, global_ / stop_, several
// blocks, comments after values, other quoted values, longer chain names, a model that comes back or is not a plain number, '?' coordinates, exponents, lines of
// more than 255 characters, bytes outside printable ASCII, ...), which restates gemmi rule for rule and is held to the live
// reference by fuzzing. tests/test_gpu_ingest.py holds this kernel to that reader on mutated files: it never builds a
// different batch and never takes a file the reader fails.
#pragma once
#include "fcz_ingest.h"

namespace fcz {

constexpr int CIF_MAXTOK = 32;                    // columns of an _atom_site row this path takes (AFDB: 25, PDB archive: 21-26)
constexpr int CIF_MAXLINE = 255;                  // characters of a lexed line
constexpr int CIF_NCOL = 23;
constexpr int CIF_TAGSET = 1024;                  // slots of the duplicate-tag table (a block has a few hundred tags)
constexpr int CIF_LINES = 512;                    // line ends of a chunk held at a time (the rest in further rounds)

__device__ const char cif_col_names[CIF_NCOL][20] = {
    "id", "group_pdb", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id", "label_seq_id",
    "pdbx_pdb_ins_code", "cartn_x", "cartn_y", "cartn_z", "occupancy", "b_iso_or_equiv", "pdbx_formal_charge", "auth_seq_id", "auth_comp_id",
    "auth_asym_id", "auth_atom_id", "pdbx_pdb_model_num", "calc_flag", "pdbx_tls_group_id"};
enum { CK_ID, CK_GROUP, CK_SYMBOL, CK_LATOM, CK_ALT, CK_LCOMP, CK_LASYM, CK_LENTITY, CK_LSEQ, CK_INS, CK_X, CK_Y, CK_Z, CK_OCC, CK_B, CK_CHARGE, CK_ASEQ,
       CK_ACOMP, CK_AASYM, CK_AATOM, CK_MODEL, CK_CALC, CK_TLS };
__device__ const double cif_pow10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};

// line classes
enum { CL_BLANK = 0, CL_DATA, CL_LOOP, CL_TAG, CL_PAIR, CL_VALUES, CL_TEXT_OPEN, CL_BAD };

struct cif_lds {
    alignas(16) uint8_t buf[IG_BACK + IG_CHUNK + 80];
    uint32_t line_end[CIF_LINES];
    uint32_t akey[64], rkey[32];
    uint8_t aval[64], rval[32];
    uint32_t tagset[CIF_TAGSET];
    uint8_t tok_s[CIF_MAXTOK][WAVE], tok_e[CIF_MAXTOK][WAVE];   // [token][lane]: a wavefront's accesses to one token ordinal touch 16 banks once
    int8_t pos[CIF_NCOL + 1];
};

#ifdef FCZ_CIF_TIMING
// measurement aid (not built into the product): wavefront-cycles in the parts of k_ingest_parse_cif, in g_ig_timing (FCZ_IG_TIMING)
#define CIF_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define CIF_STAMP(i)
#endif
__device__ __forceinline__ bool cif_is_ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r'; }
__device__ __forceinline__ uint32_t cif_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32u : c; }

__global__ __launch_bounds__(WAVE) void k_ingest_parse_cif(const uint8_t* __restrict__ text, const uint64_t* __restrict__ file_off, uint32_t n_files,
                                                           const uint64_t* __restrict__ abase, ingest_scratch T,
                                                           uint8_t* __restrict__ titles, uint32_t* __restrict__ title_len,
                                                           const int32_t* __restrict__ file_status, uint32_t* __restrict__ cif_rows) {
    __shared__ cif_lds S;
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
    if (lane == 0) cif_rows[f] = 0;                               // (rows + 1 of a file this kernel takes: k_ingest_rows_cif's work list)
    if (file_status[f] != FCZ_INGEST_HOST_FIELD) return;          // only what the PDB kernel handed back can be mmCIF
    const uint64_t f0 = file_off[f], f1 = file_off[f + 1];
    const uint8_t* base = text + f0;
    const uint64_t flen = f1 - f0;
    // gemmi::coor_format_from_content: blanks and comment lines, then `data_` (any case) says mmCIF
    {
        uint64_t i = 0; bool cif = false;
        const long long end = (long long)flen - 8;
        while ((long long)i < end) {
            const uint32_t c = base[i];
            if (c == ' ' || (c - 9u) < 5u) i++;
            else if (c == '#') { while ((long long)i < end && base[i] != '\n') i++; }
            else { cif = cif_lower(base[i]) == 'd' && cif_lower(base[i + 1]) == 'a' && cif_lower(base[i + 2]) == 't' && cif_lower(base[i + 3]) == 'a' && base[i + 4] == '_'; break; }
        }
        if (!cif) return;
    }
    // the tag table, the column map
    for (int k = lane; k < CIF_TAGSET; k += WAVE) S.tagset[k] = 0;
    if (lane <= CIF_NCOL) S.pos[lane] = -1;
    __builtin_amdgcn_wave_barrier();

#ifdef FCZ_CIF_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    const uint64_t A0 = abase[f];
    const uint32_t cap = (uint32_t)(abase[f + 1] - A0);
    uint8_t* tbuf = titles + (size_t)f * IG_TITLE_CAP;
    bool dead = false;                     // the file goes back to the host (uniform)
    // ---- automaton state (uniform) ----
    enum { CX_START, CX_NONE, CX_AFTER_TAG, CX_LOOP_HDR, CX_LOOP_BODY };
    int ctx = CX_START;
    uint32_t ntags = 0; unsigned long long nvals = 0;
    bool in_as = false, as_done = false, as_ready = false;   // the open loop is the _atom_site loop; that loop was read; its column map is final
    bool pending_special = false;          // the tag waiting for its value on a later line is one whose value this path must see
    bool in_text = false;                  // inside a text field
    uint32_t tlen = 0; bool have_title = false;
    int kAsym = -1, kComp = -1, kAtom = -1;
    uint32_t nrows = 0;                    // row records written
    uint64_t line_start = 0;

    // exact compare of a staged line's bytes [at, at + n) with a lower-case word, case folded; uniform arguments
    auto eq_lower = [&](int lo, uint32_t at, const char* w, uint32_t n) -> bool {
        bool eq = true;
        for (uint32_t i = 0; i < n; i++) eq = eq && cif_lower(S.buf[lo + at + i]) == (uint32_t)(uint8_t)w[i];
        return eq;
    };
    auto eq_exact = [&](int lo, uint32_t at, const char* w, uint32_t n) -> bool {
        bool eq = true;
        for (uint32_t i = 0; i < n; i++) eq = eq && S.buf[lo + at + i] == (uint8_t)w[i];
        return eq;
    };
    // no tag twice in a block, any case (cif_document: "duplicate tag"): 32-bit hashes, a collision only costs a hand-back
    auto tag_seen = [&](uint32_t h) -> bool {
        h |= 1u;
        for (uint32_t s = (h * 0x9E3779B1u) >> 22, n = 0; n < (uint32_t)CIF_TAGSET; s = (s + 1) & (CIF_TAGSET - 1), n++) {
            const uint32_t v = S.tagset[s];
            if (v == h) return true;
            if (v == 0) { if (lane == 0) S.tagset[s] = h; __builtin_amdgcn_wave_barrier(); return false; }
        }
        return true;                                    // table full: to the host
    };

    // unaligned dword of the staged text (the line starts anywhere)
    auto ldw = [&](int at) -> uint32_t { uint32_t v; __builtin_memcpy(&v, &S.buf[at], 4); return v; };
    // 0x80 in every byte of v that is zero; the four flags of a dword as a nibble
    auto zf = [](uint32_t v) -> uint32_t { return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu); };
    auto nib = [](uint32_t fl) -> uint32_t { return ((fl >> 7) * 0x10204080u) >> 28; };

    // ---- one step: up to 64 lines, lane = line. ls / le file-relative, lo = offset of the line's first byte in S.buf or -1 ----
    auto do_lines = [&](bool on, uint64_t ls, uint64_t le, int lo) {
        if (dead) return;
        const uint32_t len = on ? (uint32_t)(le - ls) : 0u;
        // ---- 1. lexing, every lane its own line ----
        const uint32_t c_first = !on || len == 0 ? (uint32_t)'\n' : (lo >= 0 ? (uint32_t)S.buf[lo] : (uint32_t)base[ls]);
        const bool semi = on && c_first == ';';
        const unsigned long long m_semi = __ballot(semi);
        const bool text_before = in_text ^ ((__builtin_popcountll(m_semi & ((1ull << lane) - 1ull)) & 1) != 0);   // inside a text field when this line starts
        int cls = CL_BLANK; uint32_t ntok = 0; bool bad = false;
        // lines that are lexed: outside text fields, not a text field's first line, not empty. A line that is not staged or too long
        // may only be a comment
        bool lex = on && !text_before && !semi && len != 0;
        if (on && text_before && semi) {                                   // a text field's last line: nothing but blanks may follow the ';'
            if (lo < 0 || len > (uint32_t)CIF_MAXLINE) bad = true;
            else for (uint32_t i = 1; i < len; i++) if (!cif_is_ws(S.buf[lo + i])) bad = true;
        } else if (on && !text_before && semi) cls = CL_TEXT_OPEN;
        if (lex && (lo < 0 || len > (uint32_t)CIF_MAXLINE)) {
            bool comment = false;
            for (uint32_t i = 0; i < len; i++) { const uint32_t c = lo >= 0 ? (uint32_t)S.buf[lo + i] : (uint32_t)base[ls + i]; if (cif_is_ws(c)) continue; comment = c == '#'; break; }
            bad = bad || !comment; lex = false;
        }
        // (a) blanks by dword: one bit per character, set where a blank (or a control character: those fail the line below) stands;
        //     characters past the line count as blanks
        unsigned long long W0 = ~0ull, W1 = ~0ull, W2 = ~0ull, W3 = ~0ull;
        {
            const uint32_t nd = lex ? (len + 3u) >> 2 : 0u;
            uint32_t maxd = nd;
#pragma unroll
            for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)maxd, d, WAVE); maxd = o > maxd ? o : maxd; }
            for (uint32_t d0 = 0; d0 < maxd; d0 += 4) {               // four dwords in flight per round trip to the LDS
                uint32_t xs[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xs[u] = (d0 + (uint32_t)u < nd) ? ldw(lo + 4 * (int)(d0 + (uint32_t)u)) : 0x20202020u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t d = d0 + (uint32_t)u;
                    uint32_t x = xs[u];
                    if (d < nd) { const uint32_t rest = len - 4u * d; if (rest < 4u) { const uint32_t keep = (1u << (8u * rest)) - 1u; x = (x & keep) | (0x20202020u & ~keep); } }
                    const uint32_t lt20 = zf(x & 0x60606060u);                                  // characters below 0x20 (of 7-bit characters)
                    const uint32_t wsf = zf(x ^ 0x20202020u) | lt20;
                    bool b2 = (x & 0x80808080u) != 0u || zf(x ^ 0x7f7f7f7fu) != 0u;              // outside printable ASCII
                    b2 = b2 || (lt20 & ~(zf(x ^ 0x09090909u) | zf(x ^ 0x0d0d0d0du))) != 0u;      // a control character that is no tab / CR
                    bad = bad || (d < nd && b2);
                    const unsigned long long n4 = (unsigned long long)nib(wsf) << (4u * (d & 15u));
                    const unsigned long long clr = ~(0xfull << (4u * (d & 15u)));
                    if ((d >> 4) == 0u) W0 = (W0 & clr) | n4; else if ((d >> 4) == 1u) W1 = (W1 & clr) | n4; else if ((d >> 4) == 2u) W2 = (W2 & clr) | n4; else W3 = (W3 & clr) | n4;
                }
            }
        }
        CIF_STAMP(3)
        // (b) token bounds: a token starts where a character follows a blank (or the line's start), ends at the next blank. With a
        //     token's start in a register its first characters are fetched at once (two tokens per round trip) and looked at:
        //     a quote opens a string (the character-by-character lexer takes the line), '#' a comment, '$' a frame reference, '_' a
        //     tag; reserved words open no value (CifScanner::value): data_ loop_ stop_ save_ global_, any case
        bool slow = false; uint32_t tag_like = 0, kw_first = 0;
        if (lex && !bad) {
            const unsigned long long p1 = W0 >> 63, p2 = W1 >> 63, p3 = W2 >> 63;
            const unsigned long long S0 = ~W0 & ((W0 << 1) | 1ull), S1 = ~W1 & ((W1 << 1) | p1), S2 = ~W2 & ((W2 << 1) | p2), S3 = ~W3 & ((W3 << 1) | p3);
            const unsigned long long E0 = W0 & ~((W0 << 1) | 1ull), E1 = W1 & ~((W1 << 1) | p1), E2 = W2 & ~((W2 << 1) | p2), E3 = W3 & ~((W3 << 1) | p3);
            ntok = (uint32_t)(__builtin_popcountll(S0) + __builtin_popcountll(S1) + __builtin_popcountll(S2) + __builtin_popcountll(S3));
            uint32_t t = 0, u = 0; bool comment = false;
            auto look = [&](uint32_t ti, int at, uint32_t w0) {
                const uint32_t c0 = w0 & 0xffu;
                if (c0 == '#') { if (ti == 0) comment = true; else if (!comment) bad = true; }
                if (comment) return;
                if (c0 == '\'' || c0 == '"') slow = true;
                if (c0 == '$' || (c0 == '_' && ti != 0)) bad = true;
                if (c0 == '_' && ti == 0) tag_like = 1;
                const uint32_t l4 = w0 | 0x20202020u;                                    // (letters folded; a blank or '_' among the four matches nothing)
                if (l4 == 0x61746164u || l4 == 0x706f6f6cu || l4 == 0x706f7473u || l4 == 0x65766173u || l4 == 0x626f6c67u) {   // data loop stop save glob
                    const uint32_t w1 = ldw(at + 4);
                    uint32_t kw = 0;
                    if ((w1 & 0xffu) == '_') kw = l4 == 0x61746164u ? 1u : l4 == 0x706f6f6cu ? 2u : l4 == 0x706f7473u ? 5u : l4 == 0x65766173u ? 4u : 0u;
                    if (l4 == 0x626f6c67u && ((w1 | 0x00002020u) & 0x00ffffffu) == 0x005f6c61u) kw = 3u;                                    // "al_"
                    if (kw) { if (ti == 0) kw_first = kw; else bad = true; }
                }
            };
            auto put = [&](unsigned long long ms, unsigned long long me, uint32_t off) {
                while (ms) {
                    const uint32_t q0 = off + (uint32_t)__builtin_ctzll(ms); ms &= ms - 1;
                    const bool two = ms != 0ull;
                    const uint32_t q1 = two ? off + (uint32_t)__builtin_ctzll(ms) : q0; ms &= ms - 1;
                    const uint32_t w0 = ldw(lo + (int)q0), w1 = ldw(lo + (int)q1);
                    if (t < (uint32_t)CIF_MAXTOK) S.tok_s[t][lane] = (uint8_t)q0;
                    if (two && t + 1 < (uint32_t)CIF_MAXTOK) S.tok_s[t + 1][lane] = (uint8_t)q1;
                    look(t, lo + (int)q0, w0);
                    if (two) look(t + 1, lo + (int)q1, w1);
                    t += two ? 2u : 1u;
                }
                for (; me; me &= me - 1) { if (u < (uint32_t)CIF_MAXTOK) S.tok_e[u][lane] = (uint8_t)(off + (uint32_t)__builtin_ctzll(me)); u++; }
            };
            put(S0, E0, 0u); put(S1, E1, 64u); put(S2, E2, 128u); put(S3, E3, 192u);
            if (comment) { ntok = 0; slow = false; bad = false; tag_like = 0; kw_first = 0; }   // (what a comment holds is nobody's business)
            else if (ntok > (uint32_t)CIF_MAXTOK) slow = true;              // (the character-by-character lexer counts any number of tokens)
        }
        CIF_STAMP(4)
        // (d) the character-by-character lexer for the lines that hold a quoted string (or more tokens than the table)
        if (__any(slow && !bad)) {
            if (slow && !bad) {
                ntok = 0; kw_first = 0; tag_like = 0;
                bool in_tok = false, quoted = false; uint32_t quote = 0, ts = 0, tl = 0; unsigned long long acc = 0;
                for (uint32_t i = 0; i < len; i++) {
                    const uint32_t c = S.buf[lo + i];
                    bool ends = false; uint32_t end_at = i;
                    if (!in_tok) {
                        if (cif_is_ws(c)) continue;
                        if (c == '#') { if (ntok != 0) bad = true; break; }
                        in_tok = true; ts = i; tl = 0; acc = 0; quoted = c == '\'' || c == '"'; quote = quoted ? c : 0u;
                        if (c == '$' || (c == '_' && ntok != 0)) bad = true;
                        if (c == '_' && ntok == 0) tag_like = 1;
                    } else if (quote) {
                        if (c == quote) { const uint32_t nx = i + 1 < len ? (uint32_t)S.buf[lo + i + 1] : (uint32_t)' '; if (cif_is_ws(nx) || nx == '#') { ends = true; end_at = i + 1; } }
                    } else if (cif_is_ws(c)) { ends = true; end_at = i; }
                    if (ends) {
                        if (ntok < (uint32_t)CIF_MAXTOK) { S.tok_s[ntok][lane] = (uint8_t)ts; S.tok_e[ntok][lane] = (uint8_t)end_at; }
                        ntok++; in_tok = false; quote = 0;
                        continue;
                    }
                    if (!quoted) {
                        if (tl < 7u) acc |= (unsigned long long)cif_lower(c) << (8 * tl);
                        tl++;
                        if (tl == 5u) {
                            const unsigned long long a5 = acc & 0xffffffffffull;
                            const uint32_t kw = a5 == 0x5f61746164ull ? 1u : a5 == 0x5f706f6f6cull ? 2u : a5 == 0x5f706f7473ull ? 5u : a5 == 0x5f65766173ull ? 4u : 0u;
                            if (kw) { if (ntok == 0) kw_first = kw; else bad = true; }
                        }
                        if (tl == 7u && (acc & 0xffffffffffffffull) == 0x5f6c61626f6c67ull) { if (ntok == 0) kw_first = 3u; else bad = true; }
                    }
                }
                if (in_tok) {
                    if (quote) bad = true;
                    else { if (ntok < (uint32_t)CIF_MAXTOK) { S.tok_s[ntok][lane] = (uint8_t)ts; S.tok_e[ntok][lane] = (uint8_t)len; } ntok++; }
                }
            }
        }
        if (lex && !bad) {
            const uint32_t l0 = ntok ? (uint32_t)S.tok_e[0][lane] - (uint32_t)S.tok_s[0][lane] : 0u;
            if (ntok == 0) cls = CL_BLANK;
            else if (kw_first == 2u) { cls = CL_LOOP; if (ntok != 1 || l0 != 5u) bad = true; }
            else if (kw_first == 1u) { cls = CL_DATA; if (ntok != 1 || l0 <= 5u) bad = true; }
            else if (kw_first) bad = true;                                  // save_, stop_, global_ as the first token
            else if (tag_like) { cls = ntok == 1 ? CL_TAG : CL_PAIR; if (ntok > 2 || l0 < 2u) bad = true; }
            else cls = CL_VALUES;
        }
        if (bad) cls = CL_BAD;
        if (__any(cls == CL_BAD)) { dead = true; return; }
        in_text = in_text ^ ((__builtin_popcountll(m_semi) & 1) != 0);
        CIF_STAMP(5)
        // ---- 2. the grammar over the step's lines, in order (uniform) ----
        unsigned long long rowmask = 0;
        const unsigned long long m_on = __ballot(on);
        const unsigned long long m_rowlike = __ballot(on && cls == CL_VALUES && ntok == ntags);
        const unsigned long long m_values = __ballot(on && cls == CL_VALUES), m_quiet = __ballot(on && (cls == CL_VALUES || cls == CL_BLANK));
        if (ctx == CX_LOOP_BODY && in_as && as_ready && m_on && m_rowlike == m_on) {
            rowmask = m_on; nvals += (unsigned long long)ntags * (unsigned)__builtin_popcountll(m_on);      // a step of nothing but rows
        } else if (ctx == CX_LOOP_BODY && !in_as && m_on && m_quiet == m_on) {
            // a step of nothing but values (and blank lines) of another category's loop (the per-residue tables of a predicted-structure
            // file): they only count
            uint32_t kv = ((m_values >> lane) & 1ull) ? ntok : 0u;
#pragma unroll
            for (int d = WAVE / 2; d > 0; d >>= 1) kv += (uint32_t)__shfl_xor((int)kv, d, WAVE);
            nvals += kv;
        } else {
            auto end_item = [&]() {                                                   // what stands before a new tag / loop_ must be complete
                if (ctx == CX_LOOP_BODY) { if (ntags == 0 || nvals % ntags != 0) dead = true; if (in_as) as_done = true; ctx = CX_NONE; }
                else if (ctx != CX_NONE) dead = true;
            };
            for (unsigned long long m = m_on & __ballot(cls != CL_BLANK); m && !dead; m &= m - 1) {
                const int l = __builtin_ctzll(m);
                // (l is uniform: v_readlane, not the LDS crossbar of a shuffle)
                const int c = __builtin_amdgcn_readlane(cls, l);
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)ntok, l);
                const int llo = __builtin_amdgcn_readlane(lo, l);
                if (c == CL_DATA) { if (ctx != CX_START) dead = true; ctx = CX_NONE; continue; }      // a second block: to the host
                if (ctx == CX_START) { dead = true; break; }
                if (c == CL_LOOP) { end_item(); ctx = CX_LOOP_HDR; ntags = 0; nvals = 0; in_as = false; continue; }
                if (c == CL_TAG || c == CL_PAIR) {
                    const uint32_t t0 = S.tok_s[0][l], tn = (uint32_t)S.tok_e[0][l] - t0;
                    uint32_t h = 2166136261u;
                    for (uint32_t i = 0; i < tn; i++) h = (h ^ cif_lower(S.buf[llo + t0 + i])) * 16777619u;
                    if (tag_seen(h)) { dead = true; break; }
                    const bool cat_as = tn > 11u && eq_lower(llo, t0, "_atom_site.", 11);
                    const bool cat_cell = tn > 6u && eq_lower(llo, t0, "_cell.", 6);
                    const bool is_entry = tn == 9u && eq_lower(llo, t0, "_entry.id", 9);
                    if (ctx == CX_LOOP_HDR) {
                        if (c == CL_PAIR) { dead = true; break; }                    // (a loop's last tag with the first value on its line: legal, not taken here)
                        if (cat_cell || is_entry) { dead = true; break; }             // looped cell / entry id: the host knows the rules
                        if (cat_as) {
                            if (as_done || (ntags && !in_as)) { dead = true; break; }
                            in_as = true;
                            for (int q = 0; q < CIF_NCOL; q++) {
                                uint32_t wl = 0; while (cif_col_names[q][wl]) wl++;
                                if (tn == 11u + wl && eq_lower(llo, t0 + 11u, cif_col_names[q], wl)) { if (lane == 0) S.pos[q] = (int8_t)ntags; }
                            }
                            __builtin_amdgcn_wave_barrier();
                        } else if (in_as) { dead = true; break; }                          // another category's tag inside the _atom_site loop
                        ntags++;
                        if (in_as && ntags > (uint32_t)CIF_MAXTOK) { dead = true; break; }
                        continue;
                    }
                    end_item();
                    if (dead) break;
                    if (cat_as) { dead = true; break; }                              // _atom_site items as pairs (a one-atom file): to the host
                    // values this path must see: the title, the angles of the zero-angle rule
                    const bool ang = cat_cell && ((tn == 17u && eq_lower(llo, t0, "_cell.angle_alpha", 17)) || (tn == 16u && eq_lower(llo, t0, "_cell.angle_beta", 16)));
                    if (is_entry && !eq_exact(llo, t0, "_entry.id", 9)) { dead = true; break; }   // (a pair is looked up by its exact spelling)
                    if (c == CL_TAG) { pending_special = is_entry || ang; ctx = CX_AFTER_TAG; continue; }
                    const uint32_t v0 = S.tok_s[1][l], vn = (uint32_t)S.tok_e[1][l] - v0;
                    const uint32_t vc = S.buf[llo + v0];
                    if (is_entry) {
                        if (vc == '\'' || vc == '"' || (vn == 1u && (vc == '?' || vc == '.')) || vn > (uint32_t)IG_TITLE_CAP) { dead = true; break; }
                        for (uint32_t i = (uint32_t)lane; i < vn; i += WAVE) tbuf[i] = S.buf[llo + v0 + i];
                        tlen = vn; have_title = true;
                    }
                    if (ang && !(vc - '1' < 9u)) { dead = true; break; }              // certainly not zero only when it starts with 1-9
                    ctx = CX_NONE;
                    continue;
                }
                // values (a text field counts one)
                const uint32_t kv = c == CL_TEXT_OPEN ? 1u : k;
                if (ctx == CX_AFTER_TAG) { if (kv != 1u || pending_special) { dead = true; break; } ctx = CX_NONE; continue; }
                if (ctx == CX_LOOP_HDR) {
                    if (ntags == 0) { dead = true; break; }
                    ctx = CX_LOOP_BODY; nvals = 0;
                    if (in_as) {
                        // the columns make_structure needs (mmcif.hpp: the tags without '?'), the name columns by auth_ then label_
                        const int req[10] = {CK_ID, CK_SYMBOL, CK_ALT, CK_LASYM, CK_X, CK_Y, CK_Z, CK_OCC, CK_B, CK_ASEQ};
                        for (int q = 0; q < 10; q++) if (S.pos[req[q]] < 0) dead = true;
                        kAsym = S.pos[CK_AASYM] >= 0 ? CK_AASYM : CK_LASYM; kComp = S.pos[CK_ACOMP] >= 0 ? CK_ACOMP : CK_LCOMP; kAtom = S.pos[CK_AATOM] >= 0 ? CK_AATOM : CK_LATOM;
                        if (S.pos[kComp] < 0 || S.pos[kAtom] < 0) dead = true;
                        if (dead) break;
                        as_ready = true;
                    }
                }
                if (ctx != CX_LOOP_BODY) { dead = true; break; }
                if (c == CL_VALUES) {
                    // a loop's body is a run of value lines (and blank / comment lines): the whole run from this line on is taken at
                    // once -- a masked sum of its token counts -- instead of a turn of this loop per line
                    const unsigned long long quiet = m_quiet >> l;                              // bit 0 = this line
                    const int run = ~quiet ? __builtin_ctzll(~quiet) : 64;
                    const unsigned long long in_run = (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << l;
                    const bool mine = ((in_run & m_values) >> lane) & 1ull;
                    uint32_t sum = mine ? ntok : 0u; uint32_t odd = (mine && ntok != ntags) ? 1u : 0u;
#pragma unroll
                    for (int d = WAVE / 2; d > 0; d >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, d, WAVE); odd |= (uint32_t)__shfl_xor((int)odd, d, WAVE); }
                    nvals += sum;
                    if (in_as) { if (odd) { dead = true; break; } rowmask |= in_run & m_values; }
                    m &= ~in_run;                                                              // (the loop's own `m &= m - 1` then clears nothing of ours)
                    m |= 1ull << l;
                    continue;
                }
                nvals += kv;                                                                   // a text field
                if (in_as) { dead = true; break; }
            }
        }
        CIF_STAMP(6)
        if (dead) return;
        // ---- 3. the _atom_site rows of the step leave as row records: where the line starts, the bounds of the fourteen fields ----
        if (rowmask == 0ull) return;
        const bool row = ((rowmask >> lane) & 1ull) != 0;
        // a quoted value in a row (such a line went through the character lexer): only the atom name may be one -- the primes of
        // nucleotide and ligand atoms ("O5'") force the quotes in archive files; the reader strips them (cif::as_string). Anything
        // else quoted (a quoted '?' is not a null, a quoted number is no number for as_number) is the host's
        bool qname = false;
        if (row && slow) {
            const int pa = S.pos[CK_LATOM], pb = S.pos[CK_AATOM];
            for (uint32_t t = 0; t < ntok; t++) {
                const uint32_t c0 = S.buf[lo + (int)S.tok_s[t][lane]];
                if (c0 == '\'' || c0 == '"') { if ((int)t == pa || (int)t == pb) qname = true; else bad = true; }
            }
        }
        if (__any(row && bad)) { dead = true; return; }
        const uint32_t n_new = (uint32_t)__builtin_popcountll(rowmask);
        if (nrows + n_new > cap || (flen >> 32) != 0ull) { dead = true; return; }
        {
            const int cols[14] = {S.pos[CK_ID], S.pos[CK_ASEQ], S.pos[CK_X], S.pos[CK_Y], S.pos[CK_Z], S.pos[CK_B], S.pos[kAtom], S.pos[kComp], S.pos[kAsym],
                                  S.pos[CK_ALT], S.pos[CK_INS], S.pos[CK_LSEQ], S.pos[CK_CHARGE], S.pos[CK_MODEL]};
            uint32_t w[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 14; q++) {
                const int c = cols[q] >= 0 ? cols[q] : 0;
                uint32_t ts = (row && cols[q] >= 0) ? (uint32_t)S.tok_s[c][lane] : 0u, te = (row && cols[q] >= 0) ? (uint32_t)S.tok_e[c][lane] : 0u;
                if (q == 6 && qname && te >= ts + 2u) { const uint32_t c0 = S.buf[lo + (int)ts]; if (c0 == '\'' || c0 == '"') { ts++; te--; } }   // the name between its quotes
                w[q >> 1] |= (ts | (te << 8)) << (16 * (q & 1));
            }
            if (row) {
                const size_t o = (size_t)A0 + nrows + (uint32_t)__builtin_popcountll(rowmask & ((1ull << lane) - 1ull));
                T.name[o] = (uint32_t)ls; T.resn[o] = w[0]; T.serial[o] = (int32_t)w[1]; T.resseq[o] = (int32_t)w[2];
                T.x[o] = __uint_as_float(w[3]); T.y[o] = __uint_as_float(w[4]); T.z[o] = __uint_as_float(w[5]); T.b[o] = __uint_as_float(w[6]);
            }
        }
        nrows += n_new;
        CIF_STAMP(7)
    };

    // ---- the chunk walk of k_ingest_parse: coalesced loads a chunk ahead, staged in LDS behind the previous chunk's tail ----
    uint64_t c0_staged = 0;
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
    CIF_STAMP(0)
    uint64_t c0 = 0;
    for (; c0 < flen && !dead; c0 += IG_CHUNK) {
        {
            uint32_t tail;
            __builtin_memcpy(&tail, &S.buf[IG_CHUNK + 4 * lane], 4);
            __builtin_amdgcn_wave_barrier();
            __builtin_memcpy(&S.buf[4 * lane], &tail, 4);
#pragma unroll
            for (int d = 0; d < 16; d++) __builtin_memcpy(&S.buf[IG_BACK + 4 * (d * WAVE + lane)], &pre[d], 4);
            __builtin_amdgcn_wave_barrier();
            c0_staged = c0;
            load_chunk(c0 + IG_CHUNK);
        }
        CIF_STAMP(1)
        const uint64_t my = c0 + 64ull * (uint64_t)lane;
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
                    const uint32_t m = ~(((xz & 0x7f7f7f7fu) + 0x7f7f7f7fu) | xz | 0x7f7f7f7fu);
                    const uint32_t zz = ~(((vv[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | vv[k] | 0x7f7f7f7fu);
                    const uint32_t nn = ((m >> 7) * 0x10204080u) >> 28, nz = ((zz >> 7) * 0x10204080u) >> 28;
                    if (d < 8) { nl_lo |= nn << (4 * d); z_lo |= nz << (4 * d); } else { nl_hi |= nn << (4 * (d - 8)); z_hi |= nz << (4 * (d - 8)); }
                }
            }
        }
        const uint32_t vb = my >= flen ? 0u : (flen - my >= 64u ? 64u : (uint32_t)(flen - my));
        const unsigned long long vmask = vb >= 64u ? ~0ull : ((1ull << vb) - 1ull);
        const unsigned long long nlm = (((unsigned long long)nl_hi << 32) | nl_lo) & vmask;
        const uint32_t nul = ((((unsigned long long)z_hi << 32) | z_lo) & vmask) != 0ull ? 1u : 0u;
        if (__any(nul != 0u)) { dead = true; break; }                       // a NUL in the text: the readers' business
        const uint32_t cnt = (uint32_t)__builtin_popcountll(nlm);
        uint32_t total;
        uint32_t ord = wave_excl_scan_dpp(cnt, &total);
        for (uint32_t r0 = 0; (r0 < total || r0 == 0) && !dead; r0 += CIF_LINES) {
            uint32_t o = ord;
            for (unsigned long long m = nlm; m; m &= m - 1) {
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                if (o >= r0 && o < r0 + (uint32_t)CIF_LINES) S.line_end[o - r0] = 64u * (uint32_t)lane + bit;
                o++;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_here = total - r0 < (uint32_t)CIF_LINES ? total - r0 : (uint32_t)CIF_LINES;
            for (uint32_t k0 = 0; k0 < n_here && !dead; k0 += WAVE) {
                const uint32_t k = k0 + (uint32_t)lane;
                const bool on = k < n_here;
                const uint64_t le = on ? c0 + S.line_end[k] : 0;
                const uint64_t ls = on ? (k == 0 ? line_start : c0 + S.line_end[k - 1] + 1) : 0;
                const long long rel = (long long)ls - (long long)c0;
                CIF_STAMP(2)
                do_lines(on, ls, le, (on && rel >= -(long long)IG_BACK) ? (int)(IG_BACK + rel) : -1);
            }
            if (n_here) line_start = c0 + S.line_end[n_here - 1] + 1;
            __builtin_amdgcn_wave_barrier();
            if (total == 0) break;
        }
    }
    if (!dead && line_start < flen) {                                      // a last line without a line end
        const long long rel = (long long)line_start - (long long)c0_staged;
        do_lines(lane == 0, line_start, flen, rel >= -(long long)IG_BACK ? (int)(IG_BACK + rel) : -1);
    }
    // the end of the file closes what is open: a loop with whole rows, nothing else
    if (!dead) {
        if (in_text) dead = true;
        if (ctx == CX_LOOP_BODY) { if (ntags == 0 || nvals % ntags != 0) dead = true; if (in_as) as_done = true; }
        else if (ctx != CX_NONE) dead = true;
        if (!as_done || !have_title || nrows == 0) dead = true;             // no atoms, no title: the host reports what the reference reports
    }
    if (lane == 0 && !dead) {
        title_len[f] = tlen;
        cif_rows[f] = nrows + 1u;                                          // the file's status stays "to the host" until its rows are read
    }
#ifdef FCZ_CIF_TIMING
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_ig_timing[i], tacc[i]);
#endif
}

// ---- the rows of the files k_ingest_parse_cif took: wavefront = file, lane = row record ----
struct cif_rows_lds { uint32_t akey[64], rkey[32]; uint8_t aval[64], rval[32]; };

__global__ __launch_bounds__(WAVE) void k_ingest_rows_cif(const uint8_t* __restrict__ text, const uint64_t* __restrict__ file_off, uint32_t n_files,
                                                          uint64_t text_bytes, const uint64_t* __restrict__ abase, ingest_scratch T,
                                                          uint32_t* __restrict__ n_kept, int32_t* __restrict__ file_status,
                                                          const uint32_t* __restrict__ cif_rows) {
    __shared__ cif_rows_lds S;
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
    const uint32_t rows1 = cif_rows[f];
    if (rows1 == 0u) return;                                       // not a file the lexer took
    const uint32_t nrows = rows1 - 1u;
    const uint8_t* base = text + file_off[f];
    const uint8_t* lim = text + text_bytes;
    // name -> code tables (as in k_ingest_parse)
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
    const uint64_t A0 = abase[f];
    uint32_t kept = 0;
    bool have_last = false; uint32_t last_name = 0, last_comp = 0, last_ch = 0, last_ic = 0; int32_t last_num = 0;
    unsigned long long last_mdl = 0; uint32_t last_mdl_n = 0;
    bool dead = false;

    struct fld { uint32_t w[4]; uint32_t n; };
    auto ch_at = [](const fld& fd, int i) -> uint32_t { return (fd.w[i >> 2] >> (8 * (i & 3))) & 0xffu; };
    auto is_null = [&](const fld& fd) -> bool { return fd.n == 1u && ((fd.w[0] & 0xffu) == '?' || (fd.w[0] & 0xffu) == '.'); };
    // the numeric readers come in two widths: every numeric field of the step at most eight characters (what coordinate files
    // hold: the wavefront decides it from the field bounds before a character is loaded) -> two dwords and a 32-bit mantissa;
    // otherwise sixteen characters and the 64-bit one. Same value either way: the digits are exact in both.
    auto lt = [](uint32_t a, uint32_t b) -> uint32_t { return (a - b) >> 31; };            // a < b for a, b < 2^31
    auto is_digit = [](uint32_t d) -> uint32_t { return ((d | (9u - d)) >> 31) ^ 1u; };    // d = c - '0' as two's complement, |d| < 2^31
    auto eq = [](uint32_t a, uint32_t b) -> uint32_t { const uint32_t x = a ^ b; return ((x | (0u - x)) >> 31) ^ 1u; };
    auto integer = [&](const fld& f, int32_t* out, auto NC) -> bool {        // [+-]digits, at most nine of them
        constexpr int N = decltype(NC)::value;
        const uint32_t c0 = f.w[0] & 0xffu;
        const uint32_t neg = eq(c0, '-'), sg = neg | eq(c0, '+');
        uint32_t v = 0, nd = 0, ok = lt(f.n, (uint32_t)N + 1u);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t d = ch_at(f, i) - '0';
            const uint32_t in = i == 0 ? (lt(0u, f.n) & (sg ^ 1u)) : lt((uint32_t)i, f.n);
            const uint32_t dig = is_digit(d);
            ok &= (in ^ 1u) | dig;
            const uint32_t t = 0u - in;                                            // all ones where the character counts
            v = ((v * 10u + d) & t) | (v & ~t); nd += in;
        }
        *out = (int32_t)((v ^ (0u - neg)) + neg);
        return (ok & lt(0u, nd) & lt(nd, 10u)) != 0u;
    };
    auto decimal = [&](const fld& f, float* out, auto NC) -> bool {          // -digits.digits, at most 15 digits in 16 characters: cif::as_number's fast path
        constexpr int N = decltype(NC)::value;
        using mant = typename std::conditional<(N <= 9), uint32_t, unsigned long long>::type;
        const uint32_t neg = eq(f.w[0] & 0xffu, '-');
        mant m = 0; uint32_t nd = 0, nf = 0, point = 0, ok = lt(f.n, (uint32_t)N + 1u);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t c = ch_at(f, i), d = c - '0';
            const uint32_t in = i == 0 ? (lt(0u, f.n) & (neg ^ 1u)) : lt((uint32_t)i, f.n);
            const uint32_t dig = is_digit(d), pt = eq(c, '.');
            ok &= (in ^ 1u) | dig | (pt & (point ^ 1u));
            const uint32_t take = in & dig;
            const mant t = (mant)0 - (mant)take;
            m = ((m * (mant)10 + (mant)d) & t) | (m & ~t); nd += take; nf += take & point;
            point |= in & pt;
        }
        const double v = (double)m / cif_pow10[nf & 15u];
        *out = (float)(neg ? -v : v);
        return (ok & lt(0u, nd) & lt(nd, 16u)) != 0u;
    };
    auto pack = [&](const fld& fd, uint32_t* out) -> bool {                   // a name of one to four characters
        *out = fd.n >= 4u ? fd.w[0] : (fd.w[0] & ((1u << (8u * fd.n)) - 1u));
        return (fd.n >= 1u) & (fd.n <= 4u) & !is_null(fd);
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>; using I10 = std::integral_constant<int, 10>; using I16 = std::integral_constant<int, 16>;

    for (uint32_t s0 = 0; s0 < nrows; s0 += WAVE) {
        const bool row = s0 + (uint32_t)lane < nrows;
        const unsigned long long rowmask = __ballot(row);
        const size_t at = (size_t)A0 + s0 + (uint32_t)lane;
        // the record: line start, fourteen (start, end) byte pairs
        uint32_t ls = 0, w[7] = {0, 0, 0, 0, 0, 0, 0};
        if (row) {
            ls = T.name[at]; w[0] = T.resn[at]; w[1] = (uint32_t)T.serial[at]; w[2] = (uint32_t)T.resseq[at];
            w[3] = __float_as_uint(T.x[at]); w[4] = __float_as_uint(T.y[at]); w[5] = __float_as_uint(T.z[at]); w[6] = __float_as_uint(T.b[at]);
        }
        uint32_t fs[14], fn[14];
#pragma unroll
        for (int q = 0; q < 14; q++) { const uint32_t p2 = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu; fs[q] = p2 & 0xffu; fn[q] = (p2 >> 8) - (p2 & 0xffu); }
        const uint8_t* lp = base + ls;
        auto load = [&](int q, auto NW) -> fld {
            constexpr int W = decltype(NW)::value;
            fld fd; fd.n = fn[q];
            const uint8_t* p = lp + fs[q];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t v = 0;
                if (k < W && row) {
                    if (p + 4 * k + 4 <= lim) v = ld_u32(p + 4 * k);
                    else for (int b = 0; b < 4; b++) if (p + 4 * k + b < lim) v |= (uint32_t)p[4 * k + b] << (8 * b);
                }
                fd.w[k] = v;
            }
            return fd;
        };
        uint32_t widest = 0;
        { const int numeric[8] = {0, 1, 2, 3, 4, 5, 11, 12};
#pragma unroll
          for (int q = 0; q < 8; q++) widest = max(widest, fn[numeric[q]]); }
        const bool narrow = !__any(widest > 8u);
        bool rbad = false;
        uint32_t an = 0, rn = 0, ch = ' ', ic = 0; int32_t serial = 0, num = 0; float x = 0.f, y = 0.f, z = 0.f, bf = 0.f; unsigned long long mdl = 0; uint32_t mdl_n = 0;
        auto fields = [&](auto NI, auto ND, auto NWD) {
            const fld f0 = load(0, NWD), f1 = load(1, NWD), f2 = load(2, NWD), f3 = load(3, NWD), f4 = load(4, NWD), f5 = load(5, NWD);
            const fld f6 = load(6, I1{}), f7 = load(7, I1{}), f8 = load(8, I1{}), f9 = load(9, I1{}), f10 = load(10, I1{});
            const fld f11 = load(11, NWD), f12 = load(12, NWD), f13 = load(13, I2{});
            rbad = rbad | !integer(f0, &serial, NI) | !integer(f1, &num, NI) | !decimal(f2, &x, ND) | !decimal(f3, &y, ND) | !decimal(f4, &z, ND);
            // the chain's name: one to four characters, packed (archive files of large complexes name chains "AA", "B2" ...); a longer
            // one is the host's. The insertion code (pdbx_PDB_ins_code): absent, null, or ONE character (cif::as_char); residue
            // identity is (number, code) with the code's case folded (gemmi SeqId::operator==), 0 = none
            rbad = rbad | !decimal(f5, &bf, ND) | !pack(f6, &an) | !pack(f7, &rn) | !pack(f8, &ch) | (!is_null(f9) & (f9.n != 1u));
            int32_t dummy;
            // an optional column that the loop does not have has no characters
            rbad = rbad | ((f10.n != 0u) & !is_null(f10) & (f10.n != 1u));
            ic = ((f10.n == 1u) & !is_null(f10)) ? ((f10.w[0] & 0xffu) & ~0x20u) : 0u;
            rbad = rbad | ((f10.n == 1u) & !is_null(f10) & (ic == 0u));                      // (a blank cannot be a token; a code that folds to nothing: not here)
            rbad = rbad | ((f11.n != 0u) & !is_null(f11) & !integer(f11, &dummy, NI));
            rbad = rbad | ((f12.n != 0u) & !is_null(f12) & !integer(f12, &dummy, NI));
            rbad = rbad | (f13.n > 8u);
            const unsigned long long v8 = (unsigned long long)f13.w[0] | ((unsigned long long)f13.w[1] << 32);
            mdl = f13.n >= 8u ? v8 : (v8 & ((1ull << (8u * f13.n)) - 1ull)); mdl_n = f13.n;
        };
        if (narrow) fields(I8{}, I8{}, I2{}); else fields(I10{}, I16{}, I4{});
        rbad = rbad & row;
        // models one after the other (below); residues of a chain run in rising order (the reader regroups anything else); keep rule
        // of removeAlternativePosition
        const uint32_t pl = ig_prev_lane(rowmask, lane);
        const int src = pl < 64u ? (int)pl : 0;
        const int32_t s_num = __shfl(num, src, WAVE);
        const uint32_t s_rn = (uint32_t)__shfl((int)rn, src, WAVE), s_ch = (uint32_t)__shfl((int)ch, src, WAVE), s_an = (uint32_t)__shfl((int)an, src, WAVE);
        const uint32_t s_ic = (uint32_t)__shfl((int)ic, src, WAVE);
        const bool has_p = pl < 64u ? true : have_last;
        const int32_t p_num = pl < 64u ? s_num : last_num;
        const uint32_t p_rn = pl < 64u ? s_rn : last_comp, p_ch = pl < 64u ? s_ch : last_ch, p_an = pl < 64u ? s_an : last_name, p_ic = pl < 64u ? s_ic : last_ic;
        // Several models (NMR ensembles of the archive): the reader gives every model name an ordinal at its first row and sorts the
        // atoms by (model, chain, residue) -- file order as long as no row returns to an EARLIER model. Here: the model's name may
        // change where the new one is a plain number larger than the one before it (1, 2, 3 ...: no name can come twice then); a
        // chain starts anew there, whatever its name (make_structure_from_block: `chain = nullptr` at a new model,
        // lib/gemmi/mmcif.hpp:560-680), so the order rule below does not look across the step. Anything else: the host's.
        const unsigned long long s_mdl = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(mdl >> 32), src, WAVE) << 32) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)mdl, src, WAVE);
        const uint32_t s_mn = (uint32_t)__shfl((int)mdl_n, src, WAVE);
        const unsigned long long p_mdl = pl < 64u ? s_mdl : last_mdl;
        const uint32_t p_mn = pl < 64u ? s_mn : last_mdl_n;
        const bool new_model = row && has_p && (p_mdl != mdl || p_mn != mdl_n);
        if (__any(new_model)) {
            auto plain_number = [](unsigned long long v, uint32_t n) -> bool {          // digits only, no leading zero: name <-> number one to one
                if (n == 0u || n > 8u) return false;
                bool ok = n == 1u || (v & 0xffull) != '0';
                for (uint32_t i = 0; i < n; i++) { const uint32_t c = (uint32_t)(v >> (8u * i)) & 0xffu; ok = ok && c >= '0' && c <= '9'; }
                return ok;
            };
            auto rises = [](unsigned long long a, uint32_t na, unsigned long long b, uint32_t nb) -> bool {   // number(a) < number(b)
                if (na != nb) return na < nb;
                return __builtin_bswap64(a) < __builtin_bswap64(b);            // the first character is the low byte: most significant after the swap
            };
            if (new_model && !(plain_number(p_mdl, p_mn) && plain_number(mdl, mdl_n) && rises(p_mdl, p_mn, mdl, mdl_n))) rbad = true;
        }
        // (file order is the reader's order as long as, inside a run of one chain name, every new residue has a larger (number,
        //  insertion code) than the one before it: find_or_add_residue then never finds an earlier residue to append to)
        if (row && !rbad && has_p && !new_model && p_ch == ch && !(p_num == num && p_ic == ic && p_rn == rn) && !(num > p_num || (num == p_num && ic > p_ic))) rbad = true;
        if (__any(row && rbad)) { dead = true; break; }
        const bool keep = row && !(has_p && p_an == an);
        const unsigned long long m_keep = __ballot(keep);
        if (keep) {
            const size_t o = (size_t)A0 + kept + (uint32_t)__builtin_popcountll(m_keep & ((1ull << lane) - 1ull));   // <= `at`: never a record still to be read
            T.name[o] = an; T.resn[o] = rn; T.serial[o] = serial; T.resseq[o] = num;
            T.x[o] = x; T.y[o] = y; T.z[o] = z; T.b[o] = bf; T.chain[o] = ch;
            T.acode[o] = (uint8_t)atom_code_of(an);
            T.rcode[o] = (int8_t)res_code_of(rn);
        }
        kept += (uint32_t)__builtin_popcountll(m_keep);
        const int hl = 63 - __builtin_clzll(rowmask);
        last_name = (uint32_t)__builtin_amdgcn_readlane((int)an, hl); last_comp = (uint32_t)__builtin_amdgcn_readlane((int)rn, hl); last_ch = (uint32_t)__builtin_amdgcn_readlane((int)ch, hl);
        last_num = __builtin_amdgcn_readlane(num, hl); last_ic = (uint32_t)__builtin_amdgcn_readlane((int)ic, hl); have_last = true;
        last_mdl = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mdl >> 32), hl) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mdl, hl);
        last_mdl_n = (uint32_t)__builtin_amdgcn_readlane((int)mdl_n, hl);
    }
    if (lane == 0 && !dead && kept != 0u) {
        n_kept[f] = kept;
        file_status[f] = FCZ_OK;
    }
}

}  // namespace fcz
