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
//     3. the marked rows are parsed by their lanes: columns by token ordinal, decimals as (double)digits / 10^f (exact operands,
//        IEEE division = strtod's result for <= 15 digits = gemmi's fast_float), names packed, codes through the LDS hash,
//        removeAlternativePosition by the keep rule of the PDB kernel, kept atoms appended to the file's scratch slice.
//   The same k_ingest_frags / k_ingest_fill then build the batch.
//
// The device never guesses. It takes the shape every predicted-structure file has -- one data_ block, items one per line (or
// a tag line followed by its value line / text field), loop rows of whole lines, `_atom_site` rows of exactly one line each
// without quotes, single-character chain names, integer residue numbers without insertion codes, one model, residues in
// rising order inside a chain run -- and hands EVERYTHING else back to the host reader (save_ frames, global_ / stop_, several
// blocks, comments after values, quoted atom names, multi-letter chains, several models, '?' coordinates, exponents, lines of
// more than 255 characters, bytes outside printable ASCII, ...), which restates gemmi rule for rule and is held to the live
// reference by fuzzing. tests/test_gpu_ingest.py holds this kernel to that reader on mutated files: it never builds a
// different batch and never takes a file the reader fails.
#pragma once
#include "fcz_ingest.h"

namespace fcz {

constexpr int CIF_MAXTOK = 32;                    // columns of an _atom_site row this path takes (AFDB: 25, PDB archive: 21-26)
constexpr int CIF_MAXLINE = 255;                  // characters of a lexed line
constexpr int CIF_NCOL = 23;
constexpr int CIF_TAGSET = 2048;                  // slots of the duplicate-tag table (a block has a few hundred tags)

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
    uint32_t line_end[IG_LINES];
    uint32_t akey[64], rkey[32];
    uint8_t aval[64], rval[32];
    uint32_t tagset[CIF_TAGSET];
    uint8_t tok_s[WAVE][CIF_MAXTOK], tok_e[WAVE][CIF_MAXTOK];
    int8_t pos[CIF_NCOL + 1];
};

__device__ __forceinline__ bool cif_is_ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r'; }
__device__ __forceinline__ uint32_t cif_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32u : c; }

__global__ __launch_bounds__(WAVE) void k_ingest_parse_cif(const uint8_t* __restrict__ text, const uint64_t* __restrict__ file_off, uint32_t n_files,
                                                           const uint64_t* __restrict__ abase, ingest_scratch T,
                                                           uint8_t* __restrict__ titles, uint32_t* __restrict__ title_len,
                                                           uint32_t* __restrict__ n_kept, int32_t* __restrict__ file_status) {
    __shared__ cif_lds S;
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_files) return;
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
    // name -> code tables (as in k_ingest_parse), the tag table, the column map
    S.akey[lane] = 0; if (lane < 32) S.rkey[lane] = 0;
    for (int k = lane; k < CIF_TAGSET; k += WAVE) S.tagset[k] = 0;
    if (lane <= CIF_NCOL) S.pos[lane] = -1;
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
    const uint32_t cap = (uint32_t)(abase[f + 1] - A0);
    uint8_t* tbuf = titles + (size_t)f * IG_TITLE_CAP;
    bool dead = false;                     // the file goes back to the host (uniform)
    // ---- automaton state (uniform) ----
    enum { CX_START, CX_NONE, CX_AFTER_TAG, CX_LOOP_HDR, CX_LOOP_BODY };
    int ctx = CX_START;
    uint32_t ntags = 0; unsigned long long nvals = 0;
    bool in_as = false, other_cat = false, as_done = false, as_ready = false;   // the open loop is the _atom_site loop / holds other tags too; the loop was read; its column map is final
    bool pending_special = false;          // the tag waiting for its value on a later line is one whose value this path must see
    bool in_text = false;                  // inside a text field
    uint32_t tlen = 0; bool have_title = false;
    int kAsym = -1, kComp = -1, kAtom = -1;
    // ---- row state (uniform) ----
    uint32_t kept = 0;
    bool have_last = false; uint32_t last_name = 0, last_comp = 0, last_ch = 0; int32_t last_num = 0;
    unsigned long long model0 = 0; bool have_model = false;
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
        for (uint32_t s = (h * 0x9E3779B1u) >> 21, n = 0; n < (uint32_t)CIF_TAGSET; s = (s + 1) & (CIF_TAGSET - 1), n++) {
            const uint32_t v = S.tagset[s];
            if (v == h) return true;
            if (v == 0) { if (lane == 0) S.tagset[s] = h; __builtin_amdgcn_wave_barrier(); return false; }
        }
        return true;                                    // table full: to the host
    };

    // ---- one step: up to 64 lines, lane = line. ls / le file-relative, lo = offset of the line's first byte in S.buf or -1 ----
    auto do_lines = [&](bool on, uint64_t ls, uint64_t le, int lo) {
        if (dead) return;
        uint32_t len = on ? (uint32_t)(le - ls) : 0u;
        // ---- 1. lexing, every lane its own line ----
        const uint32_t c_first = !on || len == 0 ? (uint32_t)'\n' : (lo >= 0 ? (uint32_t)S.buf[lo] : (uint32_t)base[ls]);
        const bool semi = on && c_first == ';';
        const unsigned long long m_semi = __ballot(semi);
        const bool text_before = in_text ^ ((__builtin_popcountll(m_semi & ((1ull << lane) - 1ull)) & 1) != 0);   // inside a text field when this line starts
        int cls = CL_BLANK; uint32_t ntok = 0, tag_hash = 0, kw_first = 0; bool bad = false;
        if (on && text_before) {
            if (semi) {                                                    // the closing line: nothing but blanks may follow the ';'
                if (lo < 0 || len > (uint32_t)CIF_MAXLINE) bad = true;
                else for (uint32_t i = 1; i < len; i++) if (!cif_is_ws(S.buf[lo + i])) bad = true;
            }
        } else if (on && semi) cls = CL_TEXT_OPEN;
        else if (on && len) {
            if (lo < 0 || len > (uint32_t)CIF_MAXLINE) {
                // only a comment line may be that long here
                bool comment = false;
                for (uint32_t i = 0; i < len; i++) { const uint32_t c = lo >= 0 ? (uint32_t)S.buf[lo + i] : (uint32_t)base[ls + i]; if (cif_is_ws(c)) continue; comment = c == '#'; break; }
                bad = !comment;
            } else {
                bool in_tok = false, quoted = false, is_tag = false; uint32_t quote = 0, ts = 0, tl = 0; unsigned long long acc = 0; uint32_t h = 2166136261u;
                auto close = [&](uint32_t end) {
                    if (ntok < (uint32_t)CIF_MAXTOK) { S.tok_s[lane][ntok] = (uint8_t)ts; S.tok_e[lane][ntok] = (uint8_t)end; }
                    ntok++; in_tok = false; quote = 0;
                };
                for (uint32_t i = 0; i < len; i++) {
                    const uint32_t c = S.buf[lo + i];
                    if ((c < 0x20u && c != '\t' && c != '\r') || c > 0x7eu) { bad = true; break; }
                    if (!in_tok) {
                        if (cif_is_ws(c)) continue;
                        if (c == '#') { if (ntok != 0) bad = true; break; }          // a comment line; a comment after tokens goes to the host
                        in_tok = true; ts = i; tl = 0; acc = 0; quoted = c == '\'' || c == '"'; quote = quoted ? c : 0u;
                        is_tag = c == '_';
                        if (c == '$' || (is_tag && ntok != 0)) bad = true;            // a frame reference; a tag after something else on its line
                        if (is_tag) h = 2166136261u;
                    } else if (quote) {
                        if (c == quote) { const uint32_t nx = i + 1 < len ? (uint32_t)S.buf[lo + i + 1] : (uint32_t)' '; if (cif_is_ws(nx) || nx == '#') { close(i + 1); continue; } }
                    } else if (cif_is_ws(c)) { close(i); continue; }
                    if (in_tok && !quoted) {
                        // reserved words open no value (CifScanner::value): data_ loop_ stop_ save_ global_, any case
                        if (tl < 7u) acc |= (unsigned long long)cif_lower(c) << (8 * tl);
                        tl++;
                        if (tl == 5u) {
                            const unsigned long long a5 = acc & 0xffffffffffull;
                            const uint32_t kw = a5 == 0x5f61746164ull ? 1u : a5 == 0x5f706f6f6cull ? 2u : a5 == 0x5f706f7473ull ? 5u : a5 == 0x5f65766173ull ? 4u : 0u;   // "data_" "loop_" "stop_" "save_"
                            if (kw) { if (ntok == 0) kw_first = kw; else bad = true; }
                        }
                        if (tl == 7u && (acc & 0xffffffffffffffull) == 0x5f6c61626f6c67ull) bad = true;                                                                  // "global_"
                        if (is_tag && ntok == 0) h = (h ^ cif_lower(c)) * 16777619u;
                    }
                }
                if (in_tok) { if (quote) bad = true; else close(len); }
                if (!bad) {
                    const uint32_t l0 = ntok ? (uint32_t)S.tok_e[lane][0] - (uint32_t)S.tok_s[lane][0] : 0u;
                    const bool tag0 = ntok && S.buf[lo + S.tok_s[lane][0]] == '_';
                    if (ntok == 0) cls = CL_BLANK;
                    else if (kw_first == 2u) { cls = CL_LOOP; if (ntok != 1 || l0 != 5u) bad = true; }
                    else if (kw_first == 1u) { cls = CL_DATA; if (ntok != 1 || l0 <= 5u) bad = true; }
                    else if (kw_first) bad = true;                                  // save_, stop_ as the first token
                    else if (tag0) { cls = ntok == 1 ? CL_TAG : CL_PAIR; if (ntok > 2 || l0 < 2u) bad = true; tag_hash = h; }
                    else cls = CL_VALUES;
                }
            }
        }
        if (bad) cls = CL_BAD;
        if (__any(cls == CL_BAD)) { dead = true; return; }
        in_text = in_text ^ ((__builtin_popcountll(m_semi) & 1) != 0);
        // ---- 2. the grammar over the step's lines, in order (uniform) ----
        unsigned long long rowmask = 0;
        const unsigned long long m_on = __ballot(on);
        const unsigned long long m_rowlike = __ballot(on && cls == CL_VALUES && ntok == ntags);
        if (ctx == CX_LOOP_BODY && in_as && as_ready && m_on && m_rowlike == m_on) {
            rowmask = m_on; nvals += (unsigned long long)ntags * (unsigned)__builtin_popcountll(m_on);      // a step of nothing but rows
        } else {
            auto end_item = [&]() {                                                   // what stands before a new tag / loop_ must be complete
                if (ctx == CX_LOOP_BODY) { if (ntags == 0 || nvals % ntags != 0) dead = true; if (in_as) as_done = true; ctx = CX_NONE; }
                else if (ctx != CX_NONE) dead = true;
            };
            for (unsigned long long m = m_on & __ballot(cls != CL_BLANK); m && !dead; m &= m - 1) {
                const int l = __builtin_ctzll(m);
                const int c = __shfl(cls, l, WAVE);
                const uint32_t k = (uint32_t)__shfl((int)ntok, l, WAVE);
                const int llo = __shfl(lo, l, WAVE);
                if (c == CL_DATA) { if (ctx != CX_START) dead = true; ctx = CX_NONE; continue; }      // a second block: to the host
                if (ctx == CX_START) { dead = true; break; }
                if (c == CL_LOOP) { end_item(); ctx = CX_LOOP_HDR; ntags = 0; nvals = 0; in_as = false; other_cat = false; continue; }
                if (c == CL_TAG || c == CL_PAIR) {
                    const uint32_t t0 = S.tok_s[l][0], tn = (uint32_t)S.tok_e[l][0] - t0;
                    if (tag_seen((uint32_t)__shfl((int)tag_hash, l, WAVE))) { dead = true; break; }
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
                        } else { if (in_as) { dead = true; break; } other_cat = true; }
                        ntags++;
                        if (in_as && ntags > (uint32_t)CIF_MAXTOK) { dead = true; break; }
                        continue;
                    }
                    end_item();
                    if (dead) break;
                    if (cat_as) { dead = true; break; }                              // _atom_site items as pairs (a one-atom file): to the host
                    // values this path must see: the title, the angles of the zero-angle rule
                    const bool ang = cat_cell && (tn == 17u && eq_lower(llo, t0, "_cell.angle_alpha", 17)) | (tn == 16u && eq_lower(llo, t0, "_cell.angle_beta", 16));
                    if (is_entry && !eq_exact(llo, t0, "_entry.id", 9)) { dead = true; break; }   // (a pair is looked up by its exact spelling)
                    if (c == CL_TAG) { pending_special = is_entry || ang; ctx = CX_AFTER_TAG; continue; }
                    const uint32_t v0 = S.tok_s[l][1], vn = (uint32_t)S.tok_e[l][1] - v0;
                    const uint32_t vc = S.buf[llo + v0];
                    if (is_entry) {
                        if (vc == '\'' || vc == '"' || (vn == 1u && (vc == '?' || vc == '.')) || vn > (uint32_t)IG_TITLE_CAP) { dead = true; break; }
                        if (lane < (int)vn) tbuf[lane] = S.buf[llo + v0 + lane];
                        for (uint32_t i = WAVE + lane; i < vn; i += WAVE) tbuf[i] = S.buf[llo + v0 + i];
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
                nvals += kv;
                if (in_as) { if (c != CL_VALUES || k != ntags) { dead = true; break; } rowmask |= 1ull << l; }
            }
        }
        if (dead) return;
        // ---- 3. the _atom_site rows of the step, every lane its own ----
        if (rowmask == 0ull) return;
        const bool row = ((rowmask >> lane) & 1ull) != 0;
        bool rbad = false;
        uint32_t an = 0, rn = 0, ch = ' '; int32_t serial = 0, num = 0; float x = 0.f, y = 0.f, z = 0.f, bf = 0.f; unsigned long long mdl = 0;
        if (row) {
            auto tb = [&](int col, uint32_t i) -> uint32_t { return S.buf[lo + S.tok_s[lane][col] + i]; };
            auto tl = [&](int col) -> uint32_t { return (uint32_t)S.tok_e[lane][col] - (uint32_t)S.tok_s[lane][col]; };
            auto is_null = [&](int col) -> bool { return tl(col) == 1u && (tb(col, 0) == '?' || tb(col, 0) == '.'); };
            auto integer = [&](int col, int32_t* out) -> bool {                      // [+-]digits, at most nine of them
                const uint32_t n = tl(col); uint32_t i = 0; bool neg = false;
                if (n && (tb(col, 0) == '-' || tb(col, 0) == '+')) { neg = tb(col, 0) == '-'; i = 1; }
                if (i >= n || n - i > 9u) return false;
                uint32_t v = 0;
                for (; i < n; i++) { const uint32_t d = tb(col, i) - '0'; if (d > 9u) return false; v = v * 10u + d; }
                *out = neg ? -(int32_t)v : (int32_t)v;
                return true;
            };
            auto decimal = [&](int col, float* out) -> bool {                        // -digits.digits, at most 15 digits: cif::as_number's fast path
                const uint32_t n = tl(col); uint32_t i = 0; bool neg = false;
                if (n && tb(col, 0) == '-') { neg = true; i = 1; }
                unsigned long long m = 0; uint32_t nd = 0, nf = 0; bool point = false;
                for (; i < n; i++) {
                    const uint32_t c = tb(col, i);
                    if (c == '.') { if (point) return false; point = true; continue; }
                    const uint32_t d = c - '0';
                    if (d > 9u) return false;
                    m = m * 10ull + d; nd++; if (point) nf++;
                }
                if (nd == 0u || nd > 15u) return false;
                const double v = (double)m / cif_pow10[nf];
                *out = (float)(neg ? -v : v);
                return true;
            };
            auto pack = [&](int col, uint32_t* out) -> bool {                        // a name of one to four characters
                const uint32_t n = tl(col);
                if (n == 0u || n > 4u || is_null(col)) return false;
                uint32_t w = 0;
                for (uint32_t i = 0; i < n; i++) w |= tb(col, i) << (8 * i);
                *out = w;
                return true;
            };
            // no quotes anywhere in the row (a quoted value's content is what the reader takes: the host strips them)
            for (uint32_t t = 0; t < ntags; t++) { const uint32_t c = tb((int)t, 0); if (c == '\'' || c == '"') rbad = true; }
            const int pId = S.pos[CK_ID], pAlt = S.pos[CK_ALT], pIns = S.pos[CK_INS], pLs = S.pos[CK_LSEQ], pCh = S.pos[CK_CHARGE], pMo = S.pos[CK_MODEL];
            rbad = rbad | !integer(pId, &serial) | !integer(S.pos[CK_ASEQ], &num);
            rbad = rbad | !decimal(S.pos[CK_X], &x) | !decimal(S.pos[CK_Y], &y) | !decimal(S.pos[CK_Z], &z) | !decimal(S.pos[CK_B], &bf);
            rbad = rbad | !pack(S.pos[kAtom], &an) | !pack(S.pos[kComp], &rn);
            if (tl(S.pos[kAsym]) != 1u || is_null(S.pos[kAsym])) rbad = true; else ch = tb(S.pos[kAsym], 0);
            if (!is_null(pAlt) && tl(pAlt) != 1u) rbad = true;
            if (pIns >= 0 && !is_null(pIns)) rbad = true;
            int32_t dummy;
            if (pLs >= 0 && !is_null(pLs) && !integer(pLs, &dummy)) rbad = true;
            if (pCh >= 0 && !is_null(pCh) && !integer(pCh, &dummy)) rbad = true;
            if (pMo >= 0) { const uint32_t n = tl(pMo); if (n > 8u) rbad = true; else for (uint32_t i = 0; i < n; i++) mdl |= (unsigned long long)tb(pMo, i) << (8 * i); }
        }
        // one model; residues of a chain run in rising order (the reader regroups anything else); keep rule of removeAlternativePosition
        const uint32_t pl = ig_prev_lane(rowmask, lane);
        {
            const int src = pl < 64u ? (int)pl : 0;
            const int32_t s_num = __shfl(num, src, WAVE);
            const uint32_t s_rn = (uint32_t)__shfl((int)rn, src, WAVE), s_ch = (uint32_t)__shfl((int)ch, src, WAVE), s_an = (uint32_t)__shfl((int)an, src, WAVE);
            const bool has_p = pl < 64u ? true : have_last;
            const int32_t p_num = pl < 64u ? s_num : last_num;
            const uint32_t p_rn = pl < 64u ? s_rn : last_comp, p_ch = pl < 64u ? s_ch : last_ch, p_an = pl < 64u ? s_an : last_name;
            const unsigned long long m0 = have_model ? model0 : (unsigned long long)__shfl((long long)mdl, __builtin_ctzll(rowmask), WAVE);
            if (row && mdl != m0) rbad = true;
            if (row && !rbad && has_p && p_ch == ch && !(p_num == num && p_rn == rn) && !(num > p_num)) rbad = true;
            if (__any(row && rbad)) { dead = true; return; }
            if (!have_model) { model0 = m0; have_model = true; }
            const bool keep = row && !(has_p && p_an == an);
            const unsigned long long m_keep = __ballot(keep);
            const uint32_t n_new = (uint32_t)__builtin_popcountll(m_keep);
            if (kept + n_new > cap) { dead = true; return; }
            if (keep) {
                const size_t o = (size_t)A0 + kept + (uint32_t)__builtin_popcountll(m_keep & ((1ull << lane) - 1ull));
                T.name[o] = an; T.resn[o] = rn; T.serial[o] = serial; T.resseq[o] = num;
                T.x[o] = x; T.y[o] = y; T.z[o] = z; T.b[o] = bf; T.chain[o] = (uint8_t)ch;
                T.acode[o] = (uint8_t)atom_code_of(an);
                T.rcode[o] = (int8_t)res_code_of(rn);
            }
            kept += n_new;
            const int hl = 63 - __builtin_clzll(rowmask);
            last_name = (uint32_t)__shfl((int)an, hl, WAVE); last_comp = (uint32_t)__shfl((int)rn, hl, WAVE); last_ch = (uint32_t)__shfl((int)ch, hl, WAVE);
            last_num = __shfl(num, hl, WAVE); have_last = true;
        }
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
        for (uint32_t r0 = 0; (r0 < total || r0 == 0) && !dead; r0 += IG_LINES) {
            uint32_t o = ord;
            for (unsigned long long m = nlm; m; m &= m - 1) {
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                if (o >= r0 && o < r0 + (uint32_t)IG_LINES) S.line_end[o - r0] = 64u * (uint32_t)lane + bit;
                o++;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_here = total - r0 < (uint32_t)IG_LINES ? total - r0 : (uint32_t)IG_LINES;
            for (uint32_t k0 = 0; k0 < n_here && !dead; k0 += WAVE) {
                const uint32_t k = k0 + (uint32_t)lane;
                const bool on = k < n_here;
                const uint64_t le = on ? c0 + S.line_end[k] : 0;
                const uint64_t ls = on ? (k == 0 ? line_start : c0 + S.line_end[k - 1] + 1) : 0;
                const long long rel = (long long)ls - (long long)c0;
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
        if (!as_done || !have_title || kept == 0) dead = true;              // no atoms, no title: the host reports what the reference reports
    }
    if (lane == 0 && !dead) {
        title_len[f] = tlen;
        n_kept[f] = kept;
        file_status[f] = FCZ_OK;
    }
}

}  // namespace fcz
