// fcz_inflate.h -- gzip members -> text in HBM on the device (the `.pdb.gz` / `.cif.gz` files AFDB ships).
//
// What the reference does with zlib before its reader sees a byte: gemmi::MaybeGzipped for files (lib/gemmi/gz.hpp:105-133: the
// buffer is sized from ISIZE, the last four bytes of the file), uncompressBuffer for database / tar entries
// (src/structure_reader.cpp:156-203: inflateInit2(15 | 32) + inflate()). zlib itself is not vendored in the reference tree (it links
// the system's: zlib 1.2.11 in this image); the format is RFC 1952 (gzip member) around RFC 1951 (DEFLATE), restated here.
//
//   k_inflate   wavefront = member. The DEFLATE symbol stream is serial, so ONE symbol is decoded per step with wave-uniform
//               control flow (the bit buffer lives in SGPRs: the compressed bytes are fetched 256 at a time, one dword per lane,
//               and enter the bit buffer through v_readlane), and everything around the serial step is lane-parallel:
//                 * code-length sets -> canonical codes: counting sort by ballots (lane = symbol), per-length first code / count /
//                   offset held by lane = length; the 512- / 256-entry fast tables are filled lane = entry;
//                 * a code longer than the fast table is decoded by ONE ballot: lane l tests "is the l-bit prefix a code of
//                   length l" (canonical codes: first[l] <= prefix < first[l] + count[l]);
//                 * LZ77 copies move up to 64 bytes per LDS round trip through a 16 KB ring of the newest output (in-order LDS
//                   pipeline: a later chunk reads what an earlier chunk wrote); matches farther back than the ring read the text
//                   the wavefront already flushed to HBM;
//                 * the ring leaves in rounds of 4 KB: 16-byte coalesced stores, and the CRC-32 of the round by 64 lanes x 64
//                   bytes (table in LDS) combined with carry-less multiplications by x^(8 n) mod P (zlib's crc32_combine).
//               Output goes straight to text[text_off[f] ..) in the layout fcz_ingest_pdb_dev takes. The member's CRC-32 and ISIZE are
//               verified here. The device never guesses: a member it does not decode to the last bit exactly as zlib's inflate()
//               would accept it (header flags it does not read, a code set zlib rejects or that is incomplete, a distance before the
//               start, output that differs from text_off's size, bytes left between the final block and the trailer, CRC / ISIZE
//               mismatch ...) gets a non-zero status, its text range is blanked, and the FILE goes back to the caller's zlib.
#pragma once
#include "fcz_kernels.h"

#ifndef FCZ_INFLATE_RING_BITS
#define FCZ_INFLATE_RING_BITS 13
#endif

namespace fcz {
namespace inflate {

constexpr uint32_t RING = 1u << FCZ_INFLATE_RING_BITS, RMASK = RING - 1u;
constexpr uint32_t ROUND = 4096;                 // bytes per flush round (64 lanes x 64 bytes)
#ifndef FCZ_INFLATE_LBITS
#define FCZ_INFLATE_LBITS 9
#endif
constexpr int LBITS = FCZ_INFLATE_LBITS, DBITS = 8;              // index bits of the literal/length and distance fast tables
constexpr uint32_t CRC_POLY = 0xEDB88320u;       // CRC-32 (RFC 1952 section 8), reflected

// status of a member (0 = inflated and verified); everything else: the caller's zlib decides (fcz_hip.h FCZ_INFLATE_*)
constexpr int32_t ST_OK = 0, ST_HEADER = 1, ST_BLOCK = 2, ST_CODE = 3, ST_SIZE = 4, ST_INPUT = 5, ST_CHECK = 6;
#ifdef FCZ_INFLATE_DEBUG
#define INF_FAIL(code) ((code) | (__LINE__ << 8))    // where the member was refused (tests print it)
#else
#define INF_FAIL(code) (code)
#endif

// ---- CRC-32 arithmetic in the reflected representation (x^0 = 0x80000000; multiplying by x = one shift right) ----
constexpr uint32_t crc_xstep(uint32_t v) { return (v >> 1) ^ ((v & 1u) ? CRC_POLY : 0u); }
constexpr uint32_t crc_mul_c(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int k = 31; k >= 0; k--) { if ((a >> k) & 1u) p ^= b; b = crc_xstep(b); }
    return p;
}
struct crc_consts {
    uint32_t tab[256];     // the byte table
    uint32_t xp8[65];      // x^(8 n), n = 0 .. 64
    uint32_t k64[64];      // x^(512 (63 - j)): what lane j's 64 bytes of a full round are multiplied by
    uint32_t x4096;        // x^(8 * 4096): what the running CRC is multiplied by per full round
    constexpr crc_consts() : tab(), xp8(), k64(), x4096(0) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = crc_xstep(c); tab[i] = c; }
        xp8[0] = 0x80000000u;
        for (int n = 0; n < 64; n++) { uint32_t v = xp8[n]; for (int k = 0; k < 8; k++) v = crc_xstep(v); xp8[n + 1] = v; }
        k64[63] = 0x80000000u;
        for (int j = 62; j >= 0; j--) k64[j] = crc_mul_c(k64[j + 1], xp8[64]);
        x4096 = crc_mul_c(k64[0], xp8[64]);
    }
};
__device__ const crc_consts g_crc = crc_consts();

__device__ __forceinline__ uint32_t crc_mul(uint32_t a, uint32_t b) {      // a * b mod P, branch-free (zlib crc32.c multmodp)
    uint32_t p = 0;
#pragma unroll
    for (int k = 31; k >= 0; k--) {
        p ^= b & (0u - ((a >> k) & 1u));
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
    }
    return p;
}

// ---- wave helpers ----
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// LDS traffic of one wavefront is processed in issue order; this only keeps the compiler from moving accesses across
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v ^= (uint32_t)__shfl_xor((int)v, d, WAVE);
    return v;
}

// inclusive running maximum over the wavefront on the DPP network (zeros shift in)
__device__ __forceinline__ uint32_t wave_incl_max_dpp(uint32_t v) {
    v = umax(v, dpp_u32_or0<0x111, 0xf>(v));   // row_shr:1
    v = umax(v, dpp_u32_or0<0x112, 0xf>(v));   // row_shr:2
    v = umax(v, dpp_u32_or0<0x114, 0xf>(v));   // row_shr:4
    v = umax(v, dpp_u32_or0<0x118, 0xf>(v));   // row_shr:8
    v = umax(v, dpp_u32_or0<0x142, 0xa>(v));   // row_bcast:15 into rows 1 and 3
    v = umax(v, dpp_u32_or0<0x143, 0xc>(v));   // row_bcast:31 into rows 2 and 3
    return v;
}

struct lds_t {
    alignas(16) uint8_t ring[RING];        // the newest RING bytes of output, byte p of the text at (p & RMASK)
    uint32_t lit_tab[1 << LBITS];          // fast tables (word layout: resolve_lit / resolve_dist below; 0: not in the table)
    uint32_t dist_tab[1 << DBITS];
    uint32_t crc_tab[256];
    uint32_t in_ring[256];                 // the compressed stream around the read position: 256-byte block b at dwords (b & 3) * 64 ..
    uint16_t lit_sorted[288];              // symbols ordered by (code length, symbol)
    uint16_t dist_sorted[32];
    uint16_t cl_sorted[32];
    uint8_t lens[320 + 64];                // code lengths of the literal/length then the distance alphabet
    uint8_t own[64];                       // a window's text: byte -> lane + 1 of the symbol that starts there
};

// per-length view of a canonical code, lane l = code length l (lanes 1 .. 15)
struct canon { uint32_t first, count, off; };

// Fast-table words are laid out so that the scalar unit reads them without unpacking: bits [4:0] = code length L (bit 5 is zero: the
// word itself is the shift count of s_lshr_b64) and bits [22:16] = number of extra bits, which makes the word the control operand
// of s_bfe_u32 (offset [4:0], width [22:16], other bits ignored): the extra bits that follow the code come out of the low dword
// of the bit buffer in ONE instruction. A word of 0 = "not in this table" (longer code, or no code).
//   literal/length (RFC 1951 section 3.2.5): [7:6] kind (0 literal, 1 length, 2 end of block, 3 invalid: 286, 287),
//                                            [15:8] the literal, [31:23] base length
__device__ __forceinline__ uint32_t resolve_lit(uint32_t sym, uint32_t L) {
    uint32_t kind, extra = 0, base = 0, lit = 0;
    if (sym < 256u) { kind = 0; lit = sym; }
    else if (sym == 256u) { kind = 2; }
    else if (sym > 285u) { kind = 3; }
    else {
        const uint32_t i = sym - 257u;
        kind = 1;
        if (i < 8u) base = i + 3u;
        else if (i == 28u) base = 258u;
        else { extra = (i - 4u) >> 2; base = ((4u + (i & 3u)) << extra) + 3u; }
    }
    return L | (kind << 6) | (lit << 8) | (extra << 16) | (base << 23);
}
//   distance: [6] invalid (30, 31), [9:8] m with base distance = (m << extra) + 1  (symbols 0 .. 3: m = symbol, no extra bits;
//             from 4 on: m = 2 + (symbol & 1), extra = symbol / 2 - 1)
__device__ __forceinline__ uint32_t resolve_dist(uint32_t sym, uint32_t L) {
    uint32_t extra = 0, m, bad = 0;
    if (sym < 4u) m = sym;
    else if (sym < 30u) { extra = (sym >> 1) - 1u; m = 2u + (sym & 1u); }
    else { bad = 1; m = 0; }
    return L | (bad << 6) | (m << 8) | (extra << 16);
}
__device__ __forceinline__ uint32_t s_bfe(uint32_t src, uint32_t ctl) {      // src[ctl[4:0] +: ctl[22:16]] on the scalar unit
    uint32_t r;
    asm("s_bfe_u32 %0, %1, %2" : "=s"(r) : "s"(src), "s"(ctl) : "scc");
    return r;
}

// code lengths lens[0 .. nsym) (LDS) -> sorted[] and the per-length view. Returns 0: complete code, 1: incomplete, 2: over-subscribed
// (zlib inftrees.c:116-123); *nonzero = symbols that have a code, *maxlen = longest code.
template <int CHUNKS>
__device__ __forceinline__ int build_canon(const uint8_t* lens, uint32_t nsym, uint16_t* sorted, canon& cn, uint32_t lane,
                                           uint32_t* nonzero, uint32_t* maxlen) {
    uint32_t cnt = 0;                        // lane l: symbols of length l so far
    uint32_t rank[CHUNKS], mylen[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const uint32_t s = (uint32_t)c * 64u + lane;
        const uint32_t len = s < nsym ? (uint32_t)lens[s] : 0u;
        mylen[c] = len;
        uint32_t r = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 15u; l++) {
            const uint64_t b = __ballot(len == l);
            const uint32_t before = rdl(cnt, l);
            if (len == l) r = before + mbcnt(b);
            cnt += (lane == l) ? (uint32_t)__popcll(b) : 0u;
        }
        rank[c] = r;
    }
    uint32_t code = 0, off = 0, vf = 0, vo = 0, mx = 0;
    int left = 1; bool over = false;
#pragma unroll
    for (uint32_t l = 1; l <= 15u; l++) {
        const uint32_t c = rdl(cnt, l);
        left = (left << 1) - (int)c;
        if (left < 0) { over = true; left = 0; }
        vf = (lane == l) ? code : vf;
        vo = (lane == l) ? off : vo;
        code = (code + c) << 1;
        off += c;
        if (c) mx = l;
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const uint32_t base = (uint32_t)__shfl((int)vo, (int)mylen[c], WAVE);
        if (mylen[c]) sorted[base + rank[c]] = (uint16_t)((uint32_t)c * 64u + lane);
    }
    wave_fence();
    cn.first = vf; cn.count = cnt; cn.off = vo;
    *nonzero = off; *maxlen = mx;
    return over ? 2 : (left > 0 ? 1 : 0);
}

// one code of at most 15 bits by ballot: `peek` = the next 15 bits of the stream (first bit = bit 0). Returns the index into
// sorted[] (or -1: no code of this set starts like this) and the code's length.
__device__ __forceinline__ int canon_decode(uint32_t peek, const canon& cn, uint32_t lane, uint32_t* L) {
    const uint32_t rev = __brev(peek) >> 17;                      // the 15 bits, first bit of the stream on top
    const uint32_t d = (rev >> ((15u - lane) & 15u)) - cn.first;
    const bool hit = (lane - 1u) < 15u && d < cn.count;
    const uint64_t m = __ballot(hit);
    if (!m) return -1;
    const uint32_t l = (uint32_t)__builtin_ctzll(m);
    *L = l;
    return (int)rdl(cn.off + d, l);
}

template <int BITS, bool DIST>
__device__ __forceinline__ void fill_table(uint32_t* tab, const uint16_t* sorted, const canon& cn, uint32_t lane) {
    uint32_t f[BITS + 1], n[BITS + 1], o[BITS + 1];
#pragma unroll
    for (int l = 1; l <= BITS; l++) { f[l] = rdl(cn.first, l); n[l] = rdl(cn.count, l); o[l] = rdl(cn.off, l); }
    for (uint32_t i = lane; i < (1u << BITS); i += 64u) {
        const uint32_t rev = __brev(i) >> (32 - BITS);
        uint32_t L = 0, idx = 0;
#pragma unroll
        for (int l = BITS; l >= 1; l--) {                          // (a prefix code: at most one length matches)
            const uint32_t d = (rev >> (BITS - l)) - f[l];
            if (d < n[l]) { L = (uint32_t)l; idx = o[l] + d; }
        }
        uint32_t e = 0;
        if (L) { const uint32_t sym = sorted[idx]; e = DIST ? resolve_dist(sym, L) : resolve_lit(sym, L); }
        tab[i] = e;
    }
    wave_fence();
}

// ---- the compressed bytes: 256 at a time in one VGPR (lane k = dword k of the block), the next block already in flight ----
struct bitreader {
    const uint8_t* base;      // 4-byte aligned address at or below the member's first byte
    uint32_t avail;           // bytes from base to the member's end
    uint32_t lane;
    uint64_t bb; uint32_t bn; // bit buffer (wave-uniform): the next bn bits of the stream, first bit = bit 0; zeros beyond the end
    uint32_t w;               // next dword (from base) to enter the bit buffer
    uint32_t cur, nxt;        // this lane's dword of the 256-byte block dword w lies in / of the block after it

    __device__ __forceinline__ uint32_t load_block(uint32_t b) const {
        const uint32_t o = (b * 64u + lane) * 4u;
        if (o + 4u <= avail) return *reinterpret_cast<const uint32_t*>(base + o);
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4u; k++) if (o + k < avail) v |= (uint32_t)base[o + k] << (8u * k);
        return v;
    }
    __device__ __forceinline__ uint32_t word() {                    // dword w, then w++
        const uint32_t d = rdl(cur, w & 63u);
        w++;
        if ((w & 63u) == 0u) { cur = nxt; nxt = load_block((w >> 6) + 1u); }
        return d;
    }
    __device__ __forceinline__ void seek(uint32_t byte_off) {       // next bit = bit 0 of the byte at base + byte_off
        w = byte_off >> 2;
        cur = load_block(w >> 6); nxt = load_block((w >> 6) + 1u);
        const uint32_t sk = (byte_off & 3u) * 8u;
        bb = (uint64_t)(word() >> sk); bn = 32u - sk;
    }
    __device__ __forceinline__ void refill() { if (bn <= 32u) { bb |= (uint64_t)word() << bn; bn += 32u; } }   // -> more than 32 bits
    __device__ __forceinline__ uint32_t bits(uint32_t n) { const uint32_t v = (uint32_t)bb & ((1u << n) - 1u); bb >>= n; bn -= n; return v; }   // n <= 16
    __device__ __forceinline__ uint64_t bitpos() const { return (uint64_t)w * 32u - bn; }    // bits consumed, counted from base
    __device__ __forceinline__ bool overrun() const { return bitpos() > (uint64_t)avail * 8u; }
};

// ---- the text: ring -> HBM in rounds of 4 KB, CRC-32 on the way ----
struct sink {
    uint8_t* out;             // 4 KB aligned address at or below the member's first output byte; positions below count from it
    uint32_t q0, qcap;        // the member's text is [q0, qcap)
    uint32_t q;               // write head
    uint32_t flushed;         // [.., flushed) has left the ring (a multiple of ROUND)
    uint32_t crc;             // CRC-32 of [q0, flushed)
    uint32_t k64;             // lane constant: g_crc.k64[lane]
    uint32_t lane;
    lds_t* L;

    // one round [a, a + ROUND) cut to the member's range and to `limit` (FULL: every lane holds 64 bytes of the member)
    template <bool FULL>
    __device__ __forceinline__ void round(uint32_t a, uint32_t limit) {
        const uint32_t seg = a + 64u * lane;
        const uint4* rp = reinterpret_cast<const uint4*>(L->ring + (seg & RMASK));
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = rp[j];
        const uint32_t lo = FULL ? seg : umax(seg, q0), hi = FULL ? seg + 64u : umin(seg + 64u, limit);
        if (FULL) {
            uint4* gp = reinterpret_cast<uint4*>(out + seg);
#pragma unroll
            for (int j = 0; j < 4; j++) gp[j] = v[j];
        } else {
            if (hi > lo && hi - lo == 64u) {
                uint4* gp = reinterpret_cast<uint4*>(out + seg);
#pragma unroll
                for (int j = 0; j < 4; j++) gp[j] = v[j];
            } else {
                for (uint32_t p = lo; p < hi; p++) out[p] = L->ring[p & RMASK];
            }
        }
        uint32_t c = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t wd[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint32_t byte = (wd[t >> 2] >> (8 * (t & 3))) & 0xffu;
                const uint32_t nc = L->crc_tab[(c ^ byte) & 0xffu] ^ (c >> 8);
                if (FULL) c = nc;
                else { const uint32_t p = seg + 16u * (uint32_t)j + (uint32_t)t; c = (p >= lo && p < hi) ? nc : c; }
            }
        }
        c = ~c;                                                     // CRC-32 of this lane's bytes (0 for none)
        uint32_t m = k64, xr = g_crc.x4096;
        if (!FULL && limit != a + ROUND) {                          // the last round: bytes after lane i's = 64 (ilast - i - 1) + n_last
            const uint32_t ilast = (limit - 1u - a) >> 6, n_last = limit - (a + 64u * ilast);
            const uint32_t xl = g_crc.xp8[n_last];
            const int mm = (int)ilast - (int)lane - 1;
            m = mm < 0 ? 0x80000000u : crc_mul(g_crc.k64[63 - (mm > 62 ? 62 : mm)], xl);
            xr = crc_mul(g_crc.k64[63u - ilast], xl);
        }
        const uint32_t r = rfl(wave_xor(crc_mul(m, c)));
        crc = crc_mul(xr, crc) ^ r;                                 // (the first round multiplies a zero)
    }
    __device__ __forceinline__ void flush_one() {                   // q - flushed >= ROUND
#ifdef FCZ_INFLATE_ABL_NOFLUSH
        flushed += ROUND; return;     // measurement build: nothing leaves the ring, no CRC
#endif
        if (flushed >= q0) round<true>(flushed, flushed + ROUND); else round<false>(flushed, flushed + ROUND);
        flushed += ROUND;
    }
    __device__ __forceinline__ void finish() {
        while (q - flushed >= ROUND) flush_one();
        if (q > flushed) { round<false>(flushed, q); flushed = q; }
    }
};

// one LZ77 copy of `len` bytes from `dist` bytes back to text position p (ring; beyond the ring's reach: the flushed text in HBM).
// ahead = how far beyond p this turn has already written into the ring (literals of the same window): the ring's reach is shorter by it
__device__ __forceinline__ void lz_copy(sink& sk, lds_t& L, uint32_t lane, uint32_t p, uint32_t len, uint32_t dist, uint32_t reach) {
    const uint32_t src = p - dist;
    wave_fence();
#ifdef FCZ_INFLATE_ABL_NOFAR
    reach = 0xffffffffu;              // measurement build: every match through the ring (wrong text beyond its reach; the time is the answer)
#endif
    if (__builtin_expect(dist <= reach, 1)) {
        if (__builtin_expect(len <= 64u && dist >= len, 1)) {
            uint8_t v = 0;
            if (lane < len) v = L.ring[(src + lane) & RMASK];
            wave_fence();
            if (lane < len) L.ring[(p + lane) & RMASK] = v;
        } else if (dist >= 64u) {
            for (uint32_t j = 0; j < len; j += 64u) {                 // (a later chunk reads what an earlier one wrote: in-order LDS)
                const uint32_t k = j + lane;
                uint8_t v = 0;
                if (k < len) v = L.ring[(src + k) & RMASK];
                wave_fence();
                if (k < len) L.ring[(p + k) & RMASK] = v;
                wave_fence();
            }
        } else {
            // the match overlaps its own output inside one chunk: every byte is one of the `dist` bytes before it
            for (uint32_t j = 0; j < len; j += 64u) {
                const uint32_t k = j + lane;
                uint8_t v = 0;
                if (k < len) v = L.ring[(src + (dist == 1u ? 0u : k % dist)) & RMASK];
                wave_fence();
                if (k < len) L.ring[(p + k) & RMASK] = v;
                wave_fence();
            }
        }
    } else {
        // farther back than the ring reaches: that text has left for HBM, in whole rounds this wavefront stored itself
        // (vmcnt(0): the stores are acknowledged; no line of it was read before it was complete)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t j = 0; j < len; j += 64u) {
            const uint32_t k = j + lane;
            if (k < len) L.ring[(p + k) & RMASK] = sk.out[src + k];
        }
        asm volatile("" ::: "memory");
    }
    wave_fence();
}

// The symbols of one Huffman-coded block. The stream is serial -- a symbol starts where the one before it ends -- but what a symbol
// WOULD be if it started at a given bit does not depend on the others. So every turn lane k decodes the symbol that would start at bit
// bp + k (two table gathers: literal/length, then distance; extra bits by v_bfe on a 64-bit view of the stream that starts at the
// lane's bit), and the real symbols are the chain 0 -> 0 + used(0) -> ... that the scalar unit follows with one v_readlane per symbol.
// Per window of about 4.4 symbols (14.5 bits each in PDB text at level 6) the scalar unit -- one per CU, the scarce issue port of the
// serial loop this replaces -- spends a fifth of the instructions, and the four vector units do the rest.
//   * text positions of the chain's symbols: a wave scan of their lengths; its literals are stored at once, its matches copied in order;
//   * a lane whose symbol the tables do not settle (a code longer than the fast table, end of block, an invalid code) ends the
//     chain: that ONE symbol is decoded by wave-uniform code below, then the next window starts behind it;
//   * a turn writes at most WIN_OUT bytes (the chain is cut in front of the symbol that would exceed it), so that the ring keeps
//     RING - WIN_OUT bytes of reach and an unflushed round.
constexpr uint32_t WIN_OUT = 1024;
__device__ __forceinline__ int32_t decode_body(const bitreader& br, sink& sk, lds_t& L, const canon& cn_lit, const canon& cn_dist,
                                               uint32_t lane, uint32_t* bp_io) {
    constexpr uint32_t LMASK = (1u << LBITS) - 1u, DMASK = (1u << DBITS) - 1u;
    static_assert(RING >= ROUND + 2u * WIN_OUT + 512u, "ring too small for a round and a window");
    uint32_t bp = *bp_io;
    // the stream around bp in LDS: blocks cb and cb + 1; block cb + 2 on its way in a register
    uint32_t cb = bp >> 11;
    L.in_ring[(cb & 3u) * 64u + lane] = br.load_block(cb);
    L.in_ring[((cb + 1u) & 3u) * 64u + lane] = br.load_block(cb + 1u);
    uint32_t pf = br.load_block(cb + 2u);
    wave_fence();
    // text read from HBM in the last turn (lane = byte pend_pos + lane, lanes of pendm): stored into the ring at the next turn
    constexpr uint32_t REACH = RING - WIN_OUT;
    uint64_t pendm = 0;
    uint32_t pend_pos = 0;
    uint8_t pend_v = 0;
    auto complete_pending = [&]() {
        if (pendm) {
            if ((pendm >> lane) & 1ull) L.ring[(pend_pos + lane) & RMASK] = pend_v;
            pendm = 0;
            wave_fence();
        }
    };
    for (;;) {
        // ---- every lane: the symbol that would start at bit bp + lane ----
        const uint32_t o = (bp & 31u) + lane, j = (bp >> 5) + (o >> 5), sh = o & 31u;
        const uint32_t d0 = L.in_ring[j & 255u], d1 = L.in_ring[(j + 1u) & 255u], d2 = L.in_ring[(j + 2u) & 255u];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);   // 64 bits from the lane's bit
        const uint32_t e = L.lit_tab[lo & LMASK];
        const uint32_t cl = e & 31u, kind = (e >> 6) & 3u, xb = (e >> 16) & 127u;
        const uint32_t len = (e >> 23) + __builtin_amdgcn_ubfe(lo, cl, xb);
        const uint32_t u1 = cl + xb;                                         // <= 20
        const uint32_t rest = __builtin_amdgcn_alignbit(hi, lo, u1);
        const uint32_t de = L.dist_tab[rest & DMASK];
        const uint32_t dl = de & 31u, dxb = (de >> 16) & 127u;
        const uint32_t dist = (((de >> 8) & 3u) << dxb) + 1u + __builtin_amdgcn_ubfe(rest, dl, dxb);
        const bool is_len = kind == 1u;
        const bool special = cl == 0u || kind >= 2u || (is_len && (dl == 0u || (de & 0x40u) != 0u));
        const uint32_t nxt = special ? 128u + lane : lane + (is_len ? u1 + dl + dxb : cl);
        // ---- the chain of real symbols ----
        uint64_t on = 0;
        uint32_t c = 0;
        do { on |= 1ull << c; c = rdl(nxt, c); } while (c < 64u);
        uint32_t sp = 64;                                                    // the lane of a symbol left to the uniform code, if any
        if (c >= 128u) { sp = c - 128u; on &= ~(1ull << sp); c = sp; }
        // ---- where their text goes ----
        const bool onl = ((on >> lane) & 1ull) != 0ull;
        uint32_t tot;
        const uint32_t outn = is_len ? len : 1u;
        const uint32_t ex = wave_excl_scan_dpp(onl ? outn : 0u, &tot);
        if (__builtin_expect(tot > WIN_OUT, 0)) {
            const uint64_t over = __ballot(onl && ex != 0u && ex + outn > WIN_OUT);      // (never the first symbol: a turn makes progress)
            const uint32_t t = (uint32_t)__builtin_ctzll(over);
            on &= (1ull << t) - 1ull; c = t; sp = 64; tot = rdl(ex, t);
        }
        const bool mine = ((on >> lane) & 1ull) != 0ull;
        complete_pending();                                                  // (the ring was not looked at above)
        // ---- its text. The usual window writes at most 64 bytes and its matches copy text from in front of the window: then lane =
        //      byte. The owner of a byte (the symbol it belongs to) is the newest start marker at or before it (a max-scan over
        //      markers the chain's lanes drop into LDS), its length / distance / literal come over the LDS crossbar, and ONE gather
        //      and ONE store move every literal and every match of the window. Bytes that lie beyond the ring's reach are read from
        //      the text in HBM and stored a turn later, behind that turn's table work. ----
        const uint32_t ld = len | (dist << 9);
        const uint32_t pos = sk.q + ex;
        const bool okm = !is_len || (dist >= ex + len && dist <= pos - sk.q0);
#if defined(FCZ_INFLATE_ABL_NOCOPY)
        if (false) {
#elif defined(FCZ_INFLATE_ABL_ALLBYTE)
        if (true) {                   // measurement build: every window takes the lane = byte form (wrong text beyond its conditions)
#else
        if (__builtin_expect(tot <= 64u && __ballot(mine && !okm) == 0ull, 1)) {
#endif
            L.own[lane] = 0;
            wave_fence();
            if (mine && ex < 64u) L.own[ex] = (uint8_t)(lane + 1u);
            wave_fence();
            const uint32_t k = wave_incl_max_dpp(L.own[lane]) - 1u;
            const uint32_t old = (uint32_t)__shfl((int)ld, (int)k, WAVE), oe = (uint32_t)__shfl((int)e, (int)k, WAVE);
            const uint32_t od = old >> 9;
            const bool act = lane < tot, omatch = ((oe >> 6) & 3u) == 1u;
#ifdef FCZ_INFLATE_ABL_NOFAR
            const bool isfar = false;
#else
            const bool isfar = act && omatch && od > REACH;
#endif
            const uint64_t fm = __ballot(isfar);
            if (fm) {
                // (rounds this wavefront stored: acknowledged at vmcnt(0). Every lane loads -- the others the member's first byte --
                //  so that the register is written by the load alone)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pend_v = sk.out[isfar ? sk.q + lane - od : sk.q0];
                pendm = fm; pend_pos = sk.q;
            }
            uint8_t val = (uint8_t)(oe >> 8);
            if (act && omatch && !isfar) val = L.ring[(sk.q + lane - od) & RMASK];
            wave_fence();
            if (act && !isfar) L.ring[(sk.q + lane) & RMASK] = val;
            wave_fence();
        } else {
            if (mine && !is_len) L.ring[pos & RMASK] = (uint8_t)(e >> 8);
            uint64_t mm = on & __ballot(is_len);
#ifdef FCZ_INFLATE_ABL_NOCOPY
            mm = 0;                   // measurement build: no match is copied (wrong text; the time is the answer)
#endif
            while (mm) {
                const uint32_t k = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1ull;
                const uint32_t v = rdl(ld, k), p = rdl(pos, k);
                if (__builtin_expect((v >> 9) > p - sk.q0, 0)) return INF_FAIL(ST_CODE);  // "invalid distance too far back"
                lz_copy(sk, L, lane, p, v & 511u, v >> 9, REACH);
            }
        }
        sk.q += tot; bp += c;
        // ---- the symbol the tables did not settle: wave-uniform, once ----
        if (__builtin_expect(sp < 64u, 0)) {
            complete_pending();
            const uint32_t slo = rdl(lo, sp), shi = rdl(hi, sp);
            uint32_t se = rdl(e, sp), cb2;
            if ((se & 31u) == 0u) {
                const int ix = canon_decode(slo & 0x7fffu, cn_lit, lane, &cb2);
                if (ix < 0) return INF_FAIL(ST_CODE);
                se = resolve_lit(rfl(L.lit_sorted[ix]), cb2);
            }
            const uint32_t sk_kind = (se >> 6) & 3u, scl = se & 31u;
            if (sk_kind == 0u) {
                L.ring[sk.q & RMASK] = (uint8_t)(se >> 8);
                sk.q++; bp += scl;
            } else if (sk_kind == 2u) {
                *bp_io = bp + scl;
                return ST_OK;                                                 // end of block
            } else if (sk_kind == 3u) {
                return INF_FAIL(ST_CODE);                                     // "invalid literal/length code"
            } else {
                const uint32_t sxb = (se >> 16) & 127u, su1 = scl + sxb;
                const uint32_t slen = (se >> 23) + ((slo >> scl) & ((1u << sxb) - 1u));
                const uint32_t srest = (uint32_t)((((uint64_t)shi << 32) | slo) >> su1);
                uint32_t sde = rfl(L.dist_tab[srest & DMASK]);
                if ((sde & 31u) == 0u) {
                    uint32_t db;
                    const int ix = canon_decode(srest & 0x7fffu, cn_dist, lane, &db);
                    if (ix < 0) return INF_FAIL(ST_CODE);
                    sde = resolve_dist(rfl(L.dist_sorted[ix]), db);
                }
                if (sde & 0x40u) return INF_FAIL(ST_CODE);                    // "invalid distance code"
                const uint32_t sdl = sde & 31u, sdxb = (sde >> 16) & 127u;
                const uint32_t sdist = (((sde >> 8) & 3u) << sdxb) + 1u + ((srest >> sdl) & ((1u << sdxb) - 1u));
                if (sdist > sk.q - sk.q0) return INF_FAIL(ST_CODE);
                lz_copy(sk, L, lane, sk.q, slen, sdist, REACH);
                sk.q += slen; bp += su1 + sdl + sdxb;
            }
        }
        // ---- a round of text leaves; the next block of the stream comes in ----
        if (sk.q - sk.flushed >= ROUND) {
            if (sk.q > sk.qcap) return INF_FAIL(ST_SIZE);
            complete_pending();
            wave_fence(); sk.flush_one();
        }
        while ((bp >> 11) > cb) {
            cb++;
            L.in_ring[((cb + 1u) & 3u) * 64u + lane] = pf;
            pf = br.load_block(cb + 2u);
            wave_fence();
        }
    }
}

// the DEFLATE stream of one member -> ring -> text. Returns ST_OK or why the member is left to zlib.
__device__ __forceinline__ int32_t inflate_stream(bitreader& br, sink& sk, lds_t& L, uint32_t lane) {
    canon cl{}, cn_lit{}, cn_dist{};
    for (;;) {                                                      // blocks (RFC 1951 section 3.2.3)
        if (br.overrun()) return INF_FAIL(ST_INPUT);
        br.refill();
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (type == 3u) return INF_FAIL(ST_BLOCK);
        if (type == 0u) {
            // stored: skip to the byte boundary, LEN, NLEN, LEN bytes (zlib inflate.c STORED)
            br.bits(br.bn & 7u);
            br.refill();
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if ((len ^ 0xffffu) != nlen) return INF_FAIL(ST_BLOCK);
            const uint32_t pos = (uint32_t)(br.bitpos() >> 3);      // (the bit position is a multiple of 8 here)
            if ((uint64_t)pos + len > br.avail) return INF_FAIL(ST_INPUT);
            if (len > sk.qcap - sk.q) return INF_FAIL(ST_SIZE);
            const uint32_t qs = sk.q;
            for (uint32_t j = 0; j < len; j += 64u) {
                const uint32_t k = j + lane;
                if (k < len) L.ring[(qs + k) & RMASK] = br.base[pos + k];
                wave_fence();
                sk.q = qs + umin(len, j + 64u);
                if (sk.q - sk.flushed >= ROUND) sk.flush_one();
            }
            br.seek(pos + len);
        } else {
            uint32_t nlen = 288, ndist = 30;
            if (type == 1u) {
                // fixed code (section 3.2.6): the same builder on the fixed lengths
                for (uint32_t s = lane; s < 320u; s += 64u) L.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : s < 288u ? 8 : 5;
                ndist = 32;
                wave_fence();
            } else {
                // dynamic code (section 3.2.7; zlib inflate.c TABLE .. CODELENS)
                br.refill();
                nlen = br.bits(5) + 257u; ndist = br.bits(5) + 1u;
                const uint32_t ncode = br.bits(4) + 4u;
                if (nlen > 286u || ndist > 30u) return INF_FAIL(ST_BLOCK);
                constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint32_t mine = 0;                                  // lane s: length of code-length symbol s
#pragma unroll
                for (int i = 0; i < 19; i++) {
                    if ((uint32_t)i < ncode) {
                        br.refill();
                        const uint32_t v = br.bits(3);
                        mine = (lane == order[i]) ? v : mine;
                    }
                }
                if (lane < 32u) L.lens[lane] = (uint8_t)(lane < 19u ? mine : 0u);
                wave_fence();
                uint32_t nz, mx;
                if (build_canon<1>(L.lens, 19u, L.cl_sorted, cl, lane, &nz, &mx) != 0) return INF_FAIL(ST_BLOCK);   // (zlib: "invalid code lengths set"; none at all ends in "missing end-of-block")
                uint32_t have = 0, prev = 0;
                const uint32_t total = nlen + ndist;
                while (have < total) {
                    br.refill();
                    uint32_t cb;
                    const int ix = canon_decode((uint32_t)br.bb & 0x7fffu, cl, lane, &cb);
                    if (ix < 0) return INF_FAIL(ST_BLOCK);
                    const uint32_t s = rfl(L.cl_sorted[ix]);
                    br.bits(cb);
                    uint32_t rep = 1, val = s;
                    if (s == 16u) { if (have == 0u) return INF_FAIL(ST_BLOCK); val = prev; rep = 3u + br.bits(2); }
                    else if (s == 17u) { val = 0; rep = 3u + br.bits(3); }
                    else if (s == 18u) { val = 0; rep = 11u + br.bits(7); }
                    if (have + rep > total) return INF_FAIL(ST_BLOCK);
                    for (uint32_t k = lane; k < rep; k += 64u) L.lens[have + k] = (uint8_t)val;
                    have += rep; prev = val;
                    if (br.overrun()) return INF_FAIL(ST_INPUT);
                }
                wave_fence();
                if (rfl(L.lens[256]) == 0u) return INF_FAIL(ST_BLOCK);                   // "invalid code -- missing end-of-block"
            }
            {
                uint32_t nz, mx;
                // literal/length set: complete, or left to zlib (which takes an incomplete one only when its longest code is one bit)
                if (build_canon<5>(L.lens, nlen, L.lit_sorted, cn_lit, lane, &nz, &mx) != 0) return INF_FAIL(ST_BLOCK);
                // distance set: complete; or no code at all / a single one-bit code (inftrees.c: incomplete && max == 1)
                const int rc = build_canon<1>(L.lens + nlen, ndist, L.dist_sorted, cn_dist, lane, &nz, &mx);
                if (rc == 2 || (rc == 1 && mx > 1u)) return INF_FAIL(ST_BLOCK);
            }
            fill_table<LBITS, false>(L.lit_tab, L.lit_sorted, cn_lit, lane);
            fill_table<DBITS, true>(L.dist_tab, L.dist_sorted, cn_dist, lane);
            // symbols, a WINDOW of 64 bit positions per turn (decode_body below)
            uint32_t bp = (uint32_t)br.bitpos();
            const int32_t err = decode_body(br, sk, L, cn_lit, cn_dist, lane, &bp);
            if (err == ST_OK) { br.seek(bp >> 3); br.bits(bp & 7u); }
            if (err != ST_OK) return err;
        }
        if (last) return ST_OK;
    }
}

// members in[in_off[f] .. in_off[f + 1]) -> text[text_off[f] .. text_off[f + 1]); kind[f] (may be null = all 1): 1 gzip member,
// 0 plain bytes (copied). status[f]: ST_OK or why the member was not inflated (its text range is blanked with spaces then).
__global__ __launch_bounds__(WAVE) void k_inflate(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off, uint32_t n,
                                                  const uint8_t* __restrict__ kind, const uint64_t* __restrict__ text_off,
                                                  uint8_t* text, int32_t* __restrict__ status) {
    __shared__ lds_t L;
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    if (f >= n) return;
    const uint64_t i0 = in_off[f], i1 = in_off[f + 1], t0 = text_off[f], t1 = text_off[f + 1];
    const uint8_t* src = in + i0;
    uint8_t* dst = text + t0;
    const uint64_t cap = t1 - t0, ilen = i1 - i0;
    if (kind && kind[f] == 0) {
        // plain bytes: 16 per lane and step (the source at any alignment)
        int32_t st = ST_OK;
        if (cap != ilen) st = INF_FAIL(ST_SIZE);
        else {
            const uint64_t head = umin64(cap, (16u - ((uintptr_t)dst & 15u)) & 15u);
            if (lane < head) dst[lane] = src[lane];
            const uint64_t body = (cap - head) & ~(uint64_t)15;
            for (uint64_t o = head + 16u * lane; o < head + body; o += 16u * WAVE) {
                uint4 v; __builtin_memcpy(&v, src + o, 16);
                *reinterpret_cast<uint4*>(dst + o) = v;
            }
            const uint64_t tail = head + body;
            if (tail + lane < cap) dst[tail + lane] = src[tail + lane];
        }
        if (st != ST_OK) for (uint64_t o = lane; o < cap; o += WAVE) dst[o] = ' ';
        if (lane == 0) status[f] = st;
        return;
    }
    int32_t st = ST_OK;
    do {
        if (ilen < 18u || ilen >= (1ull << 28) || cap >= (1ull << 31)) { st = INF_FAIL(ST_HEADER); break; }
        for (uint32_t i = lane; i < 256u; i += WAVE) L.crc_tab[i] = g_crc.tab[i];
        bitreader br;
        br.lane = lane;
        const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
        br.base = src - mis; br.avail = (uint32_t)ilen + mis;
        br.cur = br.load_block(0); br.nxt = 0; br.w = 0; br.bb = 0; br.bn = 0;
        // ---- member header (RFC 1952 section 2.3; zlib inflate.c HEAD .. HCRC) from the first 256 bytes ----
        auto hb = [&](uint32_t o) { const uint32_t p = mis + o; return (rdl(br.cur, (p >> 2) & 63u) >> (8u * (p & 3u))) & 0xffu; };
        const uint32_t hmax = umin((uint32_t)ilen, 256u - 4u);          // header bytes this parse can see
        if (hb(0) != 0x1fu || hb(1) != 0x8bu || hb(2) != 8u) { st = INF_FAIL(ST_HEADER); break; }
        const uint32_t flg = hb(3);
        if (flg & 0xe0u) { st = INF_FAIL(ST_HEADER); break; }                    // "unknown header flags set"
        if (flg & 0x02u) { st = INF_FAIL(ST_HEADER); break; }                    // FHCRC: the header CRC is zlib's to check
        uint32_t pos = 10;
        bool bad = false;
        if (flg & 0x04u) { if (pos + 2u > hmax) bad = true; else pos += 2u + hb(pos) + (hb(pos + 1u) << 8); }
        for (uint32_t bit = 0x08u; bit <= 0x10u && !bad; bit <<= 1) {            // FNAME, FCOMMENT: zero-terminated
            if (!(flg & bit)) continue;
            for (;;) { if (pos >= hmax) { bad = true; break; } if (hb(pos++) == 0u) break; }
        }
        if (bad || pos > hmax || (uint64_t)pos + 8u > ilen) { st = INF_FAIL(ST_HEADER); break; }
        sink sk;
        sk.lane = lane; sk.L = &L;
        const uint32_t omis = (uint32_t)((uintptr_t)dst & (ROUND - 1u));
        sk.out = dst - omis; sk.q0 = omis; sk.qcap = omis + (uint32_t)cap; sk.q = omis; sk.flushed = 0; sk.crc = 0;
        sk.k64 = g_crc.k64[lane];
        wave_fence();
        br.seek(mis + pos);
        st = inflate_stream(br, sk, L, lane);
        if (st != ST_OK) break;
        if (sk.q != sk.qcap) { st = INF_FAIL(ST_SIZE); break; }       // (before the last rounds leave: nothing is stored past the range)
        wave_fence();
        sk.finish();
        // ---- trailer: CRC-32, ISIZE (zlib inflate.c CHECK, LENGTH); nothing may be left after it ----
        br.bits(br.bn & 7u);
        const uint64_t end = br.bitpos() >> 3;                                    // bytes consumed from br.base
        if (end + 8u != br.avail) { st = INF_FAIL(ST_INPUT); break; }
        const uint8_t* tr = br.base + end;
        const uint32_t tb = lane < 8u ? (uint32_t)tr[lane] : 0u;
        const uint32_t crc_t = rdl(tb, 0) | (rdl(tb, 1) << 8) | (rdl(tb, 2) << 16) | (rdl(tb, 3) << 24);
        const uint32_t isz_t = rdl(tb, 4) | (rdl(tb, 5) << 8) | (rdl(tb, 6) << 16) | (rdl(tb, 7) << 24);
        if (crc_t != sk.crc || isz_t != (uint32_t)cap) { st = INF_FAIL(ST_CHECK); break; }
    } while (false);
    if (st != ST_OK) {
        // nothing of a refused member reaches the parser: its range becomes blanks (aligned 16-byte stores between byte edges)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t head = umin64(cap, (16u - ((uintptr_t)dst & 15u)) & 15u);
        if (lane < head) dst[lane] = ' ';
        const uint64_t body = (cap - head) & ~(uint64_t)15;
        const uint4 sp = {0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u};
        for (uint64_t o = head + 16u * lane; o < head + body; o += 16u * WAVE) *reinterpret_cast<uint4*>(dst + o) = sp;
        const uint64_t tail = head + body;
        if (tail + lane < cap) dst[tail + lane] = ' ';
    }
    if (lane == 0) status[f] = st;
}

// file_status of the ingest <- what the inflate refused (the FILE goes back to the caller's zlib + reader)
__global__ void k_inflate_merge_status(const int32_t* __restrict__ inflate_status, uint32_t n, int32_t host_gzip, int32_t* file_status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && inflate_status[i] != ST_OK) file_status[i] = host_gzip;
}

}  // namespace inflate
}  // namespace fcz
