// fcz_backbone_fast.h -- backbone reconstruction in plain float arithmetic (FCZ_NUMERICS_FAST), the decompress stage of
// fcz_kernels.h::k_backbone rebuilt around the parallelism the algorithm has once bit-equality with the reference's x86
// evaluation order is not asked for (reference: Foldcomp::decompress segment loop src/foldcomp.cpp:814-858,
// reconstructBackboneAtoms :167-246, reconstructBackboneReverse :248-273, Nerf::place_atom src/nerf.cpp:39-104,
// weightedAverage src/atom_coordinate.cpp:145-163).
//
// The exact kernel walks a chain atom by atom because every float operation must round where the reference rounds; the only
// parallelism left to it is across chains (lane = chain) and the forward atoms of a segment make a round trip through HBM.
// Here a NeRF step is a rigid transform of a local frame (origin = the atom just placed, axes = bond direction, in-plane
// normal, plane normal) that depends on (bond length, bond angle, torsion) alone:
//     e1' = -cos(b) e1 + cos(t) sin(b) e2 + sin(t) sin(b) e3        o' = o + L e1'
//     e3' =  cos(t) e3 - sin(t) e2                                   e2' = e3' x e1'
// (no square root, no division: the frame stays orthonormal by construction). Transforms compose, so a segment is cut into
// 8 runs of residues, one lane each: every lane builds its run from the identity frame (local coordinates), an 8-lane scan
// of the runs' total transforms (DPP row shifts, 3 steps) gives every lane its starting frame, and the local atoms are
// moved into place. The reverse pass is the same chain from the next anchor, with the bond angles and torsions of the
// forward pass (the angle "re-measured on the forward atoms" by the reference IS the angle the forward atom was placed
// with; only the first one of a segment, between the three start atoms, is measured). Forward atoms wait in LDS, the blend
// overwrites them, 12 bytes per atom leave the CU. 8 chains per wavefront, segments of a chain in sequence (the start of a
// segment is the blended end of the previous one). Segments longer than 32 residue steps (the last segment of chains beyond
// ~600 residues at -b 25) run in chunks of 32 steps with the forward atoms parked in a global scratch column instead of LDS.
//
// Results differ from the exact path by float rounding only: max |delta| ~1e-4 A over a chain (tests/test_gpu_fast_numerics.py
// holds it below 1e-3 A on every golden and on mixed-length batches, and the reference's two RMSD pins hold unchanged).
#pragma once
#include "fcz_kernels.h"

namespace fcz {

#ifndef FCZ_FB_G
#define FCZ_FB_G 8
#endif
constexpr int FB_G = FCZ_FB_G;               // lanes per chain (8 or 16: two or one chains per DPP row of 16 lanes; measured: 8 lanes,
                                             // 2 wavefronts per SIMD 3.5 ms per 262 144 chains, 16 lanes 4.3 ms, 3 wavefronts per SIMD spill)
constexpr int FB_CH = WAVE / FB_G;           // chains per wavefront
constexpr int FB_S = 32 / FB_G;              // residue steps per lane and chunk
constexpr int FB_K = FB_G * FB_S;            // residue steps per chunk
constexpr int FB_ROW = 3 * 3 * (FB_K + 1) + 1;   // floats per chain in LDS: 3 start atoms + 3 per step, odd stride

struct fframe { v3 e1, e2, e3, o; };

__device__ __forceinline__ v3 fv_add(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ v3 fv_scale(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 fv_fma(v3 a, float s, v3 c) { return v3{__builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y), __builtin_fmaf(a.z, s, c.z)}; }
__device__ __forceinline__ v3 fv_cross(v3 a, v3 b) {
    return v3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float fv_dot(v3 a, v3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ v3 fv_unit(v3 a) { return fv_scale(a, __builtin_amdgcn_rsqf(fv_dot(a, a))); }
// M v for the frame's axes as columns
__device__ __forceinline__ v3 fr_rot(const fframe& F, v3 v) { return fv_fma(F.e3, v.z, fv_fma(F.e2, v.y, fv_scale(F.e1, v.x))); }
__device__ __forceinline__ v3 fr_apply(const fframe& F, v3 p) { return fv_fma(F.e3, p.z, fv_fma(F.e2, p.y, fv_fma(F.e1, p.x, F.o))); }
// A o B: first B (inner, local), then A
__device__ __forceinline__ fframe fr_compose(const fframe& A, const fframe& B) {
    return fframe{fr_rot(A, B.e1), fr_rot(A, B.e2), fr_rot(A, B.e3), fr_apply(A, B.o)};
}
__device__ __forceinline__ fframe fr_identity() { return fframe{v3{1.f, 0.f, 0.f}, v3{0.f, 1.f, 0.f}, v3{0.f, 0.f, 1.f}, v3{0.f, 0.f, 0.f}}; }
// the frame place_atom(a, b, c, ...) builds: bond direction b->c, plane normal of (a, b, c), their cross product; origin c
__device__ __forceinline__ fframe fr_from_atoms(v3 a, v3 b, v3 c) {
    fframe F;
    F.e1 = fv_unit(vsub(c, b));
    F.e3 = fv_unit(fv_cross(vsub(b, a), F.e1));
    F.e2 = fv_cross(F.e3, F.e1);
    F.o = c;
    return F;
}
// one NeRF step; returns the new atom (= the new origin)
__device__ __forceinline__ v3 fr_step(fframe& F, float L, float cb, float sb, float ct, float st) {
    const v3 d = v3{-cb, ct * sb, st * sb};
    const v3 e1 = fr_rot(F, d);
    const v3 e3 = fv_fma(F.e3, ct, fv_scale(F.e2, -st));
    F.e1 = e1; F.e3 = e3; F.e2 = fv_cross(e3, e1);
    F.o = fv_fma(e1, L, F.o);
    return F.o;
}

// sine and cosine of an angle given in degrees as Nerf::place_atom takes them (src/nerf.cpp:63-71): the reference converts to
// radians, ROUNDS TO FLOAT, then calls sinf / cosf. That rounding moves the angle by up to 2.4e-7 rad -- four times the error of
// a good float sine -- and it is the same for every occurrence of an angle value, so it is reproduced here (one double
// multiplication, as the exact path's deg2rad does without its tie check); the quadrant reduction runs in double (three
// instructions), the two kernels in float with the reduced argument carried as hi + lo. Result: within 0.8 ulp of the true
// sine / cosine of the rounded radian value, i.e. within one float rounding of glibc's.
__device__ __forceinline__ void sincos_deg_fast(float deg, float* sn, float* cs) {
    const float radf = (float)((double)deg * 0.017453292519943295);
    const double xd = (double)radf;
    const double kd = __builtin_rint(xd * 0.63661977236758134);
    const double rd = __builtin_fma(-kd, 1.5707963267948966, xd);
    const float t = (float)rd, tl = (float)(rd - (double)t);
    const float t2 = t * t;
    float S = __builtin_fmaf(t2, -1.9515295891e-4f, 8.3321608736e-3f);
    S = __builtin_fmaf(S, t2, -1.6666654611e-1f);
    const float u = S * t2;
    float C = __builtin_fmaf(t2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    C = __builtin_fmaf(C, t2, 4.166664568298827e-2f);
    const float y = (t2 * t2) * C;
    const float hz = 0.5f * t2, a = 1.0f - hz, corr = (1.0f - a) - hz;      // 1 - t2/2 with its rounding error
    const float s0 = __builtin_fmaf(t, u, t), c0 = a + (corr + y);
    const float s = t + __builtin_fmaf(t, u, tl * c0);                      // sin(t + tl)
    const float c = a + ((corr + y) - tl * s0);                             // cos(t + tl)
    const int q = (int)kd;
    const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = ((q & 2) != 0) ? -s1 : s1;
    *cs = (((q + 1) & 2) != 0) ? -c1 : c1;
}

// DPP row shifts (rows of 16 lanes hold two chains of 8): lanes that would read across their chain's border keep `v`
template <int N> __device__ __forceinline__ float dpp_shr(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ float dpp_shl(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x100 + N, 0xf, 0xf, false));
}
template <int N, bool UP> __device__ __forceinline__ v3 dpp_v3(v3 v) {
    return UP ? v3{dpp_shr<N>(v.x), dpp_shr<N>(v.y), dpp_shr<N>(v.z)} : v3{dpp_shl<N>(v.x), dpp_shl<N>(v.y), dpp_shl<N>(v.z)};
}
template <int N, bool UP> __device__ __forceinline__ fframe dpp_frame(const fframe& F) {
    return fframe{dpp_v3<N, UP>(F.e1), dpp_v3<N, UP>(F.e2), dpp_v3<N, UP>(F.e3), dpp_v3<N, UP>(F.o)};
}
// Inclusive scan of the lanes' transforms over the FB_G lanes of a chain. UP: lane j ends with T_0 o T_1 o ... o T_j (the
// forward pass: lane 0 comes first); !UP: T_last o ... o T_j (the reverse pass starts at the far end).
template <bool UP> __device__ __forceinline__ fframe scan8(fframe T, int sub) {
    {
        const fframe O = dpp_frame<1, UP>(T);
        if (UP ? sub >= 1 : sub < FB_G - 1) T = fr_compose(O, T);
    }
    {
        const fframe O = dpp_frame<2, UP>(T);
        if (UP ? sub >= 2 : sub < FB_G - 2) T = fr_compose(O, T);
    }
    {
        const fframe O = dpp_frame<4, UP>(T);
        if (UP ? sub >= 4 : sub < FB_G - 4) T = fr_compose(O, T);
    }
    if (FB_G > 8) {
        const fframe O = dpp_frame<8, UP>(T);
        if (UP ? sub >= 8 : sub < FB_G - 8) T = fr_compose(O, T);
    }
    return T;
}

// trig of one residue step (one packed word): cos / sin of psi, omega, phi and of the three bond angles
struct step_trig { float c_psi, s_psi, c_om, s_om, c_phi, s_phi, c_can, s_can, c_cna, s_cna, c_nca, s_nca; float l_nca; };

__device__ __forceinline__ step_trig trig_of_word(uint64_t raw, const bb_params& P) {
    const bb_word w = decode_word(raw, P);
    step_trig t;
    sincos_deg_fast(w.psi, &t.s_psi, &t.c_psi);
    sincos_deg_fast(w.omega, &t.s_om, &t.c_om);
    sincos_deg_fast(w.phi, &t.s_phi, &t.c_phi);
    sincos_deg_fast(w.can, &t.s_can, &t.c_can);
    sincos_deg_fast(w.cna, &t.s_cna, &t.c_cna);
    sincos_deg_fast(w.nca, &t.s_nca, &t.c_nca);
    t.l_nca = (w.res != FCZ_RES_PRO) ? 1.4581f : 1.353f;     // N-CA length chosen by the PRECEDING residue, src/foldcomp.cpp:204-212
    return t;
}

#ifndef FCZ_BACKBONE_FAST_MIN_WAVES
#define FCZ_BACKBONE_FAST_MIN_WAVES 2
#endif

// One wavefront per 8 entries (perm order: grouped by length). scratch: per (block, chain) a column of `scratch_atoms` atoms for
// the forward atoms of segments that do not fit the LDS row (may be null when no segment of the batch is longer than FB_K steps).
__global__ __launch_bounds__(WAVE, FCZ_BACKBONE_FAST_MIN_WAVES) void k_backbone_fast(
        const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries, uint32_t n_slots,
        const uint32_t* __restrict__ res_off, const uint32_t* __restrict__ perm, v3* __restrict__ scratch, uint32_t scratch_atoms,
        v3* __restrict__ bb) {
    __shared__ float S_f[FB_CH][FB_ROW];
    const int lane = threadIdx.x, sub = lane & (FB_G - 1), ch = lane / FB_G;
    const uint32_t slot = blockIdx.x * FB_CH + (uint32_t)ch;
    const uint32_t c = slot < n_slots ? perm[slot] : n_entries;
    const bool valid = c < n_entries && res_off[c + 1] != res_off[c];
    const uint8_t* e = blob + (valid ? off[c] : off[0]);
    entry_view v; v.n = 0; v.n_anchor = 1; v.e = e; v.L = make_layout(0, 0, 0, 0);
    bb_params P{};
    if (valid) { v = view_entry(e); P = load_params(e); }
    const uint8_t* words = e + v.L.o_words;
    const uint32_t nseg = valid ? v.n_anchor - 1 : 0;
    uint32_t maxseg = nseg;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(maxseg, d, WAVE); maxseg = o > maxseg ? o : maxseg; }
    v3* Bc = bb + 3 * (size_t)(valid ? res_off[c] : 0);
    v3* Sg = scratch ? scratch + ((size_t)blockIdx.x * FB_CH + (size_t)ch) * scratch_atoms : nullptr;
    float* Fl = &S_f[ch][0];
    v3 p0{0.f, 0.f, 0.f}, p1 = p0, p2 = p0;
    int first = 0, next = 0;
    // Everything a segment reads from the record is fetched one segment ahead (the next anchor's atoms, the anchor index after
    // it, this lane's packed words of the segment's first chunk): unconditional loads from clamped positions, issued before the
    // arithmetic of the current segment, so that no segment starts with a memory round trip.
    const uint32_t n_w = v.n ? v.n - 1 : 0;                  // last word index
    v3 nA0 = p0, nA1 = p0, nA2 = p0;
    int nnext2 = 0;
    uint64_t nw[FB_S];
#pragma unroll
    for (int q = 0; q < FB_S; q++) nw[q] = 0;
    auto fetch_segment = [&](uint32_t sg, int fst, int nxt) {          // segment sg = residues fst .. nxt
        const uint32_t a1 = sg + 1 <= nseg ? sg + 1 : nseg, a2 = sg + 2 <= nseg ? sg + 2 : nseg;
        const uint8_t* anc = e + v.L.o_anchor + 36 * (size_t)a1;
        nA0 = ld_v3(anc); nA1 = ld_v3(anc + 12); nA2 = ld_v3(anc + 24);
        nnext2 = (int)ld_u32(e + v.L.o_aidx + 4 * (size_t)a2);
        const int Kq = nxt - fst, kc = Kq < FB_K ? (Kq > 0 ? Kq : 0) : FB_K, per = (kc + FB_G - 1) / FB_G;
#pragma unroll
        for (int q = 0; q < FB_S; q++) {
            uint32_t wi = (uint32_t)(fst + sub * per + q);
            wi = wi < n_w ? wi : n_w;
            nw[q] = ld_u64(words + 8 * (size_t)wi);
        }
    };
    if (valid) {
        p0 = ld_v3(e + v.L.o_anchor); p1 = ld_v3(e + v.L.o_anchor + 12); p2 = ld_v3(e + v.L.o_anchor + 24);
        first = (int)ld_u32(e + v.L.o_aidx); next = (int)ld_u32(e + v.L.o_aidx + 4);
        fetch_segment(0, first, next);
    }
    for (uint32_t s = 0; s < maxseg; s++) {
        const bool act = s < nseg;
        const int K = act ? next - first : 0;                 // residue steps of the segment (len - 1)
        const int T = 3 * (K + 1);
        const bool in_lds = K <= FB_K;
        const int next2 = act ? nnext2 : next;
        const v3 A0 = nA0, A1 = nA1, A2 = nA2;
        uint64_t wraw[FB_S];
#pragma unroll
        for (int q = 0; q < FB_S; q++) wraw[q] = nw[q];
        if (act && s + 1 < nseg) fetch_segment(s + 1, next, next2);
        // forward atom store: LDS row of the chain (segment fits a chunk) or the chain's scratch column
        auto f_put = [&](int f, v3 p) {
            if (in_lds) { Fl[3 * f] = p.x; Fl[3 * f + 1] = p.y; Fl[3 * f + 2] = p.z; } else Sg[f] = p;
        };
        auto f_get = [&](int f) -> v3 { return in_lds ? v3{Fl[3 * f], Fl[3 * f + 1], Fl[3 * f + 2]} : Sg[f]; };
        int kmax = K;
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) { const int o = __shfl_xor(kmax, d, WAVE); kmax = o > kmax ? o : kmax; }
        const int nchunk = (kmax + FB_K - 1) / FB_K;
        if (act && sub == 0) { f_put(0, p0); f_put(1, p1); f_put(2, p2); }
        // ---- forward pass, chunk by chunk ----
        fframe G = fr_from_atoms(p0, p1, p2);                 // running global frame at the start of the chunk
        step_trig tr[FB_S];                                   // trig of this lane's steps: reused by the reverse pass of a one-chunk segment
#pragma unroll
        for (int q = 0; q < FB_S; q++) tr[q] = step_trig{};
        for (int cq = 0; cq < nchunk; cq++) {
            const int k0 = cq * FB_K;
            const int kc = K - k0 < FB_K ? (K - k0 > 0 ? K - k0 : 0) : FB_K;     // steps of this chain in the chunk
            const int per = (kc + FB_G - 1) / FB_G;                              // steps per lane (balanced)
            const int i0 = k0 + sub * per, i1 = i0 + per < k0 + kc ? i0 + per : k0 + kc;   // this lane's steps [i0, i1)
            fframe Tl = fr_identity();
            v3 loc[FB_S][3];
#pragma unroll
            for (int q = 0; q < FB_S; q++) {
                loc[q][0] = loc[q][1] = loc[q][2] = v3{0.f, 0.f, 0.f};
                const int i = i0 + q;
                if (i < i1) {
                    const step_trig t = trig_of_word(cq == 0 ? wraw[q] : ld_u64(words + 8 * (size_t)(first + i)), P);
                    tr[q] = t;
                    loc[q][0] = fr_step(Tl, 1.3311f, t.c_can, t.s_can, t.c_psi, t.s_psi);
                    loc[q][1] = fr_step(Tl, t.l_nca, t.c_cna, t.s_cna, t.c_om, t.s_om);
                    loc[q][2] = fr_step(Tl, 1.5281f, t.c_nca, t.s_nca, t.c_phi, t.s_phi);
                }
            }
            const fframe Pj = scan8<true>(Tl, sub);           // T_0 o ... o T_j
            fframe Ex = dpp_frame<1, true>(Pj);               // T_0 o ... o T_{j-1}
            if (sub == 0) Ex = fr_identity();
            const fframe Gj = fr_compose(G, Ex);              // this lane's starting frame
#pragma unroll
            for (int q = 0; q < FB_S; q++) {
                const int i = i0 + q;
                if (i < i1) {
                    f_put(3 * i + 3, fr_apply(Gj, loc[q][0])); f_put(3 * i + 4, fr_apply(Gj, loc[q][1])); f_put(3 * i + 5, fr_apply(Gj, loc[q][2]));
                }
            }
            // frame at the end of the chunk = G o (T_0 o ... o T_7): lane 7 of the chain holds the product
            const fframe Gend = fr_compose(G, Pj);
            const int src = (lane & ~(FB_G - 1)) + FB_G - 1;
            G.e1 = v3{__shfl(Gend.e1.x, src, WAVE), __shfl(Gend.e1.y, src, WAVE), __shfl(Gend.e1.z, src, WAVE)};
            G.e2 = v3{__shfl(Gend.e2.x, src, WAVE), __shfl(Gend.e2.y, src, WAVE), __shfl(Gend.e2.z, src, WAVE)};
            G.e3 = v3{__shfl(Gend.e3.x, src, WAVE), __shfl(Gend.e3.y, src, WAVE), __shfl(Gend.e3.z, src, WAVE)};
            G.o = v3{__shfl(Gend.o.x, src, WAVE), __shfl(Gend.o.y, src, WAVE), __shfl(Gend.o.z, src, WAVE)};
        }
        __threadfence_block();
        __syncthreads();
        // ---- carry (src/foldcomp.cpp:855-857): blended last three atoms; also what the last segment keeps (:847-851) ----
        v3 c0 = p0, c1 = p1, c2 = p2;
        float invT = 0.f;
        v3 m0{0.f, 0.f, 0.f}, m1 = m0, m2 = m0;
        if (act) {
            invT = 1.0f / (float)T;
            const v3 q0 = f_get(T - 3), q1 = f_get(T - 2), q2 = f_get(T - 1);
            c0 = fv_scale(fv_fma(A0, (float)(T - 3), fv_scale(q0, 3.0f)), invT);
            c1 = fv_scale(fv_fma(A1, (float)(T - 2), fv_scale(q1, 2.0f)), invT);
            c2 = fv_scale(fv_fma(A2, (float)(T - 1), fv_scale(q2, 1.0f)), invT);
            m0 = f_get(0); m1 = f_get(1); m2 = f_get(2);      // the one bond angle that is measured: between the start atoms
        }
        // cos / sin of the angle at F[1] between F[0] and F[2] (getBondAngles on the forward atoms, src/nerf.cpp:495-508)
        float c_first, s_first;
        {
            const v3 u = vsub(m0, m1), w = vsub(m2, m1);
            const float cc = fv_dot(u, w) * __builtin_amdgcn_rsqf(fv_dot(u, u) * fv_dot(w, w));
            c_first = cc; s_first = __builtin_amdgcn_sqrtf(__builtin_fmaxf(__builtin_fmaf(-cc, cc, 1.0f), 0.0f));
        }
        // ---- reverse pass + blend, chunk by chunk from the far end (Nerf::reconstructWithReversed src/nerf.cpp:342-379) ----
        fframe H = fr_from_atoms(A2, A1, A0);                 // running frame: origin = R[T-3] = the anchor's N
        for (int cq = nchunk - 1; cq >= 0; cq--) {
            const int k0 = cq * FB_K;
            const int kc = K - k0 < FB_K ? (K - k0 > 0 ? K - k0 : 0) : FB_K;
            const int per = (kc + FB_G - 1) / FB_G;
            const int i0 = k0 + sub * per, i1 = i0 + per < k0 + kc ? i0 + per : k0 + kc;
            fframe Tl = fr_identity();
            v3 loc[FB_S][3];
            // The N-CA-C angle a residue's N is placed with belongs to the previous word. One-chunk segments (every wavefront of
            // a -b 25 batch of 350-residue chains) reuse the forward pass's trig from registers: the previous word is the previous
            // slot, or the previous lane's last slot (all lanes before an active one hold `per` steps). Chunked segments decode again.
            const bool reuse = nchunk == 1;
            float pc_nca = c_first, ps_nca = s_first;
            if (reuse) {
                float lc = tr[0].c_nca, ls = tr[0].s_nca;
#pragma unroll
                for (int q = 1; q < FB_S; q++) if (q == per - 1) { lc = tr[q].c_nca; ls = tr[q].s_nca; }
                const float uc = dpp_shr<1>(lc), us = dpp_shr<1>(ls);
                if (sub > 0) { pc_nca = uc; ps_nca = us; }
            }
#pragma unroll
            for (int q = FB_S - 1; q >= 0; q--) {
                loc[q][0] = loc[q][1] = loc[q][2] = v3{0.f, 0.f, 0.f};
                const int i = i0 + q;
                if (i < i1) {
                    step_trig t;
                    float c_n, s_n;
                    if (reuse) {
                        t = tr[q];
                        c_n = q > 0 ? tr[q > 0 ? q - 1 : 0].c_nca : pc_nca; s_n = q > 0 ? tr[q > 0 ? q - 1 : 0].s_nca : ps_nca;
                    } else {
                        t = trig_of_word(ld_u64(words + 8 * (size_t)(first + i)), P);
                        c_n = c_first; s_n = s_first;
                        if (i > 0) {
                            const bb_word wp = decode_word(ld_u64(words + 8 * (size_t)(first + i - 1)), P);
                            sincos_deg_fast(wp.nca, &s_n, &c_n);
                        }
                    }
                    loc[q][2] = fr_step(Tl, 1.3311f, t.c_cna, t.s_cna, t.c_phi, t.s_phi);     // C of residue i
                    loc[q][1] = fr_step(Tl, 1.5281f, t.c_can, t.s_can, t.c_om, t.s_om);       // CA
                    loc[q][0] = fr_step(Tl, 1.4581f, c_n, s_n, t.c_psi, t.s_psi);             // N
                }
            }
            const fframe Pj = scan8<false>(Tl, sub);          // T_7 o ... o T_j
            fframe Ex = dpp_frame<1, false>(Pj);              // T_7 o ... o T_{j+1}
            if (sub == FB_G - 1) Ex = fr_identity();
            const fframe Hj = fr_compose(H, Ex);
#pragma unroll
            for (int q = 0; q < FB_S; q++) {
                const int i = i0 + q;
                if (i < i1) {
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const int f = 3 * i + a;
                        const v3 R = fr_apply(Hj, loc[q][a]);
                        const v3 F = f_get(f);
                        // weightedAverage: (fwd (T - f) + rev f) / T
                        Bc[3 * (size_t)first + f] = fv_scale(fv_fma(R, (float)f, fv_scale(F, (float)(T - f))), invT);
                    }
                }
            }
            const fframe Hend = fr_compose(H, Pj);
            const int src = lane & ~(FB_G - 1);               // lane 0 of the chain holds the whole product
            H.e1 = v3{__shfl(Hend.e1.x, src, WAVE), __shfl(Hend.e1.y, src, WAVE), __shfl(Hend.e1.z, src, WAVE)};
            H.e2 = v3{__shfl(Hend.e2.x, src, WAVE), __shfl(Hend.e2.y, src, WAVE), __shfl(Hend.e2.z, src, WAVE)};
            H.e3 = v3{__shfl(Hend.e3.x, src, WAVE), __shfl(Hend.e3.y, src, WAVE), __shfl(Hend.e3.z, src, WAVE)};
            H.o = v3{__shfl(Hend.o.x, src, WAVE), __shfl(Hend.o.y, src, WAVE), __shfl(Hend.o.z, src, WAVE)};
        }
        if (act && s + 1 == nseg && sub == 0) {               // only the last segment keeps its final three atoms
            Bc[3 * (size_t)first + T - 3] = c0; Bc[3 * (size_t)first + T - 2] = c1; Bc[3 * (size_t)first + T - 1] = c2;
        }
        __syncthreads();
        if (act) { p0 = c0; p1 = c1; p2 = c2; first = next; next = next2; }
    }
}

}  // namespace fcz
