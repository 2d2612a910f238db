/* fcz_oracle.c -- TEST INFRASTRUCTURE ONLY: a plain-C, CPU restatement of the reference algorithm of
 * steineggerlab/foldcomp's per-chain compress/decompress path, on the same SoA batch layout the product
 * C-ABI uses. It is the checker for the HIP path and (optionally) the timed CPU baseline; it is never
 * linked into, loaded by, or a fallback for libfcz_hip.so.
 *
 * Parity status: PINNED. tests/test_oracle_vs_golden.py (against reference-minted vectors) and the live-reference
 * cases of tests/test_ingest_vs_reference.py / tests/test_gpu_parity.py check this file byte-for-byte (FCZ) and
 * bit-for-bit (decompressed float32 coordinates) against the real reference built from its own sources
 * (oracle/_ref/libfoldcomp_ref.so, see build_ref.sh) on the reference's fixtures and seeded synthetic
 * chains, and tests/golden/ holds reference-generated vectors for the GPU box where /root/reference
 * is absent.
 *
 * Every function cites the reference file:line it follows. Numeric semantics (float vs double
 * promotion points, evaluation order) are the contract -- see SURVEY.md Appendix B.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (x86-64 baseline: SSE2 scalar math, no FMA).
 */
#include "fcz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FCZ_TABLE_QUAL static const
#include "../foldcomp_amd/csrc/aa_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { float x, y, z; } v3;

static int g_restated_trig = 0;
void fcz_oracle_use_restated_trig(int on) { g_restated_trig = on; }

/* ---------------------------------------------------------------------------------------------
 * glibc 2.35 sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c --
 * the ARM "optimized-routines" single-precision algorithm, |x| < 120 branch). glibc is a third-party
 * dependency of the reference (unqualified sinf/cosf in src/nerf.cpp:67-71), absent from
 * /root/reference; the algorithm is restated here from its published description, with the table
 * constants cross-checked against the .rodata of /lib/x86_64-linux-gnu/libm.so.6 (2.35-0ubuntu3.11).
 * Plain double arithmetic, no contraction = the __sinf_sse2/__cosf_sse2 ifunc variants; the
 * __sinf_fma variants differ in the last float bit for a tiny fraction of inputs.
 * tests/test_oracle_trig.py pins this against the host libm. Only |x| < 120 is reachable from the
 * codec (angles are degrees in [-360,360] converted to radians); larger |x| defers to libm.
 * ------------------------------------------------------------------------------------------- */
static const double SC_HPI_INV = 0x1.45F306DC9C883p+23; /* 2/pi * 2^24 */
static const double SC_HPI = 0x1.921FB54442D18p0;       /* pi/2 */
static const double SC_C0 = 0x1p0, SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                    SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
static const double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7,
                    SC_S3 = -0x1.994eb3774cf24p-13;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float f) { return (f2u(f) >> 20) & 0x7ff; }

/* sinf_poly of sincosf.h; neg_cos selects the 2nd table entry (cosine coefficients negated) */
static inline float sc_poly(double x, double x2, int n, int neg_cos) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = SC_S2 + x2 * SC_S3;
        double x7 = x3 * x2;
        double s = x + x3 * SC_S1;
        return (float)(s + x7 * s1);
    } else {
        double sg = neg_cos ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = sg * SC_C3 + x2 * (sg * SC_C4);
        double c1 = sg * SC_C0 + x2 * (sg * SC_C1);
        double x6 = x4 * x2;
        double c = c1 + x4 * (sg * SC_C2);
        return (float)(c + x6 * c2);
    }
}

static inline double sc_reduce_fast(double x, int* np) {
    double r = x * SC_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * SC_HPI;
}

float fcz_oracle_sinf(float y) {
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { /* |y| < pi/4 */
        double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sc_poly(x, s, 0, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = sc_reduce_fast(x, &n);
        double s = sign[n & 3];
        return sc_poly(x * s, x * x, n, (n & 2) != 0);
    }
    return sinf(y);
}

float fcz_oracle_cosf(float y) {
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sc_poly(x, x2, 1, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        int n;
        x = sc_reduce_fast(x, &n);
        double s = sign[n & 3];
        return sc_poly(x * s, x * x, n ^ 1, (n & 2) != 0);
    }
    return cosf(y);
}

static inline float o_sinf(float x) { return g_restated_trig ? fcz_oracle_sinf(x) : sinf(x); }
static inline float o_cosf(float x) { return g_restated_trig ? fcz_oracle_cosf(x) : cosf(x); }

/* ---------------------------------------------------------------------------------------------
 * vector helpers -- src/float3d.h
 * ------------------------------------------------------------------------------------------- */
static inline v3 v_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }

/* crossProduct, src/float3d.h:19-24 */
static inline v3 v_cross(v3 a, v3 b) {
    v3 r;
    r.x = a.y * b.z - b.y * a.z;
    r.y = a.z * b.x - b.z * a.x;
    r.z = a.x * b.y - b.x * a.y;
    return r;
}

/* norm, src/float3d.h:32-34: pow/sqrt bind to the double C functions; squares are exact in double */
static inline float v_norm(v3 v) {
    double s = (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z;
    return (float)sqrt(s);
}

/* getCosineTheta, src/float3d.h:36-43: float dot products, double sqrt and divide, float result */
static inline float v_cos_theta(v3 a, v3 b) {
    float ip = (a.x * b.x) + (a.y * b.y) + (a.z * b.z);
    float s1 = a.x * a.x + a.y * a.y + a.z * a.z;
    float s2 = b.x * b.x + b.y * b.y + b.z * b.z;
    float p = s1 * s2;
    return (float)((double)ip / sqrt((double)p));
}

/* one window of getTorsionFromXYZ, src/torsion_angle.cpp:50-94 */
static float dihedral(v3 a, v3 b, v3 c, v3 d) {
    v3 d1 = v_sub(b, a), d2 = v_sub(c, b), d3 = v_sub(d, c);
    v3 u1 = v_cross(d1, d2), u2 = v_cross(d2, d3);
    float ct = v_cos_theta(u1, u2);
    double A = acos((double)ct);
    float t;
    if (isnan(A)) {
        t = (ct < 0) ? 180.0f : 0.0f;
    } else {
        t = (float)(A * 180.0 / M_PI);
    }
    v3 w = v_cross(u2, d2);
    if ((u1.x * w.x) + (u1.y * w.y) + (u1.z * w.z) < 0) t = -1 * t;
    return t;
}

/* angle, src/float3d.h:55-65 (no NaN guard) */
static float bond_angle(v3 a, v3 b, v3 c) {
    v3 d1 = v_sub(a, b), d2 = v_sub(c, b);
    float ct = v_cos_theta(d1, d2);
    return (float)(acos((double)ct) * 180.0 / M_PI);
}

/* Nerf::place_atom, src/nerf.cpp:39-104 */
static v3 place_atom(v3 a, v3 b, v3 c, float bond_length, float bond_angle_deg, float torsion_deg) {
    v3 ab = v_sub(b, a), bc = v_sub(c, b);
    float bc_norm = v_norm(bc);
    v3 bcn = {bc.x / bc_norm, bc.y / bc_norm, bc.z / bc_norm};
    float ba = (float)((double)bond_angle_deg * M_PI / 180.0);
    float ta = (float)((double)torsion_deg * M_PI / 180.0);
    v3 d2;
    d2.x = -1 * bond_length * o_cosf(ba);
    d2.y = bond_length * o_cosf(ta) * o_sinf(ba);
    d2.z = bond_length * o_sinf(ta) * o_sinf(ba);
    v3 n = v_cross(ab, bcn);
    float n_norm = v_norm(n);
    n.x = n.x / n_norm; n.y = n.y / n_norm; n.z = n.z / n_norm;
    v3 nbc = v_cross(n, bcn);
    v3 D = {0.0f, 0.0f, 0.0f};
    D.x += (bcn.x * d2.x); D.x += (nbc.x * d2.y); D.x += (n.x * d2.z);
    D.y += (bcn.y * d2.x); D.y += (nbc.y * d2.y); D.y += (n.y * d2.z);
    D.z += (bcn.z * d2.x); D.z += (nbc.z * d2.y); D.z += (n.z * d2.z);
    D.x += c.x; D.y += c.y; D.z += c.z;
    return D;
}

/* ---------------------------------------------------------------------------------------------
 * quantisers -- src/discretizer.cpp
 * ------------------------------------------------------------------------------------------- */
typedef struct { float min, max, disc_f, cont_f; } quant;

/* Discretizer::Discretizer(values, nb), src/discretizer.cpp:22-33 (std::min_element/max_element:
 * first occurrence wins) */
static quant quant_fit(const float* v, uint32_t n, unsigned nb) {
    quant q; memset(&q, 0, sizeof q);
    if (n == 0) return q;
    float mn = v[0], mx = v[0];
    for (uint32_t i = 1; i < n; i++) {
        if (v[i] < mn) mn = v[i];
        if (mx < v[i]) mx = v[i];
    }
    q.min = mn; q.max = mx;
    q.disc_f = (float)nb / (mx - mn);
    q.cont_f = (mx - mn) / (float)nb;
    return q;
}

/* vector discretize, src/discretizer.cpp:43-53: float product, double +0.5, truncation.
 * (unsigned)NaN is 0 with x86-64 gcc (cvttsd2si -> 0x8000000000000000 -> low 32 bits) */
static inline unsigned quant_round(const quant* q, float v) {
    double d = (double)((v - q->min) * q->disc_f) + 0.5;
    if (isnan(d)) return 0u;
    return (unsigned)(int64_t)d;
}

/* FixedAngleDiscretizer(255), src/discretizer.h:89-96; scalar discretize (truncating),
 * src/discretizer.cpp:55-57 */
static inline quant quant_fixed_angle(void) {
    quant q;
    q.min = (float)-180.0; q.max = (float)180.0;
    q.disc_f = (float)255u / (q.max - q.min);
    q.cont_f = (q.max - q.min) / (float)255u;
    return q;
}
static inline unsigned quant_trunc(const quant* q, float v) {
    float f = (v - q->min) * q->disc_f;
    if (isnan(f)) return 0u;
    return (unsigned)(int64_t)f;
}
/* continuize, src/discretizer.cpp:64,71 / _continuize src/foldcomp.cpp:155-158 */
static inline float dequant(unsigned q, float mn, float cont_f) { return ((float)q * cont_f) + mn; }

/* ---------------------------------------------------------------------------------------------
 * FCZ layout -- src/foldcomp.h:118-136, src/foldcomp.cpp:1038-1109
 * ------------------------------------------------------------------------------------------- */
static inline void put_u16(uint8_t* p, unsigned v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; }
static inline void put_u32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
static inline void put_f32(uint8_t* p, float v) { memcpy(p, &v, 4); }
static inline unsigned get_u16(const uint8_t* p) { return p[0] | (p[1] << 8); }
static inline uint32_t get_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float get_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

/* Foldcomp::_setAnchor, src/foldcomp.cpp:745-755 */
static int anchor_indices(int n_res, int threshold, int* idx /* >= n_res/threshold + 2 */) {
    int n_inner = n_res / threshold;
    int n_all = n_inner + 2;
    int interval = n_res / (n_all - 1);
    for (int i = 0; i < n_all - 1; i++) idx[i] = i * interval;
    idx[n_all - 1] = n_res - 1;
    return n_all;
}

static int chain_sc_count(uint32_t n_res, const uint8_t* res_code) {
    int s = 0;
    for (uint32_t r = 0; r < n_res; r++) s += fcz_res_natoms[res_code[r] < 24 ? res_code[r] : 23] - 3;
    return s;
}

static long fcz_size(uint32_t n_res, int n_anchor, uint32_t title_len, int n_sc) {
    /* Foldcomp::getSize, src/foldcomp.cpp:1190-1214 */
    return 4 + 72 + 4L * n_anchor + title_len + 36L * n_anchor + 1 + 12 + 8L * n_res + n_sc + 8 + n_res;
}

/* first atom of residue with a given code (findFirstAtomCoords, src/sidechain.cpp:140-147:
 * missing -> (0,0,0)) */
static v3 find_atom(uint32_t a0, uint32_t a1, const float* x, const float* y, const float* z,
                    const uint8_t* atom_code, int code) {
    v3 r = {0, 0, 0};
    for (uint32_t a = a0; a < a1; a++)
        if (atom_code[a] == code) { r.x = x[a]; r.y = y[a]; r.z = z[a]; return r; }
    return r;
}

typedef struct {
    float *phi, *psi, *omega, *n_ca_c, *ca_c_n, *c_n_ca; /* n_res-1 each */
    v3* bb;                                               /* 3*n_res backbone atoms */
} chain_angles;

/* Foldcomp::preprocess geometry part, src/foldcomp.cpp:484-505 */
static int compute_backbone(uint32_t n_res, const uint32_t* atom_off, const float* x, const float* y,
                            const float* z, const uint8_t* atom_code, chain_angles* A) {
    /* filterBackbone (src/atom_coordinate.cpp:135-143): N, CA, C atoms in input order. With the ABI
     * precondition (each residue holds N, CA, C in this order) that is the first N/CA/C of a residue. */
    for (uint32_t r = 0; r < n_res; r++)
        for (int k = 0; k < 3; k++)
            A->bb[3 * r + k] = find_atom(atom_off[r], atom_off[r + 1], x, y, z, atom_code, k);
    uint32_t nb = 3 * n_res;
    /* getTorsionFromXYZ(backbone, 1), src/torsion_angle.cpp:46-96; split src/foldcomp.cpp:488-492 */
    for (uint32_t i = 0; i + 3 < nb; i++) {
        float t = dihedral(A->bb[i], A->bb[i + 1], A->bb[i + 2], A->bb[i + 3]);
        uint32_t k = i / 3;
        if (i % 3 == 0) A->psi[k] = t; else if (i % 3 == 1) A->omega[k] = t; else A->phi[k] = t;
    }
    /* Nerf::getBondAngles, src/nerf.cpp:495-508: out[i-1] = angle at atom i, i = 1..nb-2;
     * split src/foldcomp.cpp:497-505 skips out[0] (first N-CA-C): j = index into out, j >= 1 */
    for (uint32_t j = 1; j + 2 < nb; j++) {
        float t = bond_angle(A->bb[j], A->bb[j + 1], A->bb[j + 2]);
        uint32_t k = (j - 1) / 3;
        if (j % 3 == 0) A->n_ca_c[k] = t; else if (j % 3 == 1) A->ca_c_n[k] = t; else A->c_n_ca[k] = t;
    }
    return 0;
}

/* calculateTorsionAnglesInResidue, src/sidechain.cpp:149-168 */
static int sidechain_torsions(uint32_t a0, uint32_t a1, const float* x, const float* y, const float* z,
                              const uint8_t* atom_code, int rc, float* out) {
    int na = fcz_res_natoms[rc];
    for (int j = 3; j < na; j++) {
        unsigned pk = fcz_res_prev[rc][j];
        v3 p0 = find_atom(a0, a1, x, y, z, atom_code, fcz_res_atom[rc][pk & 15]);
        v3 p1 = find_atom(a0, a1, x, y, z, atom_code, fcz_res_atom[rc][(pk >> 4) & 15]);
        v3 p2 = find_atom(a0, a1, x, y, z, atom_code, fcz_res_atom[rc][(pk >> 8) & 15]);
        v3 cu = find_atom(a0, a1, x, y, z, atom_code, fcz_res_atom[rc][j]);
        out[j - 3] = dihedral(p0, p1, p2, cu);
    }
    return na - 3;
}

static int valid_res_code(int rc) { return (rc >= 0 && rc < 20) || rc == 23; }

int fcz_oracle_angles_chain(uint32_t n_res, const uint32_t* atom_off, const float* x, const float* y,
                            const float* z, const uint8_t* atom_code, const uint8_t* res_code,
                            float* phi, float* psi, float* omega, float* n_ca_c, float* ca_c_n,
                            float* c_n_ca, float* sc) {
    if (n_res < 2) return FCZ_E_TOO_SHORT;
    chain_angles A = {phi, psi, omega, n_ca_c, ca_c_n, c_n_ca, NULL};
    A.bb = (v3*)malloc(sizeof(v3) * 3 * n_res);
    compute_backbone(n_res, atom_off, x, y, z, atom_code, &A);
    free(A.bb);
    int k = 0;
    for (uint32_t r = 0; r < n_res; r++) {
        if (!valid_res_code(res_code[r])) return FCZ_E_RESIDUE;
        k += sidechain_torsions(atom_off[r], atom_off[r + 1], x, y, z, atom_code, res_code[r], sc + k);
    }
    return k;
}

/* Foldcomp::compress + writeStream, src/foldcomp.cpp:450-606, 1038-1109 */
long fcz_oracle_compress_chain(uint32_t n_res, const uint32_t* atom_off, const float* x, const float* y,
                               const float* z, const uint8_t* atom_code, const uint8_t* res_code,
                               const float* bfac_ca, int32_t first_res_index, int32_t first_atom_index,
                               char chain_id, const char* title, uint32_t title_len,
                               int32_t anchor_threshold, uint8_t* out, long out_cap) {
    if (n_res < 2) return FCZ_E_TOO_SHORT;
    if (anchor_threshold <= 0) return FCZ_E_INVALID_ARG;
    for (uint32_t r = 0; r < n_res; r++)
        if (!valid_res_code(res_code[r])) return FCZ_E_RESIDUE;
    int n_sc = chain_sc_count(n_res, res_code);
    int n_anchor = (int)n_res / anchor_threshold + 2;
    long size = fcz_size(n_res, n_anchor, title_len, n_sc);
    if (size > out_cap) return FCZ_E_INVALID_ARG;

    uint32_t m = n_res - 1;
    float* buf = (float*)malloc(sizeof(float) * 6 * m);
    chain_angles A = {buf, buf + m, buf + 2 * m, buf + 3 * m, buf + 4 * m, buf + 5 * m, NULL};
    A.bb = (v3*)malloc(sizeof(v3) * 3 * n_res);
    compute_backbone(n_res, atom_off, x, y, z, atom_code, &A);

    /* six quantisers, src/foldcomp.cpp:508-519 (2^bits - 1 bins) */
    quant qphi = quant_fit(A.phi, m, 4095), qpsi = quant_fit(A.psi, m, 4095), qomg = quant_fit(A.omega, m, 2047);
    quant qnca = quant_fit(A.n_ca_c, m, 255), qcan = quant_fit(A.ca_c_n, m, 255), qcna = quant_fit(A.c_n_ca, m, 255);
    quant qtmp = quant_fit(bfac_ca, n_res, 255); /* src/foldcomp.cpp:549-550 */

    uint32_t a_first = atom_off[0], a_end = atom_off[n_res];
    uint32_t n_atom = a_end - a_first;
    int has_oxt = atom_code[a_end - 1] == FCZ_ATOM_OXT; /* src/foldcomp.cpp:474 */

    uint8_t* p = out;
    memcpy(p, "FCMP", 4); p += 4;
    /* CompressedFileHeader, src/foldcomp.h:118-136; get_header src/foldcomp.cpp:1340-1367.
     * The 4 struct padding bytes (file offsets 14,15,22,23) are written as zero. */
    memset(p, 0, 72);
    put_u16(p + 0, n_res & 0xffff);
    put_u16(p + 2, n_atom & 0xffff);
    put_u16(p + 4, (unsigned)first_res_index & 0xffff);
    put_u16(p + 6, (unsigned)first_atom_index & 0xffff);
    p[8] = (uint8_t)n_anchor;
    p[9] = (uint8_t)chain_id;
    put_u32(p + 12, (uint32_t)n_sc);
    p[16] = (uint8_t)fcz_res1[res_code[0]];            /* firstResidue: atoms[0].residue */
    p[17] = (uint8_t)fcz_res1[res_code[n_res - 1]];    /* lastResidue: atoms[last].residue */
    put_u32(p + 20, title_len);
    const quant* qs[6] = {&qphi, &qpsi, &qomg, &qnca, &qcan, &qcna};
    for (int i = 0; i < 6; i++) { put_f32(p + 24 + 4 * i, qs[i]->min); put_f32(p + 48 + 4 * i, qs[i]->cont_f); }
    p += 72;
    /* anchors, src/foldcomp.cpp:745-761, written :1045-1059 */
    int* aidx = (int*)malloc(sizeof(int) * (size_t)n_anchor);
    anchor_indices((int)n_res, anchor_threshold, aidx);
    for (int i = 0; i < n_anchor; i++) { put_u32(p, (uint32_t)aidx[i]); p += 4; }
    memcpy(p, title, title_len); p += title_len;
    for (int i = 0; i < n_anchor; i++)
        for (int k = 0; k < 3; k++) {
            v3 c = A.bb[3 * aidx[i] + k];
            put_f32(p, c.x); put_f32(p + 4, c.y); put_f32(p + 8, c.z); p += 12;
        }
    free(aidx);
    *p++ = (uint8_t)has_oxt;
    if (has_oxt) { put_f32(p, x[a_end - 1]); put_f32(p + 4, y[a_end - 1]); put_f32(p + 8, z[a_end - 1]); }
    else memset(p, 0, 12);
    p += 12;
    /* packed words, src/foldcomp.cpp:582-601 + convertBackboneChainToBytes :33-52 */
    for (uint32_t k = 0; k < n_res; k++) {
        unsigned res = res_code[k], om = 0, ps = 0, ph = 0, a1 = 0, a2 = 0, a3 = 0;
        if (k < m) {
            ps = quant_round(&qpsi, A.psi[k]); om = quant_round(&qomg, A.omega[k]); ph = quant_round(&qphi, A.phi[k]);
            a3 = quant_round(&qnca, A.n_ca_c[k]); a1 = quant_round(&qcan, A.ca_c_n[k]); a2 = quant_round(&qcna, A.c_n_ca[k]);
        }
        /* bit-fields truncate: omega:11, psi:12, phi:12, angles:8 (src/foldcomp.h:71-81) */
        om &= 0x7ff; ps &= 0xfff; ph &= 0xfff; a1 &= 0xff; a2 &= 0xff; a3 &= 0xff; res &= 0x1f;
        p[0] = (uint8_t)((res << 3) | (om >> 8));
        p[1] = (uint8_t)(om & 0xff);
        p[2] = (uint8_t)(ps >> 4);
        p[3] = (uint8_t)(((ps & 0xf) << 4) | (ph >> 8));
        p[4] = (uint8_t)(ph & 0xff);
        p[5] = (uint8_t)a1; p[6] = (uint8_t)a2; p[7] = (uint8_t)a3;
        p += 8;
    }
    /* side-chain torsion bytes, src/foldcomp.cpp:527-540, written :1088-1095 */
    quant qsc = quant_fixed_angle();
    for (uint32_t r = 0; r < n_res; r++) {
        float t[16];
        int nt = sidechain_torsions(atom_off[r], atom_off[r + 1], x, y, z, atom_code, res_code[r], t);
        for (int j = 0; j < nt; j++) *p++ = (uint8_t)quant_trunc(&qsc, t[j]);
    }
    /* temperature factors, src/foldcomp.cpp:1099-1106 */
    put_f32(p, qtmp.min); put_f32(p + 4, qtmp.cont_f); p += 8;
    for (uint32_t r = 0; r < n_res; r++) *p++ = (uint8_t)quant_round(&qtmp, bfac_ca[r]);
    free(buf); free(A.bb);
    return (long)(p - out);
}

/* ---------------------------------------------------------------------------------------------
 * decompress -- Foldcomp::read (src/foldcomp.cpp:904-1036) + Foldcomp::decompress (:779-902)
 * ------------------------------------------------------------------------------------------- */
static int res_code_from_one_letter(char c) {
    for (int i = 0; i < 24; i++) if (fcz_res1[i] == c) return i;
    return 23; /* getThreeLetterCode falls through to UNK */
}

int fcz_oracle_entry_info(const uint8_t* e, uint64_t len, fcz_entry_info* info) {
    memset(info, 0, sizeof *info);
    if (len < 76) { info->status = (len >= 4 && memcmp(e, "FCMP", 4) != 0) ? FCZ_E_BAD_MAGIC : FCZ_E_TRUNCATED; return info->status; }
    if (memcmp(e, "FCMP", 4) != 0) { info->status = FCZ_E_BAD_MAGIC; return info->status; }
    const uint8_t* h = e + 4;
    uint32_t n_res = get_u16(h + 0);
    info->n_residues = n_res;
    info->n_atoms_header = get_u16(h + 2);
    info->first_res_index = (int32_t)get_u16(h + 4);
    info->first_atom_index = (int32_t)get_u16(h + 6);
    info->n_anchors = h[8];
    info->chain_id = (char)h[9];
    info->n_sidechain_torsions = get_u32(h + 12);
    info->first_residue = (char)h[16];
    info->last_residue = (char)h[17];
    info->title_len = get_u32(h + 20);
    info->title_off = 76 + 4 * info->n_anchors;
    uint64_t need = 76ull + 4ull * info->n_anchors + info->title_len + 36ull * info->n_anchors + 13 +
                    8ull * n_res + info->n_sidechain_torsions + 8 + n_res;
    if (len < need) { info->status = FCZ_E_TRUNCATED; return info->status; }
    if (n_res < 2 || info->n_anchors < 2) { info->status = FCZ_E_TOO_SHORT; return info->status; }
    const uint8_t* q = e + 76 + 4 * info->n_anchors + info->title_len + 36 * info->n_anchors;
    info->has_oxt = q[0];
    const uint8_t* words = q + 13;
    uint32_t na = 0, nsc = 0;
    for (uint32_t k = 0; k < n_res; k++) {
        int rc = words[8 * k] >> 3;
        if (k == 0) rc = res_code_from_one_letter(info->first_residue); /* src/foldcomp.cpp:863 */
        if (rc >= 24) rc = 23;
        if (!valid_res_code(rc)) { info->status = FCZ_E_RESIDUE; return info->status; }
        na += fcz_res_natoms[rc];
        nsc += fcz_res_natoms[rc] - 3;
    }
    /* the reference indexes sideChainAnglesDiscretized without bounds checks; require consistency */
    if (nsc != info->n_sidechain_torsions) { info->status = FCZ_E_TRUNCATED; return info->status; }
    info->n_atoms_out = na + (info->has_oxt ? 1 : 0);
    info->status = FCZ_OK;
    return FCZ_OK;
}

int fcz_oracle_decompress_chain(const uint8_t* e, uint64_t len, int alt_order, float* ox, float* oy,
                                float* oz, float* bfac_res, uint8_t* res_code_out, uint8_t* atom_code_out) {
    fcz_entry_info info;
    int st = fcz_oracle_entry_info(e, len, &info);
    if (st != FCZ_OK) return st;
    const uint8_t* h = e + 4;
    int n = (int)info.n_residues, nA = (int)info.n_anchors;
    float mins[6], cfs[6];
    for (int i = 0; i < 6; i++) { mins[i] = get_f32(h + 24 + 4 * i); cfs[i] = get_f32(h + 48 + 4 * i); }
    const uint8_t* p = e + 76;
    int* aidx = (int*)malloc(sizeof(int) * (size_t)nA);
    for (int i = 0; i < nA; i++) aidx[i] = (int32_t)get_u32(p + 4 * i);
    p += 4 * nA + info.title_len;
    const uint8_t* anchors = p; /* nA x 9 floats: first = prevAtoms, then inner anchors, last */
    p += 36 * nA;
    p += 1; /* hasOXT */
    v3 oxt = {get_f32(p), get_f32(p + 4), get_f32(p + 8)};
    p += 12;
    const uint8_t* words = p; p += 8 * n;
    const uint8_t* scb = p; p += info.n_sidechain_torsions;
    float tmin = get_f32(p), tcf = get_f32(p + 4); p += 8;
    const uint8_t* tb = p;

    /* unpack (convertBytesToBackboneChain, src/foldcomp.cpp:60-77) + dequantise (:122-153,:784-804) */
    int* rc = (int*)malloc(sizeof(int) * (size_t)n);
    float* ang = (float*)malloc(sizeof(float) * 6 * (size_t)n);
    float *phi = ang, *psi = ang + n, *omg = ang + 2 * n, *nca = ang + 3 * n, *can = ang + 4 * n, *cna = ang + 5 * n;
    for (int k = 0; k < n; k++) {
        const uint8_t* b = words + 8 * k;
        rc[k] = b[0] >> 3; if (rc[k] >= 24) rc[k] = 23;
        unsigned om = ((b[0] & 7u) << 8) | b[1];
        unsigned ps = ((unsigned)b[2] << 4) | (b[3] >> 4);
        unsigned ph = ((b[3] & 0xfu) << 8) | b[4];
        phi[k] = dequant(ph, mins[0], cfs[0]); psi[k] = dequant(ps, mins[1], cfs[1]); omg[k] = dequant(om, mins[2], cfs[2]);
        nca[k] = dequant(b[7], mins[3], cfs[3]); can[k] = dequant(b[5], mins[4], cfs[4]); cna[k] = dequant(b[6], mins[5], cfs[5]);
    }
    /* backbone atoms of the whole chain after blending */
    v3* bb = (v3*)malloc(sizeof(v3) * 3 * (size_t)n);
    int nbb = 0;
    int maxseg = 0;
    for (int s = 0; s + 1 < nA; s++) { int l = aidx[s + 1] - aidx[s] + 1; if (l > maxseg) maxseg = l; }
    if (maxseg < 1) maxseg = 1;
    v3* F = (v3*)malloc(sizeof(v3) * 3 * (size_t)maxseg);
    v3* R = (v3*)malloc(sizeof(v3) * 3 * (size_t)maxseg);
    float* BA = (float*)malloc(sizeof(float) * 3 * (size_t)maxseg);
    v3 prev[3];
    for (int k = 0; k < 3; k++) { prev[k].x = get_f32(anchors + 12 * k); prev[k].y = get_f32(anchors + 12 * k + 4); prev[k].z = get_f32(anchors + 12 * k + 8); }
    int bad = 0;
    for (int s = 0; s + 1 < nA && !bad; s++) {
        /* segment slicing, src/foldcomp.cpp:814-844 */
        int first = aidx[s] < n - 1 ? aidx[s] : n - 1;
        int last = aidx[s + 1] + 1 < n - 1 ? aidx[s + 1] + 1 : n - 1;
        int len_s = last - first + (s == nA - 2 ? 1 : 0);
        if (first < 0 || len_s < 1 || len_s > maxseg || first + len_s > n) { bad = 1; break; }
        int T = 3 * len_s;
        /* forward NeRF: reconstructBackboneAtoms, src/foldcomp.cpp:167-246 */
        F[0] = prev[0]; F[1] = prev[1]; F[2] = prev[2];
        for (int i = 0; i < len_s - 1; i++) {
            int w = first + i;
            v3 N = place_atom(F[3 * i], F[3 * i + 1], F[3 * i + 2], (float)1.3311, can[w], psi[w]);
            float l_nca = (fcz_res1[rc[w]] != 'P') ? (float)1.4581 : (float)1.353;
            v3 CA = place_atom(F[3 * i + 1], F[3 * i + 2], N, l_nca, cna[w], omg[w]);
            v3 C = place_atom(F[3 * i + 2], N, CA, (float)1.5281, nca[w], phi[w]);
            F[3 * i + 3] = N; F[3 * i + 4] = CA; F[3 * i + 5] = C;
        }
        /* reverse pass: reconstructBackboneReverse, src/foldcomp.cpp:248-273 */
        for (int j = 1; j + 1 < T; j++) BA[j] = bond_angle(F[j - 1], F[j], F[j + 1]); /* angle at atom j */
        const uint8_t* anc = anchors + 36 * (s + 1);
        for (int k = 0; k < 3; k++) {
            R[T - 3 + k].x = get_f32(anc + 12 * k); R[T - 3 + k].y = get_f32(anc + 12 * k + 4); R[T - 3 + k].z = get_f32(anc + 12 * k + 8);
        }
        /* Nerf::reconstructWithReversed, src/nerf.cpp:342-379, in forward indexing: atom f is placed
         * from atoms f+3, f+2, f+1 with the bond angle at f+1 and the torsion of window (f..f+3).
         * torsions: sub-slice of (psi,omega,phi) triples, src/foldcomp.cpp:832-841 */
        for (int f = T - 4; f >= 0; f--) {
            float L = (f % 3 == 0) ? 1.4581f : (f % 3 == 1) ? 1.5281f : 1.3311f; /* src/nerf.h:40-41 */
            int w = first + f / 3;
            float tor = (f % 3 == 0) ? psi[w] : (f % 3 == 1) ? omg[w] : phi[w];
            R[f] = place_atom(R[f + 3], R[f + 2], R[f + 1], L, BA[f + 1], tor);
        }
        /* weightedAverage, src/atom_coordinate.cpp:145-163 */
        for (int j = 0; j < T; j++) {
            F[j].x = ((F[j].x * (float)(T - j)) + (R[j].x * (float)j)) / (float)T;
            F[j].y = ((F[j].y * (float)(T - j)) + (R[j].y * (float)j)) / (float)T;
            F[j].z = ((F[j].z * (float)(T - j)) + (R[j].z * (float)j)) / (float)T;
        }
        int keep = (s != nA - 2) ? T - 3 : T; /* src/foldcomp.cpp:847-851 */
        if (nbb + keep > 3 * n) { bad = 1; break; }
        memcpy(bb + nbb, F, sizeof(v3) * (size_t)keep);
        nbb += keep;
        prev[0] = F[T - 3]; prev[1] = F[T - 2]; prev[2] = F[T - 1];
    }
    int out_atoms = -1;
    if (!bad && nbb == 3 * n) {
        /* side chains: Nerf::reconstructAminoAcid, src/nerf.cpp:106-155; torsions dequantised with the
         * fixed-angle quantiser (src/foldcomp.cpp:338-369); B-factors :884-892; OXT :894-898 */
        quant qsc = quant_fixed_angle();
        int sci = 0, a = 0;
        for (int r = 0; r < n; r++) {
            int c = rc[r];
            if (r == 0) c = res_code_from_one_letter(info.first_residue);
            int na = fcz_res_natoms[c];
            v3 P[FCZ_MAX_RES_ATOMS];
            P[0] = bb[3 * r]; P[1] = bb[3 * r + 1]; P[2] = bb[3 * r + 2];
            for (int j = 3; j < na; j++) {
                unsigned pk = fcz_res_prev[c][j];
                float L, ANG; uint32_t lb = fcz_res_blen_bits[c][j], ab = fcz_res_bang_bits[c][j];
                memcpy(&L, &lb, 4); memcpy(&ANG, &ab, 4);
                float tor = dequant(scb[sci++], qsc.min, qsc.cont_f);
                P[j] = place_atom(P[pk & 15], P[(pk >> 4) & 15], P[(pk >> 8) & 15], L, ANG, tor);
            }
            for (int j = 0; j < na; j++) {
                int slot = alt_order ? fcz_res_alt_slot[c][j] : j;
                ox[a] = P[slot].x; oy[a] = P[slot].y; oz[a] = P[slot].z;
                if (atom_code_out) atom_code_out[a] = fcz_res_atom[c][slot];
                a++;
            }
            bfac_res[r] = dequant(tb[r], tmin, tcf);
            if (res_code_out) res_code_out[r] = (uint8_t)c;
        }
        if (info.has_oxt) {
            ox[a] = oxt.x; oy[a] = oxt.y; oz[a] = oxt.z;
            if (atom_code_out) atom_code_out[a] = FCZ_ATOM_OXT;
            a++;
        }
        out_atoms = a;
    }
    free(aidx); free(rc); free(ang); free(bb); free(F); free(R); free(BA);
    return out_atoms >= 0 ? out_atoms : FCZ_E_TRUNCATED;
}

/* Foldcomp::checkValidity, src/foldcomp.cpp:1492-1532 */
int fcz_oracle_check(const uint8_t* e, uint64_t len) {
    fcz_entry_info info;
    if (len < 76 || memcmp(e, "FCMP", 4) != 0) return FCZ_E_BAD_MAGIC;
    fcz_oracle_entry_info(e, len, &info);
    if (info.status == FCZ_E_TRUNCATED || info.status == FCZ_E_BAD_MAGIC) return info.status;
    const uint8_t* words = e + 76 + 4 * info.n_anchors + info.title_len + 36 * info.n_anchors + 13;
    const uint8_t* scb = words + 8 * info.n_residues;
    const uint8_t* tb = scb + info.n_sidechain_torsions + 8;
    int empty_bb = 1, empty_sc = 1, empty_t = 1;
    for (uint32_t k = 0; k < info.n_residues; k++) {
        const uint8_t* b = words + 8 * k;
        if ((b[0] & 7) | b[1] | b[2] | b[3] | b[4]) empty_bb = 0;
        if (tb[k]) empty_t = 0;
    }
    for (uint32_t k = 0; k < info.n_sidechain_torsions; k++) if (scb[k]) empty_sc = 0;
    if (empty_bb) return 4; /* E_EMPTY_BACKBONE_ANGLE */
    if (empty_sc) return 5;
    if (empty_t) return 6;
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * batch forms
 * ------------------------------------------------------------------------------------------- */
int fcz_oracle_compress_sizes(const fcz_chain_batch* in, uint64_t* out_off) {
    uint64_t o = 0;
    for (uint32_t c = 0; c < in->n_chains; c++) {
        out_off[c] = o;
        uint32_t r0 = in->res_off[c], n = in->res_off[c + 1] - r0;
        int n_anchor = (int)n / in->anchor_threshold + 2;
        o += (uint64_t)fcz_size(n, n_anchor, in->title_off[c + 1] - in->title_off[c], chain_sc_count(n, in->res_code + r0));
    }
    out_off[in->n_chains] = o;
    return FCZ_OK;
}

int fcz_oracle_compress_batch(const fcz_chain_batch* in, const uint64_t* out_off, uint8_t* out,
                              int32_t* status, int n_threads) {
    int rc_all = FCZ_OK;
    long C = in->n_chains;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (long c = 0; c < C; c++) {
        uint32_t r0 = in->res_off[c], n = in->res_off[c + 1] - r0;
        long sz = fcz_oracle_compress_chain(n, in->atom_off + r0, in->x, in->y, in->z, in->atom_code,
                                            in->res_code + r0, in->bfac_ca + r0, in->first_res_index[c],
                                            in->first_atom_index[c], in->chain_id[c],
                                            in->titles + in->title_off[c], in->title_off[c + 1] - in->title_off[c],
                                            in->anchor_threshold, out + out_off[c], (long)(out_off[c + 1] - out_off[c]));
        int st = sz < 0 ? (int)sz : (sz == (long)(out_off[c + 1] - out_off[c]) ? FCZ_OK : FCZ_E_INVALID_ARG);
        if (status) status[c] = st;
        if (st != FCZ_OK) rc_all = st;
    }
    return rc_all;
}

int fcz_oracle_decompress_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n, fcz_entry_info* info,
                                uint32_t* res_off, uint32_t* atom_off) {
    uint32_t r = 0, a = 0;
    for (uint32_t i = 0; i < n; i++) {
        res_off[i] = r; atom_off[i] = a;
        fcz_oracle_entry_info(blob + off[i], off[i + 1] - off[i], &info[i]);
        if (info[i].status == FCZ_OK) { r += info[i].n_residues; a += info[i].n_atoms_out; }
    }
    res_off[n] = r; atom_off[n] = a;
    return FCZ_OK;
}

int fcz_oracle_decompress_batch(const uint8_t* blob, const uint64_t* off, uint32_t n, const uint32_t* res_off,
                                const uint32_t* atom_off, int alt_order, const fcz_atoms_out* out, int n_threads) {
    (void)n_threads;
    long N = n;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (long i = 0; i < N; i++) {
        if (atom_off[i + 1] == atom_off[i]) continue;
        uint32_t a = atom_off[i], r = res_off[i];
        fcz_oracle_decompress_chain(blob + off[i], off[i + 1] - off[i], alt_order, out->x + a, out->y + a, out->z + a,
                                    out->bfac_res + r, out->res_code ? out->res_code + r : NULL,
                                    out->atom_code ? out->atom_code + a : NULL);
    }
    return FCZ_OK;
}

/* ---------------------------------------------------------------------------------------------
 * pins for the restated libm pieces (used by tests only)
 * ------------------------------------------------------------------------------------------- */
/* number of float bit patterns u in [lo_bits, hi_bits) (both signs) where the restated sinf/cosf differ
 * from the host libm */
long fcz_oracle_trig_mismatches(uint32_t lo_bits, uint32_t hi_bits, int n_threads) {
    long bad = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad) schedule(static, 1 << 18) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (uint32_t u = lo_bits; u < hi_bits; u++) {
        for (int sg = 0; sg < 2; sg++) {
            uint32_t v = u | ((uint32_t)sg << 31);
            float x, a, b;
            memcpy(&x, &v, 4);
            a = sinf(x); b = fcz_oracle_sinf(x);
            if (memcmp(&a, &b, 4)) bad++;
            a = cosf(x); b = fcz_oracle_cosf(x);
            if (memcmp(&a, &b, 4)) bad++;
        }
    }
    return bad;
}

/* host evaluation of the only way the codec observes acos: (float)(acos((double)c)*180.0/M_PI)
 * (src/torsion_angle.cpp:84, src/float3d.h:63) for count consecutive-by-stride float bit patterns */
void fcz_oracle_acos_deg_sweep(uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1 << 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (uint32_t i = 0; i < count; i++) {
        uint32_t v = start_bits + i * stride;
        float c; memcpy(&c, &v, 4);
        out[i] = (float)(acos((double)c) * 180.0 / M_PI);
    }
}
void fcz_oracle_sincos_sweep(int is_cos, uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1 << 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (uint32_t i = 0; i < count; i++) {
        uint32_t v = start_bits + i * stride;
        float c; memcpy(&c, &v, 4);
        out[i] = is_cos ? cosf(c) : sinf(c);
    }
}

/* host twin of k_selftest_math (foldcomp_amd/csrc/fcz_abi.hip): same hashed inputs, reference-ordered
 * evaluation with the host libm. modes: 3 deg2rad, 4 norm, 5 getCosineTheta, 6..8 place_atom x/y/z */
static float st_hash_float(uint32_t u, uint32_t salt, float scale) {
    uint32_t h = (u ^ salt) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(int32_t)h * scale;
}
void fcz_oracle_math_sweep(int mode, uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1 << 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (uint32_t i = 0; i < count; i++) {
        uint32_t u = start_bits + i * stride;
        float x; memcpy(&x, &u, 4);
        const float sc = 0x1p-29f;
        float r;
        if (mode == 3) r = (float)((double)x * M_PI / 180.0);
        else if (mode == 4) { v3 v = {st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)}; r = v_norm(v); }
        else if (mode == 5) {
            v3 a = {st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)};
            v3 b = {st_hash_float(u, 4, sc), st_hash_float(u, 5, sc), st_hash_float(u, 6, sc)};
            r = v_cos_theta(a, b);
        } else {
            v3 a = {st_hash_float(u, 1, sc), st_hash_float(u, 2, sc), st_hash_float(u, 3, sc)};
            v3 b = {st_hash_float(u, 4, sc), st_hash_float(u, 5, sc), st_hash_float(u, 6, sc)};
            v3 c = {st_hash_float(u, 7, sc), st_hash_float(u, 8, sc), st_hash_float(u, 9, sc)};
            float L = 1.2f + fabsf(st_hash_float(u, 10, 0x1p-33f));
            float ba = 90.0f + st_hash_float(u, 11, 0x1p-25f);
            float ta = st_hash_float(u, 12, 0x1.6p-24f);
            v3 d = place_atom(a, b, c, L, ba, ta);
            r = (mode == 6) ? d.x : (mode == 7) ? d.y : d.z;
        }
        out[i] = r;
    }
}
