// fcz_math.h -- device numerics of the FCZ codec for gfx950.
//
// Contract: every value that reaches an FCZ byte or an output coordinate is bit-identical to what the
// reference computes on x86-64/glibc 2.35. The reference mixes float and
// double at fixed points; those promotion points are reproduced literally. Compile this file with
// -ffp-contract=off (no implicit FMA); every fused operation below is an explicit __builtin_fma in
// code whose result is provably independent of the fusion (double-double error terms).
//
// Two host-libm functions sit on the path:
//   * acos(double) in the dihedral / bond-angle measurement (src/torsion_angle.cpp:77-84,
//     src/float3d.h:63). Only (float)(acos((double)c)*180.0/M_PI) of a *float* c is observable.
//     Device: fast double evaluation + a float-rounding safety test; inputs whose result lies within
//     2^-48 (relative) of a float rounding boundary are re-evaluated exactly (correctly rounded
//     double acos via a double-double cosine comparison). glibc's acos is correctly rounded except
//     for ~2^-17 of inputs, so the two agree except with probability ~1e-14 per value.
//   * sinf/cosf (float) in Nerf::place_atom (src/nerf.cpp:67-71). glibc 2.35 uses the
//     "optimized-routines" algorithm (double polynomial after one-step reduction); it is restated
//     here operation by operation. tests/test_oracle_trig.py shows the plain (non-FMA) evaluation
//     equals both glibc ifunc variants for every float |x| < 17.27, which covers every angle the
//     codec can produce.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcz {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }

// crossProduct, reference src/float3d.h:19-24
__device__ __forceinline__ v3 vcross(v3 a, v3 b) {
    v3 r;
    r.x = a.y * b.z - b.y * a.z;
    r.y = a.z * b.x - b.z * a.x;
    r.z = a.x * b.y - b.x * a.y;
    return r;
}

// ---- "round to float" fast paths --------------------------------------------------------------------
// Several reference expressions evaluate in double and are observed only after rounding to float
// (norm, getCosineTheta, the acos->degrees conversion, the degrees->radians conversion). IEEE-exact
// double sqrt/divide/acos cost 15-60 f64 instructions each on gfx950. Instead: evaluate a cheap double
// approximation v (relative error <= 2^-46, using v_rsq_f64 + one Newton step, explicit FMAs), round it
// to float, and prove the rounding is the one the exact expression would give: if v lies farther than
// 2^-41*|v| from the nearest float rounding boundary, every value within 2^-41 relative of v -- the
// exact double result included -- rounds to the same float. Otherwise (probability ~2^-16 per value)
// the exact, reference-ordered evaluation runs. The test is conservative by construction: zero,
// denormal, infinite and NaN results always take the exact path.
__device__ __forceinline__ bool round_to_float_is_safe(double v, float f) {
    // Rounding a double to a (normal) float drops the low 29 mantissa bits; the rounding boundary sits where those bits
    // read 0x10000000, in v's own binade whatever the binade of the result. 2^-41 |v| < 2^12 double ulps, so v is safe when
    // its dropped bits are more than 4096 away from the boundary: three integer instructions. Results outside the normal
    // float range (zero, denormal, overflow, NaN) drop a different number of bits: they take the exact path.
    const uint32_t lo = (uint32_t)__double2loint(v);
    const bool far = ((lo & 0x1fffffffu) - (0x10000000u - 4096u)) > 8192u;
    return far && __builtin_amdgcn_classf(f, 0x108);   // -normal | +normal
}

// norm, reference src/float3d.h:32-34 (double pow/sqrt, float result)
__device__ __noinline__ float vnorm_exact(float x, float y, float z) {
    double s = (double)x * (double)x + (double)y * (double)y + (double)z * (double)z;
    return (float)__builtin_sqrt(s);
}
__device__ __forceinline__ float vnorm(v3 v) {
    const double x = v.x, y = v.y, z = v.z;
    const double s = __builtin_fma(z, z, __builtin_fma(y, y, x * x));
    double r = __builtin_amdgcn_rsq(s);
    const double g = s * r;                       // ~sqrt(s)
    const double h = 0.5 * r;
    const double d = __builtin_fma(-g, h, 0.5);
    const double n = __builtin_fma(g, d, g);      // one Newton step: relative error ~2^-51
    float f = (float)n;
    if (__builtin_expect(!round_to_float_is_safe(n, f), 0)) f = vnorm_exact(v.x, v.y, v.z);
    return f;
}

// getCosineTheta, reference src/float3d.h:36-43: float dot products, double sqrt + divide, float result
__device__ __noinline__ float cos_theta_exact(float ip, float p) {
    return (float)((double)ip / __builtin_sqrt((double)p));
}
// the double part of getCosineTheta from its three float dot products (callers that evaluate several angles over shared
// vectors compute every dot product once)
__device__ __forceinline__ float vcos_theta_pre(float ip, float s1, float s2);
__device__ __forceinline__ float vdot_ref(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }   // float, left to right
__device__ __forceinline__ float vcos_theta(v3 a, v3 b) { return vcos_theta_pre(vdot_ref(a, b), vdot_ref(a, a), vdot_ref(b, b)); }
__device__ __forceinline__ float vcos_theta_pre(float ip, float s1, float s2) {
    const float p = s1 * s2;
    const double pd = (double)p;
    double r = __builtin_amdgcn_rsq(pd);
    const double t = pd * r;
    const double d = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(0.5 * r, d, r);             // 1/sqrt(p), relative error ~2^-51
    const double q = (double)ip * r;
    float f = (float)q;
    if (__builtin_expect(!round_to_float_is_safe(q, f), 0)) f = cos_theta_exact(ip, p);
    return f;
}

// ---- double-double helpers (slow path only) ---------------------------------------------------
struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd dd_quick_two_sum(double a, double b) {
    double s = a + b;
    return dd{s, b - (s - a)};
}
__device__ __forceinline__ dd dd_two_prod(double a, double b) {
    double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    dd t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div_d(dd a, double b) {
    double q1 = a.hi / b;
    dd p = dd_two_prod(q1, b);
    dd r = dd_add(a, dd_neg(p));
    double q2 = r.hi / b;
    p = dd_two_prod(q2, b);
    r = dd_add(r, dd_neg(p));
    double q3 = r.hi / b;
    dd q = dd_quick_two_sum(q1, q2);
    return dd_add(q, dd{q3, 0.0});
}

// cos(t) for a double-double t in [0, pi], ~100 bits: cos t = -sin(t - pi/2), Taylor series.
__device__ __noinline__ dd dd_cos_0_pi(dd t) {
    const dd half_pi = dd{0x1.921fb54442d18p+0, 0x1.1a62633145c07p-54};
    dd y = dd_add(t, dd_neg(half_pi));
    dd y2 = dd_mul(y, y);
    dd term = y, sum = y;
    for (int k = 1; k <= 22; k++) {
        term = dd_div_d(dd_mul(term, y2), (double)((2 * k) * (2 * k + 1)));
        sum = (k & 1) ? dd_add(sum, dd_neg(term)) : dd_add(sum, term);
    }
    return dd_neg(sum);
}

__device__ __forceinline__ double d_next_up(double a) {   // a > 0 finite
    return __longlong_as_double(__double_as_longlong(a) + 1);
}
__device__ __forceinline__ double d_next_down(double a) { // a > 0 finite
    return __longlong_as_double(__double_as_longlong(a) - 1);
}

// Correctly rounded acos of a float c in (-1, 1), c != 0 excluded nowhere: start from an
// approximation a0 and walk to the double nearest to acos(c) using exact midpoint tests
// acos(c) < m  <=>  c > cos(m)  (cos is decreasing on [0, pi]).
__device__ __noinline__ double acos_correctly_rounded(float c, double a0) {
    double a = a0;
    for (int it = 0; it < 8; it++) {
        double up = d_next_up(a), dn = d_next_down(a);
        // midpoints as exact double-doubles
        dd m_up = dd_two_sum(a, 0.5 * (up - a));
        dd m_dn = dd_two_sum(a, -0.5 * (a - dn));
        dd cu = dd_cos_0_pi(m_up);
        dd du = dd_add(cu, dd{-(double)c, 0.0});       // cos(m_up) - c
        if (!(du.hi < 0.0 || (du.hi == 0.0 && du.lo < 0.0))) { a = up; continue; }   // acos c >= m_up
        dd cd = dd_cos_0_pi(m_dn);
        dd dl = dd_add(cd, dd{-(double)c, 0.0});       // cos(m_dn) - c
        if (dl.hi < 0.0 || (dl.hi == 0.0 && dl.lo < 0.0)) { a = dn; continue; }       // acos c < m_dn
        break;
    }
    return a;
}

// (float)(acos((double)c) * 180.0 / M_PI), the only way the reference observes acos.
// Returns NaN for |c| > 1 or NaN input (callers apply the reference's NaN guard where it has one).
// NaN results carry the bits x86-64 / glibc give them, because the quantiser writes a NaN minimum into the record's header as it
// is (src/discretizer.cpp:22-33): a NaN cosine -- made by the arithmetic from finite coordinates (0 / 0 of a zero-length bond,
// inf - inf of an overflow): x86-64's "real indefinite", the NEGATIVE quiet NaN -- comes back from acos as it went in
// (0xFFC00000 as a float); a finite cosine beyond +-1 (float rounding of nearly collinear bonds: 1.0000001) is glibc's domain
// error, the POSITIVE quiet NaN (0x7FC00000). gfx950 would make the positive one in both cases.
__device__ __noinline__ float acos_deg_exact(float c) {
    if (c != c) return __uint_as_float(0xFFC00000u);
    if (__builtin_fabsf(c) > 1.0f) return __uint_as_float(0x7FC00000u);
    const double kPi = 3.14159265358979323846;
    double A = __ocml_acos_f64((double)c);
    double D = A * 180.0 / kPi;
    float f = (float)D;
    // float-rounding safety: distance of D to the rounding boundary on its side of f
    double e = D - (double)f;
    float other = __uint_as_float(__float_as_uint(f) + ((e > 0.0) ? 1 : -1));
    double mid = 0.5 * ((double)f + (double)other);
    double dist = __builtin_fabs(D - mid);
    // 32 ulp(double) of D; D in (0, 180]. (f == 0 only for c == 1 where A is exactly 0.)
    if (__builtin_expect(dist < D * 0x1p-47 && f > 0.0f, 0)) {
        A = acos_correctly_rounded(c, A);
        D = A * 180.0 / kPi;
        f = (float)D;
    }
    return f;
}

// (not const: the values must stay loads -- see acos_deg)
__device__ __constant__ double fcz_acos_coef[11] = {
    0x1.c8a4a8d5d7026p-6, -0x1.bf16e7c9f283cp-8, 0x1.fa1b2b4831188p-7, 0x1.512bc40e88a9ep-7, 0x1.cf5ed14c7cb7ep-7, 0x1.1c0d74beb3610p-6,
    0x1.6e8f34a32a3ecp-6, 0x1.f1c6ff7f5507fp-6, 0x1.6db6dba99e56dp-5, 0x1.33333333030cfp-4, 0x1.55555555555bbp-3};

// Fast path of the same function: asin kernel polynomial (degree 10 in z, minimax-fitted with mpmath,
// relative error 2^-50 on z in [0, 1/4]) with the usual reduction acos(x) = 2 asin(sqrt((1-|x|)/2)) for
// |x| > 1/2, one multiply by 180/pi, then the float-rounding safety test. ~45 f64 instructions.
__device__ __forceinline__ float acos_deg(float c) {
    const double x = (double)c;
    const double ax = __builtin_fabs(x);
    const bool big = ax > 0.5;
    const double z = big ? __builtin_fma(-0.5, ax, 0.5) : x * x;
    double sq;
    {
        double r = __builtin_amdgcn_rsq(z);
        const double g = z * r, h = 0.5 * r;
        const double d = __builtin_fma(-g, h, 0.5);
        sq = __builtin_fma(g, d, g);
    }
    const double s = big ? sq : ax;
#ifdef FCZ_ACOS_LITERALS
    double P = 0x1.c8a4a8d5d7026p-6;
    P = __builtin_fma(P, z, -0x1.bf16e7c9f283cp-8);
    P = __builtin_fma(P, z, 0x1.fa1b2b4831188p-7);
    P = __builtin_fma(P, z, 0x1.512bc40e88a9ep-7);
    P = __builtin_fma(P, z, 0x1.cf5ed14c7cb7ep-7);
    P = __builtin_fma(P, z, 0x1.1c0d74beb3610p-6);
    P = __builtin_fma(P, z, 0x1.6e8f34a32a3ecp-6);
    P = __builtin_fma(P, z, 0x1.f1c6ff7f5507fp-6);
    P = __builtin_fma(P, z, 0x1.6db6dba99e56dp-5);
    P = __builtin_fma(P, z, 0x1.33333333030cfp-4);
    P = __builtin_fma(P, z, 0x1.55555555555bbp-3);
#else
    // the coefficients come from constant memory (scalar loads -> SGPR pairs): a 64-bit literal cannot be an operand, and as
    // literals the compiler moved every coefficient into the accumulator's VGPR pair first (two v_mov per FMA: a fifth of
    // k_compress_pack's instructions, where this function is inlined 36 times)
    double P = fcz_acos_coef[0];
#pragma unroll
    for (int i = 1; i < 11; i++) P = __builtin_fma(P, z, fcz_acos_coef[i]);
#endif
    const double as = __builtin_fma(s * z, P, s);          // asin(s)
    const double kPi = 3.14159265358979323846, kPio2 = 1.57079632679489661923;
    const double A = big ? ((x > 0.0) ? 2.0 * as : __builtin_fma(-2.0, as, kPi))
                         : ((x > 0.0) ? kPio2 - as : kPio2 + as);
    const double D = A * 57.29577951308232087680;          // 180/pi
    float f = (float)D;
    // |c| >= 1, NaN, and every result the safety test cannot certify go through the exact evaluation
    if (__builtin_expect(!(ax < 1.0) || !round_to_float_is_safe(D, f), 0)) f = acos_deg_exact(c);
    return f;
}

// one window of getTorsionFromXYZ, reference src/torsion_angle.cpp:50-94, in two steps: the float part (cosine between the
// two plane normals, sign test) and the double part (acos -> degrees, NaN guard)
struct dih_parts { float ct; bool neg; };
__device__ __forceinline__ dih_parts dihedral_parts(v3 a, v3 b, v3 c, v3 d) {
    v3 d1 = vsub(b, a), d2 = vsub(c, b), d3 = vsub(d, c);
    v3 u1 = vcross(d1, d2), u2 = vcross(d2, d3);
    dih_parts p;
    p.ct = vcos_theta(u1, u2);
    v3 w = vcross(u2, d2);
    p.neg = (u1.x * w.x) + (u1.y * w.y) + (u1.z * w.z) < 0.0f;
    return p;
}
__device__ __forceinline__ float dihedral_finish(dih_parts p) {
    float t = acos_deg(p.ct);
    if (t != t) t = (p.ct < 0.0f) ? 180.0f : 0.0f;   // isnan(acos) guard, :77-84
    if (p.neg) t = -1.0f * t;
    return t;
}
__device__ __forceinline__ float dihedral_deg(v3 a, v3 b, v3 c, v3 d) { return dihedral_finish(dihedral_parts(a, b, c, d)); }

// acos in degrees in plain float arithmetic for values that are only observed through a coarse quantiser (the side-chain
// torsion byte: 256 bins of 1.41 degrees). asin kernel x + x z P(z) (degree 5 in z, fitted on [0, 1/4], 5e-10) with the
// reduction acos x = 2 asin(sqrt((1-|x|)/2)) for |x| > 1/2. Absolute error below 1e-4 degrees for every float in (-1, 1)
// (tests/test_device_math.py sweeps it against the exact acos_deg); the caller keeps a guard band around the bin edges and falls
// back to acos_deg inside it. Requires |c| < 1.
__device__ __forceinline__ float acos_deg_f32(float c) {
    const float ax = __builtin_fabsf(c);
    const bool big = ax > 0.5f;
    const float z = big ? __builtin_fmaf(-0.5f, ax, 0.5f) : c * c;
    const float s = big ? __builtin_amdgcn_sqrtf(z) : ax;
    float P = 0x1.14f022p-5f;
    P = __builtin_fmaf(P, z, 0x1.17cd44p-6f);
    P = __builtin_fmaf(P, z, 0x1.fdcebcp-6f);
    P = __builtin_fmaf(P, z, 0x1.6d58d8p-5f);
    P = __builtin_fmaf(P, z, 0x1.33343cp-4f);
    P = __builtin_fmaf(P, z, 0x1.555554p-3f);
    const float as = __builtin_fmaf(s * z, P, s);
    const float kPi = 3.14159274f, kPio2 = 1.57079637f;
    const float A = big ? ((c > 0.0f) ? 2.0f * as : __builtin_fmaf(-2.0f, as, kPi)) : ((c > 0.0f) ? kPio2 - as : kPio2 + as);
    return A * 57.2957802f;
}

// The side-chain torsion byte (FixedAngleDiscretizer(255).discretize of a dihedral, reference src/foldcomp.cpp:532-538,
// src/discretizer.h:89-106): (unsigned)((t + 180) * (255/360)) truncated. The float part of the dihedral is evaluated exactly;
// the angle itself first in float (error < 1e-4 degrees = 7.1e-5 bins; the subtraction and the multiplication round by at most
// 3.6e-5 bins on either side): when (t~ + 180) * disc lies farther than 3e-4 from every integer, the exact evaluation truncates
// to the same byte. Otherwise (0.06 % of values), and for |cos| >= 1 or NaN, the exact evaluation runs.
__device__ __forceinline__ uint32_t sidechain_torsion_byte(v3 a, v3 b, v3 c, v3 d) {
    const float sc_min = -180.0f, sc_disc = 255.0f / (180.0f - (-180.0f));
    const dih_parts p = dihedral_parts(a, b, c, d);
    const float th = acos_deg_f32(p.ct);
    const float v = p.neg ? -th : th;
    const float f = (v - sc_min) * sc_disc;
    const bool certain = __builtin_fabsf(p.ct) < 1.0f && __builtin_fabsf(f - __builtin_rintf(f)) > 3e-4f;
    uint32_t q = __float2uint_rz(f);
    if (__builtin_expect(!certain, 0)) {
        const float fe = (dihedral_finish(p) - sc_min) * sc_disc;
        q = (fe != fe) ? 0u : __float2uint_rz(fe);
    }
    return q;
}

// angle, reference src/float3d.h:55-65 (no NaN guard)
__device__ __forceinline__ float bond_angle_deg(v3 a, v3 b, v3 c) {
    v3 d1 = vsub(a, b), d2 = vsub(c, b);
    return acos_deg(vcos_theta(d1, d2));
}

// ---- glibc 2.35 sinf/cosf, |x| < 120 (sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h}) -----
__device__ __forceinline__ float sc_poly(double x, double x2, int n, bool neg_cos) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
        double x7 = x3 * x2;
        double s = x + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * s1);
    } else {
        double sg = neg_cos ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = sg * -0x1.6c087e89a359dp-10 + x2 * (sg * 0x1.99343027bf8c3p-16);
        double c1 = sg * 0x1p0 + x2 * (sg * -0x1.ffffffd0c621cp-2);
        double x6 = x4 * x2;
        double c = c1 + x4 * (sg * 0x1.55553e1068f19p-5);
        return (float)(c + x6 * c2);
    }
}
__device__ __forceinline__ uint32_t abstop12(float f) { return (__float_as_uint(f) >> 20) & 0x7ffu; }

// sin (is_cos = 0) or cos (is_cos = 1) of a float |y| < 120
__device__ __forceinline__ float sincosf_glibc(float y, int is_cos) {
    double x = (double)y;
    if (abstop12(y) < 0x3f4u) {            // abstop12(pi/4)
        double x2 = x * x;
        if (abstop12(y) < 0x398u)          // abstop12(0x1p-12f)
            return is_cos ? 1.0f : y;
        return sc_poly(x, x2, is_cos, false);
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    double s = ((n + 1) & 2) ? -1.0 : 1.0;  // sign table {1,-1,-1,1}[n & 3]
    return sc_poly(x * s, x * x, n ^ is_cos, (n & 2) != 0);
}
__device__ __forceinline__ float sinf_glibc(float y) { return sincosf_glibc(y, 0); }
__device__ __forceinline__ float cosf_glibc(float y) { return sincosf_glibc(y, 1); }
// sinf(y) and cosf(y) together, bit-identical to the two calls, without a lane-divergent branch: the reduction also
// serves |y| < pi/4 (it yields n = 0, s = 1 and x unchanged, i.e. exactly the operands of the small-argument branch),
// both polynomials are evaluated once and the quadrant only selects which one is the sine. glibc multiplies the argument of
// the sine polynomial and every coefficient of the cosine polynomial by a sign (+-1); every operation of both polynomials
// is odd in that sign and IEEE rounding is symmetric, so the polynomials are evaluated unsigned and the sign bit is applied to
// the float result: same bits, seven multiplications and two double selects fewer. |y| < 2^-12 returns (y, 1) as glibc does.
__device__ __forceinline__ void sincosf_pair(float y, float* sn, float* cs) {
    double x = (double)y;
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double x2 = x * x;
    const uint32_t ps = __float_as_uint(sc_poly(x, x2, 0, false)) ^ (((uint32_t)(n + 1) & 2u) << 30);   // sign table {1,-1,-1,1}[n & 3]
    const uint32_t pc = __float_as_uint(sc_poly(x, x2, 1, false)) ^ (((uint32_t)n & 2u) << 30);         // negated cosine table for n & 2
    const bool odd = (n & 1) != 0, tiny = abstop12(y) < 0x398u;
    const uint32_t s0 = odd ? pc : ps, c0 = odd ? ps : pc;
    *sn = tiny ? y : __uint_as_float(s0);
    *cs = tiny ? 1.0f : __uint_as_float(c0);
}

// ---- the same for ANY float: |y| >= 120 (glibc's reduce_large, sysdeps/ieee754/flt-32/sincosf.h), infinities, NaN ------
// No compressor writes such angles (they come from quantiser parameters of a record that was made by hand or damaged: minimum +
// q * step of thousands of degrees), but the reference decodes those records to numbers, and so does this. reduce_large
// multiplies the 24-bit mantissa by a 96-bit window of 4/pi chosen by the exponent: the quadrant is the top two bits of the
// product, the reduced argument its remaining 62 bits as a signed fraction of pi/2. Out of line, behind a test the decoder makes
// once per chain (backbone_group: chains whose parameters cannot reach 120 radians never look).
__device__ __constant__ uint32_t fcz_inv_pio4[24] = {
    0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
    0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
__device__ __noinline__ void sincosf_any_slow(float y, float* sn, float* cs) {
    uint32_t xi = __float_as_uint(y);
    if ((xi & 0x7f800000u) == 0x7f800000u) {
        // __math_invalidf: (y - y) / (y - y) -- a NaN goes through, an infinity makes x86-64's negative quiet NaN
        const float r = (xi & 0x007fffffu) ? __uint_as_float(xi | 0x00400000u) : __uint_as_float(0xFFC00000u);
        *sn = r; *cs = r;
        return;
    }
    const int sign = (int)(xi >> 31);
    const uint32_t* arr = &fcz_inv_pio4[(xi >> 26) & 15u];
    const int shift = (int)((xi >> 23) & 7u);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    unsigned long long res0 = (unsigned long long)(uint32_t)(xi * arr[0]);           // (a 32-bit product, as the reference computes it)
    const unsigned long long res1 = (unsigned long long)xi * arr[4], res2 = (unsigned long long)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const unsigned long long nq = (res0 + (1ull << 61)) >> 62;
    res0 -= nq << 62;
    const double x = (double)(long long)res0 * 0x1.921FB54442D18p-62;
    const int n = (int)nq, m = n + sign;
    const double sg = ((m + 1) & 2) ? -1.0 : 1.0;                                    // sign table {1, -1, -1, 1}[(n + sign) & 3]
    const bool negc = (m & 2) != 0;
    *sn = sc_poly(x * sg, x * x, n, negc);
    *cs = sc_poly(x * sg, x * x, n ^ 1, negc);
}
__device__ __forceinline__ void sincosf_pair_any(float y, float* sn, float* cs) {
    if (__builtin_expect(abstop12(y) >= 0x42fu, 0)) sincosf_any_slow(y, sn, cs);     // abstop12(120.0f)
    else sincosf_pair(y, sn, cs);
}

// degrees -> radians as Nerf::place_atom does (src/nerf.cpp:63-64): double multiply, double divide,
// rounded to float on assignment. Fast path: one multiply by pi/180 + the float-rounding safety test.
__device__ __noinline__ float deg2rad_exact(float deg) {
    const double kPi = 3.14159265358979323846;
    return (float)((double)deg * kPi / 180.0);
}
__device__ __forceinline__ float deg2rad(float deg) {
    const double r = (double)deg * 0.01745329251994329577;
    float f = (float)r;
    if (__builtin_expect(!round_to_float_is_safe(r, f), 0)) f = deg2rad_exact(deg);
    return f;
}

// x/d, y/d, z/d with ONE reciprocal. This is the unscaled core of the IEEE-correct float division the
// compiler emits for '/' (v_rcp_f32, one FMA refinement of the reciprocal, two of the quotient, final FMA;
// LLVM AMDGPU LowerFDIV32 between v_div_scale and v_div_fixup); those two wrappers only rescale operands whose
// quotient or intermediates could leave the normal range. The guard below sends every such case -- and zero,
// infinite or NaN denominators -- to the plain division, so results are bit-identical to three '/'.
__device__ __noinline__ v3 vdiv3_plain(v3 n, float d) { return v3{n.x / d, n.y / d, n.z / d}; }
__device__ __forceinline__ v3 vdiv3(v3 n, float d) {
    const uint32_t ed = (__float_as_uint(d) >> 23) & 0xffu;
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    v3 q;
    {
        float t = n.x * r; float e2 = __builtin_fmaf(-d, t, n.x); t = __builtin_fmaf(e2, r, t);
        const float e3 = __builtin_fmaf(-d, t, n.x); q.x = __builtin_fmaf(e3, r, t);
    }
    {
        float t = n.y * r; float e2 = __builtin_fmaf(-d, t, n.y); t = __builtin_fmaf(e2, r, t);
        const float e3 = __builtin_fmaf(-d, t, n.y); q.y = __builtin_fmaf(e3, r, t);
    }
    {
        float t = n.z * r; float e2 = __builtin_fmaf(-d, t, n.z); t = __builtin_fmaf(e2, r, t);
        const float e3 = __builtin_fmaf(-d, t, n.z); q.z = __builtin_fmaf(e3, r, t);
    }
    // safe iff the denominator is comfortably normal and every quotient is a number between 2^-90 and 2^90. A ZERO quotient is not
    // safe: the refinement steps lose its sign (fma(-d, -0, -0) = +0, then fma(+0, r, -0) = +0) where '/' keeps it -- a numerator
    // of -0 is what the blend of a segment's first atom makes of an anchor coordinate written "-0.000" (round 6: found by
    // tests/test_gpu_edge_cases.py::test_distorted_geometry, the decoded 0.0 had the wrong sign bit). Nor is an INFINITE one: an
    // infinite numerator (an anchor of a damaged record) leaves the refinement as NaN (inf - inf) where '/' gives the infinity
    // (tools/dbg/param_fuzz.py). On the bit patterns with the sign shifted out one unsigned minimum and one unsigned maximum
    // cover all three components: exponent fields 37 .. 216 pass, zeros, denormals, infinities and NaNs do not.
    const bool den_ok = (ed - 32u) < 192u;                                   // 2^-95 <= |d| < 2^97
    const uint32_t ux = __float_as_uint(q.x) << 1, uy = __float_as_uint(q.y) << 1, uz = __float_as_uint(q.z) << 1;
    const uint32_t lo2 = ux < uy ? ux : uy, lo3 = lo2 < uz ? lo2 : uz;                                          // v_min3_u32
    const uint32_t hi2 = ux > uy ? ux : uy, hi3 = hi2 > uz ? hi2 : uz;                                          // v_max3_u32
    const bool q_ok = lo3 >= (37u << 24) && hi3 < (217u << 24);              // 2^-90 = exponent field 37, 2^90 = 217
    if (__builtin_expect(!(den_ok && q_ok), 0)) q = vdiv3_plain(n, d);
    return q;
}

// Nerf::place_atom, reference src/nerf.cpp:39-104, with the trigonometry hoisted: d2 is the
// "curr_atm" vector (-L cos(ba), L cos(ta) sin(ba), L sin(ta) sin(ba)) (:66-70).
__device__ __forceinline__ v3 place_atom_d2(v3 a, v3 b, v3 c, v3 d2) {
    v3 ab = vsub(b, a), bc = vsub(c, b);
    float bc_norm = vnorm(bc);
    v3 bcn = vdiv3(bc, bc_norm);
    v3 n = vcross(ab, bcn);
    float n_norm = vnorm(n);
    n = vdiv3(n, n_norm);
    v3 nbc = vcross(n, bcn);
    v3 D = v3{0.0f, 0.0f, 0.0f};
    D.x += (bcn.x * d2.x); D.x += (nbc.x * d2.y); D.x += (n.x * d2.z);
    D.y += (bcn.y * d2.x); D.y += (nbc.y * d2.y); D.y += (n.y * d2.z);
    D.z += (bcn.z * d2.x); D.z += (nbc.z * d2.y); D.z += (n.z * d2.z);
    D.x += c.x; D.y += c.y; D.z += c.z;
    return D;
}

// the same placement in plain float arithmetic with hardware reciprocal square roots (FCZ_NUMERICS_FAST): equal to
// place_atom_d2 up to float rounding (~1e-6 A per atom)
__device__ __forceinline__ v3 place_atom_d2_fast(v3 a, v3 b, v3 c, v3 d2) {
    const v3 ab = vsub(b, a), bc = vsub(c, b);
    const float rb = __builtin_amdgcn_rsqf(__builtin_fmaf(bc.z, bc.z, __builtin_fmaf(bc.y, bc.y, bc.x * bc.x)));
    const v3 bcn = v3{bc.x * rb, bc.y * rb, bc.z * rb};
    v3 n = v3{__builtin_fmaf(ab.y, bcn.z, -(bcn.y * ab.z)), __builtin_fmaf(ab.z, bcn.x, -(bcn.z * ab.x)), __builtin_fmaf(ab.x, bcn.y, -(bcn.x * ab.y))};
    const float rn = __builtin_amdgcn_rsqf(__builtin_fmaf(n.z, n.z, __builtin_fmaf(n.y, n.y, n.x * n.x)));
    n = v3{n.x * rn, n.y * rn, n.z * rn};
    const v3 nbc = v3{__builtin_fmaf(n.y, bcn.z, -(bcn.y * n.z)), __builtin_fmaf(n.z, bcn.x, -(bcn.z * n.x)), __builtin_fmaf(n.x, bcn.y, -(bcn.x * n.y))};
    v3 D;
    D.x = __builtin_fmaf(n.x, d2.z, __builtin_fmaf(nbc.x, d2.y, __builtin_fmaf(bcn.x, d2.x, c.x)));
    D.y = __builtin_fmaf(n.y, d2.z, __builtin_fmaf(nbc.y, d2.y, __builtin_fmaf(bcn.y, d2.x, c.y)));
    D.z = __builtin_fmaf(n.z, d2.z, __builtin_fmaf(nbc.z, d2.y, __builtin_fmaf(bcn.z, d2.x, c.z)));
    return D;
}

__device__ __forceinline__ v3 nerf_d2(float L, float bond_angle_deg_, float torsion_deg) {
    const float ba = deg2rad(bond_angle_deg_), ta = deg2rad(torsion_deg);
    float sb, cb, st, ct;
    sincosf_pair(ba, &sb, &cb);
    sincosf_pair(ta, &st, &ct);
    v3 d2;
    d2.x = -1.0f * L * cb;
    d2.y = L * ct * sb;
    d2.z = L * st * sb;
    return d2;
}

// the same with the torsion's cosine and sine already known (they depend on the torsion alone, so the reverse pass of
// the backbone reuses the forward pass's values)
__device__ __forceinline__ v3 nerf_d2_trig(float L, float bond_angle_deg_, float ct, float st) {
    const float ba = deg2rad(bond_angle_deg_);
    float sb, cb;
    sincosf_pair(ba, &sb, &cb);
    v3 d2;
    d2.x = -1.0f * L * cb;
    d2.y = L * ct * sb;
    d2.z = L * st * sb;
    return d2;
}

// (ANY: the bond angle may be any float -- see sincosf_pair_any)
template <bool ANY>
__device__ __forceinline__ v3 nerf_d2_trig_t(float L, float bond_angle_deg_, float ct, float st) {
    const float ba = deg2rad(bond_angle_deg_);
    float sb, cb;
    if (ANY) sincosf_pair_any(ba, &sb, &cb); else sincosf_pair(ba, &sb, &cb);
    v3 d2;
    d2.x = -1.0f * L * cb;
    d2.y = L * ct * sb;
    d2.z = L * st * sb;
    return d2;
}

__device__ __forceinline__ v3 place_atom(v3 a, v3 b, v3 c, float L, float ba_deg, float ta_deg) {
    return place_atom_d2(a, b, c, nerf_d2(L, ba_deg, ta_deg));
}

// ---- quantisers, reference src/discretizer.cpp ----------------------------------------------------
// vector discretize (:43-53): float product, double +0.5, truncation; NaN -> 0 like x86-64 gcc
// `(unsigned int)d` of x86-64 gcc is cvttsd2si into a 64-bit register, low half taken: values of [2^32, 2^63) wrap, negative ones
// wrap as two's complement, NaN, infinities and everything from 2^63 on give the "integer indefinite" 0x8000000000000000 = 0.
// The hardware conversion here saturates. Out of line: a quantiser's operand is in [0.5, bins + 1) unless its step is degenerate
// (a chain whose B-factors are all denormal: 255 / (max - min) is infinite -- round 6, the differential fuzz; the reference
// writes bytes of 0 there, the saturating conversion wrote 255)
__device__ __noinline__ uint32_t cvt_u32_x86_slow(double d) {
    if (!(__builtin_fabs(d) < 9223372036854775808.0)) return 0u;
    return (uint32_t)(unsigned long long)(long long)d;
}
__device__ __forceinline__ uint32_t quant_round(float v, float mn, float disc_f) {
    const double d = (double)((v - mn) * disc_f) + 0.5;
    // v lies in [min, max] and disc = bins / (max - min): the operand is in [0.5, bins + 1) -- or a NaN (a NaN angle; inf * 0 of a
    // range that overflowed) -- unless disc itself is not finite. That test is the chain's, not the residue's (loop-invariant)
    uint32_t q = (d != d) ? 0u : __double2uint_rz(d);
    if (__builtin_expect(!(__builtin_fabsf(disc_f) < __builtin_huge_valf()), 0)) q = cvt_u32_x86_slow(d);
    return q;
}
// scalar discretize (:55-57): truncation of the float product
__device__ __forceinline__ uint32_t quant_trunc(float v, float mn, float disc_f) {
    float f = (v - mn) * disc_f;
    return (f != f) ? 0u : __float2uint_rz(f);
}
__device__ __forceinline__ float dequant(uint32_t q, float mn, float cont_f) { return ((float)q * cont_f) + mn; }

}  // namespace fcz
