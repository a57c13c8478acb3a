// gm_math.h — device-side f32 geometry of the order-parameter path (gfx950).
//
// Everything here is IEEE f32 with NO implicit fused multiply-add: the reference is Rust, and rustc
// never contracts a*b+c.  The translation unit is built with -ffp-contract=off and
// -fhip-fp32-correctly-rounded-divide-sqrt; fmaf() appears only inside the trig kernels, where the
// operation sequence itself is the definition (restated operation-for-operation by the oracle's
// MIRROR mode, oracle/gorder_oracle.c, so that device and oracle i64 sums compare EQUAL).
//
// Reference semantics (file:line under /root/reference):
//   vector_to / min image   src/analysis/pbc.rs:378-385 (PBC3D), :188-190 (NoPBC)   [groan_rs]
//   angle                   src/analysis/mod.rs:79                                    [nalgebra]
//   calc_sch                src/analysis/mod.rs:78-82
//   OrderValue::from(f32)   src/analysis/order.rs:21-26
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

#define GM_MI_MAX_ITER 8

// 1-D minimum image: shift by whole box lengths into [-L/2, L/2] (groan_rs).  `bad` is raised
// instead of spinning when the reference's `while` loop would need more than GM_MI_MAX_ITER steps.
// This is the literal loop; gm_min_image below is the same function with a branch-free first step.
__device__ __forceinline__ float gm_min_image_loop(float dx, float L, int &bad) {
    const float half = L / 2.0f;
    int it = 0;
#pragma clang loop unroll(disable)
    while (dx > half) {
        dx -= L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return dx; }
    }
    it = 0;
#pragma clang loop unroll(disable)
    while (dx < -half) {
        dx += L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return dx; }
    }
    return dx;
}

// One iteration of each `while`, as selects; `slow` is raised when that was not enough (the atoms are
// more than 1.5 box lengths apart), in which case the caller re-runs gm_min_image_loop on the input.
// The literal form is  a = dx > half ? dx - L : dx;  b = a < -half ? a + L : a.  If the first shift
// happened then a = RN(dx - L) >= -half (rounding is monotonic and -half is a float), so the second
// cannot; hence b is one select chain on dx itself.
__device__ __forceinline__ float gm_min_image_step(float dx, float L, bool &slow) {
    const float half = L / 2.0f;
    // dx - copysign(L, dx) is dx - L for dx > 0 and dx + L for dx < 0 (x + L == x - (-L) bit for bit)
    const float r = __builtin_fabsf(dx) > half ? dx - __builtin_copysignf(L, dx) : dx;
    // one shift was enough iff the result lies in [-half, half] (NaN: loops would not iterate either)
    slow = slow || (__builtin_fabsf(r) > half);
    return r;
}

// |gm_min_image_step(dx)| for callers that only square the component (gm_sch_axis): dx - copysign(L, dx) is
// -(L - |dx|) or +(L - |dx|), so its magnitude is |L - |dx|| bit for bit — a subtraction and a select instead of a
// copysign, a subtraction and a select.  Same `slow` condition.
__device__ __forceinline__ float gm_min_image_step_abs(float dx, float L, bool &slow) {
    const float half = L / 2.0f, a = __builtin_fabsf(dx);
    const float m = a > half ? L - a : a;
    slow = slow || (__builtin_fabsf(m) > half);
    return m;
}

__device__ __forceinline__ float gm_min_image(float dx, float L, int &bad) {
    bool slow = false;
    const float r = gm_min_image_step(dx, L, slow);
    if (__builtin_expect(slow, 0)) return gm_min_image_loop(dx, L, bad);
    return r;
}

// the periodic image of x nearest to ref: x itself shifted by whole box lengths while |x - ref| > L/2
// (oracle: nearest_image).  Used where a coordinate, not a displacement, is averaged (centre of a group).
__device__ __forceinline__ float gm_nearest_image(float x, float ref, float L, int &bad) {
    const float half = L / 2.0f;
    float d = x - ref;
    int it = 0;
#pragma clang loop unroll(disable)
    while (d > half) { d -= L; x -= L; if (++it > GM_MI_MAX_ITER) { bad = 1; return x; } }
    it = 0;
#pragma clang loop unroll(disable)
    while (d < -half) { d += L; x += L; if (++it > GM_MI_MAX_ITER) { bad = 1; return x; } }
    return x;
}

// groan_rs Vector3D::wrap into [0, L]
__device__ __forceinline__ float gm_wrap(float x, float L, int &bad) {
    int it = 0;
    while (x > L) {
        x -= L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return x; }
    }
    it = 0;
    while (x < 0.0f) {
        x += L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return x; }
    }
    return x;
}

// ---- acos, cos, sin as the reference's libm computes them -------------------------------------------------------------
// The reference evaluates `angle = acos(clamp(c))`, `angle.cos()` (mod.rs:78-82) and, for the unsaturated united-atom
// carbon, `sin_cos` of pi - gamma / 2 (uaorder.rs:1024-1045) with Rust's f32 methods, which on linux-gnu are glibc's
// acosf / cosf / sinf.  Rounds 1-3 used own polynomial kernels here (< 1 ulp, 1.6 % of the ticks one off).  Since round 4
// these are RESTATEMENTS OF GLIBC'S ALGORITHMS (2.28 - 2.40: fdlibm's e_acosf.c; s_cosf.c / s_sinf.c of 2018, double
// precision polynomials after a one-multiplication range reduction), operation for operation, so that the literal mode
// (GORDER_FLAG_TRIG_ACOS_COS) and the united-atom construction give the libm's bits: the oracle restates the same
// sequences (gorder_oracle_mirror_*), tests/test_oracle_kat.py compares those with the host's acosf / cosf / sinf over
// their whole domains (0 mismatches on glibc 2.35; tools/microbench/libm_restatement.c is the exhaustive form), and the
// device's sums are EQUAL to the oracle's LIBM mode.  Only the ranges the path can produce are implemented: acos on
// [-1, 1] (else NaN), cos and sin on [0, pi] (the range of acos) and NaN.
#define GM_ACOS_PI 3.1415925026e+00f
#define GM_ACOS_PIO2_HI 1.5707962513e+00f
#define GM_ACOS_PIO2_LO 7.5497894159e-08f

__device__ __forceinline__ float gm_div_core(float n, float d);
__device__ __forceinline__ float gm_sqrt_core(float x);

// glibc sysdeps/ieee754/flt-32/e_acosf.c.  Branch-free (a wave almost always holds lanes of every range): z, the
// rational p / q and the square root are common to the ranges, the three tails are selected.
// CORES: the division p / q, the square root and the division of the x > 1/2 range by their Newton cores (gm_sqrt_core /
// gm_div_core below).  Same bits: q lies in (0.3, 1]; where p is so small that v_div_fixup would matter (|x| < 2^-50) the
// quotient does not reach the result; for |x| >= 1/2, z = (1 - |x|) / 2 is 0 or lies in [2^-25, 1/4], the divisor s + df
// in [2^-12, 1] and z - df^2 is 0 or at least 2^-36 z (results for |x| = 1 and |x| <= 2^-57 are constants anyway).
template <bool CORES = false>
__device__ __forceinline__ float gm_acosf_t(float x) {
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    const bool small = ix < 0x3f000000u;                                     // |x| < 1/2
    const float ax = __builtin_fabsf(x);
    const float z = small ? x * x : (1.0f - ax) * 0.5f;                      // ((one + x) * 0.5 for x < 0: the same subtraction)
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = CORES ? gm_div_core(p, q) : p / q;
    const float s = CORES ? gm_sqrt_core(z) : __builtin_sqrtf(z);
    const float r_small = GM_ACOS_PIO2_HI - (x - (GM_ACOS_PIO2_LO - x * r));
    const float r_neg = GM_ACOS_PI - 2.0f * (s + (r * s - GM_ACOS_PIO2_LO));
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float num = z - df * df, den = s + df;
    const float c = CORES ? gm_div_core(num, den) : num / den;
    float r_pos = 2.0f * (df + (r * s + c));
    // (the streaming kernel evaluates several frames as interleaved straight-line chains: keep the compiler from sinking
    // a range's arithmetic into a branch — a wave always has lanes of every range)
    if (CORES) asm volatile("" : "+v"(r_pos));
    float res = small ? r_small : ((hx >> 31) ? r_neg : r_pos);
    res = ix <= 0x23000000u ? GM_ACOS_PIO2_HI + GM_ACOS_PIO2_LO : res;      // |x| <= 2^-57
    res = ix == 0x3f800000u ? ((hx >> 31) ? GM_ACOS_PI + 2.0f * GM_ACOS_PIO2_LO : 0.0f) : res;
    return ix > 0x3f800000u ? __builtin_nanf("") : res;                     // |x| > 1, NaN: (x - x) / (x - x)
}
__device__ __forceinline__ float gm_acosf(float x) { return gm_acosf_t<false>(x); }

// glibc sysdeps/ieee754/flt-32/s_cosf.c + s_sinf.c + sincosf.h (reduce_fast, sinf_poly) for t in [0, pi] or NaN: the
// argument in double, n = round(t * 2 / pi) by one multiplication and an integer shift (0, 1 or 2 here), the remainder
// t - n * (pi / 2), then the cosine or the sine polynomial in double and ONE rounding to float.  The second table of
// glibc (n & 2) holds the cosine coefficients negated: every operation is sign-symmetric, so that is the negated result.
__device__ __forceinline__ void gm_sincosf_0pi(float t, float &sn, float &cs) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16, s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (__float_as_uint(t) >> 20) & 0x7ffu;               // abstop12
    const bool first = top < 0x3f4u;                                        // abstop12(t) < abstop12(pi / 4): t < 0.75
    const bool tiny = top < 0x398u;                                         // t < 2^-12: cos = 1, sin = t
    const double x0 = (double)t;
    const int n = first ? 0 : (((int)(x0 * hpi_inv) + 0x800000) >> 24);
    const double x = first ? x0 : x0 - (double)n * hpi;
    const double x2 = x * x;
    // sinf_poly, odd n: the cosine
    const double x4 = x2 * x2, cc2 = c3 + x2 * c4, cc1 = c0 + x2 * c1, x6 = x4 * x2, cc = cc1 + x4 * c2;
    const float cosp = (float)(cc + x6 * cc2);
    // sinf_poly, even n: the sine
    const double x3 = x * x2, ss1 = s2 + x2 * s3, x7 = x3 * x2, ss = x + x3 * s1;
    const float sinp = (float)(ss + x7 * ss1);
    // cos: n = 0 cos(x), 1 -sin(x) (sign[1] = -1 on the odd polynomial), 2 -cos(x) (the negated table)
    cs = n == 1 ? -sinp : (n == 2 ? -cosp : cosp);
    // sin: n = 0 sin(x), 1 cos(x), 2 -sin(x)
    sn = n == 1 ? cosp : (n == 2 ? -sinp : sinp);
    cs = tiny ? 1.0f : cs;
    sn = tiny ? t : sn;
}
__device__ __forceinline__ float gm_cosf(float t) { float s, c; gm_sincosf_0pi(t, s, c); return c; }
__device__ __forceinline__ float gm_sinf_0pi(float t) { float s, c; gm_sincosf_0pi(t, s, c); return s; }

// P2 of the angle between the bond vector v and the membrane normal n (calc_sch, mod.rs:78-82).
//   n2 = |n|, n2sq = |n|^2, both precomputed on the host with nalgebra's f32 sequence.
//
// ACOS_COS = true restates the reference literally: c = clamp(v.n / (|v||n|)), angle = acos(c)
//   (0 if a norm is 0), S = 1.5 cos(angle)^2 - 0.5, with this file's own acos / cos kernels.
//
// ACOS_COS = false (library default, gorder_flags_t in gorder_hip.h) evaluates the same quantity from
//   the SQUARED cosine, q = (v.n)^2 / (|v|^2 |n|^2), S = 1.5 q - 0.5: no acos -> cos round trip, no
//   square root, one IEEE division.  Every operation is a correctly rounded f32 operation, restated
//   by the oracle's DIRECT mode.  Against the reference's libm pipeline 5.9 % of samples move by one
//   1e-6 tick (never more) and the mean moves by 2.6e-10 (tools/trig_fidelity.c).
// AXIS = 0/1/2: the normal is exactly the unit vector of that axis (the reference's default, z).  Then
// v.n = (vx*0 + vy*0) + vz*1 = v[AXIS] and |n|^2 = 1 for every FINITE v, and the multiplications can
// be dropped; a non-finite |v|^2 (the only case where 0 * v_k is not 0) sets `*nonfinite` and the
// caller re-evaluates that sample through the generic path.  AXIS = -1: generic normal.
template <bool ACOS_COS, int AXIS = -1>
__device__ __forceinline__ float gm_calc_sch(float vx, float vy, float vz, float nx, float ny, float nz,
                                             float n2, float n2sq, bool *nonfinite = nullptr) {
#ifdef GORDER_DEBUG_NOMATH   // timing experiment only: how fast does the data path alone stream?
    return (vx + vy) + vz;
#endif
    const float s2 = (vx * vx + vy * vy) + vz * vz;
    float prod;
    if (AXIS >= 0 && !ACOS_COS) {
        prod = AXIS == 0 ? vx : (AXIS == 1 ? vy : vz);
        if (nonfinite) *nonfinite = !(__builtin_fabsf(s2) <= 3.4028234663852886e38f);
        float q = (prod * prod) / s2;
        q = q > 1.0f ? 1.0f : q;
        q = (s2 == 0.0f) ? 1.0f : q;
        return (1.5f * q) - 0.5f;
    }
    prod = (vx * nx + vy * ny) + vz * nz;
    if (ACOS_COS) {
        const float n1 = __builtin_sqrtf(s2);
        float c = prod / (n1 * n2);
        c = c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c);     // NaN passes through, like f32::clamp
        float angle = gm_acosf(c);
        angle = (n1 == 0.0f || n2 == 0.0f) ? 0.0f : angle;  // nalgebra: angle = 0 if either norm is 0
        const float co = gm_cosf(angle);
        return (1.5f * co * co) - 0.5f;
    } else {
        float q = (prod * prod) / (s2 * n2sq);
        q = q > 1.0f ? 1.0f : q;                             // the clamp of the cosine; NaN passes through
        q = (s2 == 0.0f || n2sq == 0.0f) ? 1.0f : q;         // zero norm -> angle 0 -> cos^2 = 1
        return (1.5f * q) - 0.5f;
    }
}

// The streaming kernel's sample for a static normal along a coordinate axis: S = 1.5 vz^2 / |v|^2 - 0.5 — the same
// operations as gm_calc_sch<false, AXIS>, minus what cannot matter inside the guarded range:
//  * the quotient cannot exceed 1 (s2 = fl(fl(x^2 + y^2) + z^2) >= z^2 by monotonic rounding), so no clamp;
//  * |v|^2 outside [2^-40, 2^40] (this includes 0, inf and NaN) raises `rare` and the caller recomputes the sample
//    with the general routine;
//  * inside that range v_div_scale_f32 / v_div_fixup_f32 of the compiler's IEEE division are the identity except for
//    a numerator below 2^-103 (then the quotient is below 2^-63 and S is -0.5 whatever its last bit), so the
//    division is its Newton core alone: the same eight operations in the same order, three instructions less.
__device__ __forceinline__ float gm_div_core(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    const float e2 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e3, r, q);
}
// The same idea for the correctly rounded square root: for x in [2^-40, 2^40] the compiler's sequence needs neither
// its denormal pre-scaling nor the class fix-up — v_sqrt_f32 (1 ulp) and the two-sided correction are what remains.
__device__ __forceinline__ float gm_sqrt_core(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
}
template <int AXIS>
__device__ __forceinline__ float gm_sch_axis(float vx, float vy, float vz, bool &rare) {
    const float s2 = (vx * vx + vy * vy) + vz * vz;
    const float prod = AXIS == 0 ? vx : (AXIS == 1 ? vy : vz);
    rare = rare || !(s2 >= 0x1p-40f && s2 <= 0x1p+40f);
    return (1.5f * gm_div_core(prod * prod, s2)) - 0.5f;
}
// The reference's LITERAL evaluation (GORDER_FLAG_TRIG_ACOS_COS) for a static normal along a coordinate axis, with the
// same trims as gm_sch_axis: gm_calc_sch<true> computes c = clamp(v.n / (|v| |n|)), angle = acos(c), S = 1.5 cos^2 - 0.5.
// With n the unit vector of AXIS and a FINITE v:  v.n = (vx*0 + vy*0) + v_AXIS*1 = v_AXIS (a -0 instead of +0 gives the
// same angle, see gm_acosf_t's |x| <= 1/2 range), |n| = 1 so |v| |n| = |v|, and |v_AXIS| <= RN(sqrt(s2)) because
// s2 >= v_AXIS^2 and both roundings are monotonic — the quotient cannot leave [-1, 1], no clamp.  |v|^2 outside
// [2^-40, 2^40] (0, inf, NaN included) raises `rare` and the caller recomputes the sample with the general routine;
// inside, square root and division are their cores (|v| in [2^-20, 2^20], |v_AXIS| <= |v|; a numerator so small that
// v_div_fixup would matter gives a quotient below 2^-63, and acos of that is pi/2 whatever its last bit).
template <int AXIS>
__device__ __forceinline__ float gm_sch_axis_acos(float vx, float vy, float vz, bool &rare) {
    const float s2 = (vx * vx + vy * vy) + vz * vz;
    const float prod = AXIS == 0 ? vx : (AXIS == 1 ? vy : vz);
    rare = rare || !(s2 >= 0x1p-40f && s2 <= 0x1p+40f);
    const float c = gm_div_core(prod, gm_sqrt_core(s2));
    const float co = gm_cosf(gm_acosf_t<true>(c));
    return (1.5f * co * co) - 0.5f;
}
// gm_tick for a sample that is known not to be NaN (the caller's rare path takes those)
__device__ __forceinline__ int gm_tick_finite(float s) {
    const double t = (double)s * 1000000.0;
    return (int)(t + __builtin_copysign(0.5, t));
}

// round(f64(S) * 1e6) as i64 (order.rs:21-26) — S is in [-0.5, 1] or NaN here, so the tick fits 32 bits.
// f64::round is half-away-from-zero; NaN -> 0 (Rust `as i64`).
// t = S * 1e6 is exact in f64 (24-bit x 20-bit significands), has <= 38 significant bits and
// |t| <= 1e6, hence t + copysign(0.5, t) is exact whenever |t| >= 0.5 and stays inside (-1, 1)
// otherwise: truncating it toward zero IS round-half-away-from-zero.
__device__ __forceinline__ int gm_tick(float s) {
    const double t = (double)s * 1000000.0;
    const double r = t + __builtin_copysign(0.5, t);
    return (t != t) ? 0 : (int)r;
}
