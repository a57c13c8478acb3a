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

// ---- trig kernels: acos on [-1,1], cos on [0,pi]; < 1 ulp (acos 0.78, cos 0.6 away from pi/2).
// Coefficients from tools/derive_trig_coeffs.py (Remez fit, rounded to f32).
#define GM_PIO2_HI 0x1.921fb6p+0f
#define GM_PIO2_LO (-0x1.777a5cp-25f)
#define GM_PI_HI 0x1.921fb6p+1f
#define GM_PI_LO (-0x1.777a5cp-24f)
#define GM_PIO4 0x1.921fb6p-1f
#define GM_3PIO4 0x1.2d97c8p+1f

__device__ __forceinline__ float gm_asin_r(float z) {
    float p = 0x1.15e1a4p-5f;
    p = __builtin_fmaf(p, z, 0x1.169f76p-6f);
    p = __builtin_fmaf(p, z, 0x1.fe10bap-6f);
    p = __builtin_fmaf(p, z, 0x1.6d55e6p-5f);
    p = __builtin_fmaf(p, z, 0x1.333448p-4f);
    p = __builtin_fmaf(p, z, 0x1.555554p-3f);
    return p;
}

__device__ __forceinline__ float gm_div_core(float n, float d);
__device__ __forceinline__ float gm_sqrt_core(float x);

// Branch-free (a wave almost always holds lanes of every range); the arithmetic of each range is
// exactly the sequence restated in oracle/gorder_oracle.c (gorder_oracle_mirror_acosf).
// CORES: the square root and the division of the |x| > 1/2 range by their Newton cores (gm_sqrt_core / gm_div_core
// below).  Same bits: there z = (1 - |x|) / 2 is 0 or lies in [2^-25, 1/4) — inside the cores' guarded range, and
// gm_sqrt_core(0) = 0 —, the divisor s + s in [2^-12, 1], and the numerator z - s^2 is 0 or at least 2^-74 in
// magnitude (a multiple of ulp(s)^2); for |x| <= 1/2 both results are computed and discarded.
template <bool CORES = false>
__device__ __forceinline__ float gm_acosf_t(float x) {
    const float ax = __builtin_fabsf(x);
    // both argument reductions share the polynomial: z = x^2 (|x| <= 1/2) or (1-|x|)/2
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float r = z * gm_asin_r(z);
    const float r_small = GM_PIO2_HI - (x - (GM_PIO2_LO - x * r));
    const float s = CORES ? gm_sqrt_core(z) : __builtin_sqrtf(z);
    const float quo = CORES ? gm_div_core(__builtin_fmaf(-s, s, z), s + s) : __builtin_fmaf(-s, s, z) / (s + s);
    const float c = (s > 0.0f) ? quo : 0.0f;
    const float w = __builtin_fmaf(s, r, c);
    const float r_pos = 2.0f * (s + w);
    const float r_neg = 2.0f * (GM_PIO2_HI - (s + (w - GM_PIO2_LO)));
    float r_large = x > 0.0f ? r_pos : r_neg;
    // (the streaming kernel evaluates several frames as interleaved straight-line chains: keep the compiler from
    // sinking this range's arithmetic into a branch on `small` — a wave always has lanes of both ranges)
    if (CORES) asm volatile("" : "+v"(r_large));
    float res = small ? r_small : r_large;
    return (ax <= 1.0f) ? res : __builtin_nanf("");
}
__device__ __forceinline__ float gm_acosf(float x) { return gm_acosf_t<false>(x); }

__device__ __forceinline__ float gm_kcos(float r) {
    const float z = r * r;
    const float zl = __builtin_fmaf(r, r, -z);
    const float c = __builtin_fmaf(__builtin_fmaf(0x1.9bd908p-16f, z, -0x1.6c12d4p-10f), z, 0x1.555554p-5f);
    const float hz = 0.5f * z;
    const float w = 1.0f - hz;
    return w + ((((1.0f - w) - hz) - 0.5f * zl) + z * (z * c));
}
__device__ __forceinline__ float gm_ksin(float r) {
    const float z = r * r;
    const float s = __builtin_fmaf(
        __builtin_fmaf(__builtin_fmaf(0x1.6dbf02p-19f, z, -0x1.a013acp-13f), z, 0x1.11110ep-7f), z,
        -0x1.555556p-3f);
    return __builtin_fmaf(r * z, s, r);
}
// t in [0, pi] (the range of acos) or NaN.  Both kernels are evaluated and selected.
__device__ __forceinline__ float gm_cosf(float t) {
    const bool lo = t < GM_PIO4;
    const bool mid = !lo && (t <= GM_3PIO4);
    const float rc = lo ? t : (GM_PI_HI - t) + GM_PI_LO;
    const float rs = (GM_PIO2_HI - t) + GM_PIO2_LO;
    const float kc = gm_kcos(rc);
    const float ks = gm_ksin(rs);
    return mid ? ks : (lo ? kc : -kc);   // t = NaN: rc = NaN -> NaN
}

// sin on [0, pi] from the same two kernels (the rotation angle of the unsaturated-CH hydrogen, pi - gamma / 2,
// uaorder.rs:1024-1045, lies in [pi/2, pi]); restated by oracle/gorder_oracle.c (gorder_oracle_mirror_sinf).
__device__ __forceinline__ float gm_sinf_0pi(float t) {
    const bool lo = t < GM_PIO4;
    const bool mid = !lo && (t <= GM_3PIO4);
    const float rs = lo ? t : (GM_PI_HI - t) + GM_PI_LO;
    const float rc = (t - GM_PIO2_HI) - GM_PIO2_LO;
    const float ks = gm_ksin(rs);
    const float kc = gm_kcos(rc);
    return mid ? kc : ks;                // t = NaN: NaN
}

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
// ---- the same literal evaluation for TWO frames at a time in packed registers -----------------------------------------
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 do two IEEE f32 operations per lane and instruction; lane 0 / lane 1 of a
// gm_f2 hold the sample in two consecutive frames.  Every component goes through exactly the operations of the scalar
// routines above (gm_sqrt_core, gm_div_core, gm_acosf_t<true>, gm_cosf) in the same order, so the bits are the same;
// compares, selects, the hardware sqrt / rcp seeds and the integer steps are done per component.  The literal mode is
// bound by VALU issue, not by memory, and about half of its instructions are multiplications, additions and fmas.
typedef float gm_f2 __attribute__((ext_vector_type(2)));
typedef int gm_i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gm_f2 gm2(float v) { return gm_f2{v, v}; }
__device__ __forceinline__ gm_f2 gm2_fma(gm_f2 a, gm_f2 b, gm_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ gm_f2 gm2_min_image_step(gm_f2 d, gm_f2 L, gm_i2 &slow) {
    const gm_f2 half = L / 2.0f;
    const gm_f2 r = __builtin_elementwise_abs(d) > half ? d - __builtin_elementwise_copysign(L, d) : d;
    slow |= __builtin_elementwise_abs(r) > half;
    return r;
}
__device__ __forceinline__ gm_f2 gm2_sqrt_core(gm_f2 x) {
    const gm_f2 s = gm_f2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const gm_i2 si = __builtin_bit_cast(gm_i2, s);
    const gm_f2 sm = __builtin_bit_cast(gm_f2, si - 1), sp = __builtin_bit_cast(gm_f2, si + 1);
    const gm_f2 rm = gm2_fma(-sm, s, x), rp = gm2_fma(-sp, s, x);
    gm_f2 r = rm <= gm2(0.0f) ? sm : s;
    r = rp > gm2(0.0f) ? sp : r;
    return r;
}
__device__ __forceinline__ gm_f2 gm2_div_core(gm_f2 n, gm_f2 d) {
    gm_f2 r = gm_f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const gm_f2 e = gm2_fma(-d, r, gm2(1.0f));
    r = gm2_fma(e, r, r);
    gm_f2 q = n * r;
    const gm_f2 e2 = gm2_fma(-d, q, n);
    q = gm2_fma(e2, r, q);
    const gm_f2 e3 = gm2_fma(-d, q, n);
    return gm2_fma(e3, r, q);
}
__device__ __forceinline__ gm_f2 gm2_asin_r(gm_f2 z) {
    gm_f2 p = gm2(0x1.15e1a4p-5f);
    p = gm2_fma(p, z, gm2(0x1.169f76p-6f));
    p = gm2_fma(p, z, gm2(0x1.fe10bap-6f));
    p = gm2_fma(p, z, gm2(0x1.6d55e6p-5f));
    p = gm2_fma(p, z, gm2(0x1.333448p-4f));
    p = gm2_fma(p, z, gm2(0x1.555554p-3f));
    return p;
}
__device__ __forceinline__ gm_f2 gm2_acosf_cores(gm_f2 x) {       // gm_acosf_t<true>, |x| <= 1 (the caller's quotient)
    const gm_f2 ax = __builtin_elementwise_abs(x);
    const gm_i2 small = ax <= gm2(0.5f);
    const gm_f2 z = small ? x * x : (gm2(1.0f) - ax) * 0.5f;
    const gm_f2 r = z * gm2_asin_r(z);
    const gm_f2 r_small = gm2(GM_PIO2_HI) - (x - (gm2(GM_PIO2_LO) - x * r));
    const gm_f2 s = gm2_sqrt_core(z);
    const gm_f2 quo = gm2_div_core(gm2_fma(-s, s, z), s + s);
    const gm_f2 c = s > gm2(0.0f) ? quo : gm2(0.0f);
    const gm_f2 w = gm2_fma(s, r, c);
    const gm_f2 r_pos = 2.0f * (s + w);
    const gm_f2 r_neg = 2.0f * (gm2(GM_PIO2_HI) - (s + (w - gm2(GM_PIO2_LO))));
    gm_f2 r_large = x > gm2(0.0f) ? r_pos : r_neg;
    asm volatile("" : "+v"(r_large));           // (see gm_acosf_t: keep this range's arithmetic out of a branch)
    return small ? r_small : r_large;
}
__device__ __forceinline__ gm_f2 gm2_kcos(gm_f2 r) {
    const gm_f2 z = r * r;
    const gm_f2 zl = gm2_fma(r, r, -z);
    const gm_f2 c = gm2_fma(gm2_fma(gm2(0x1.9bd908p-16f), z, gm2(-0x1.6c12d4p-10f)), z, gm2(0x1.555554p-5f));
    const gm_f2 hz = 0.5f * z;
    const gm_f2 w = gm2(1.0f) - hz;
    return w + ((((gm2(1.0f) - w) - hz) - 0.5f * zl) + z * (z * c));
}
__device__ __forceinline__ gm_f2 gm2_ksin(gm_f2 r) {
    const gm_f2 z = r * r;
    const gm_f2 s = gm2_fma(gm2_fma(gm2_fma(gm2(0x1.6dbf02p-19f), z, gm2(-0x1.a013acp-13f)), z, gm2(0x1.11110ep-7f)), z,
                            gm2(-0x1.555556p-3f));
    return gm2_fma(r * z, s, r);
}
__device__ __forceinline__ gm_f2 gm2_cosf(gm_f2 t) {              // gm_cosf
    const gm_i2 lo = t < gm2(GM_PIO4);
    const gm_i2 mid = ~lo & (t <= gm2(GM_3PIO4));
    const gm_f2 rc = lo ? t : (gm2(GM_PI_HI) - t) + gm2(GM_PI_LO);
    const gm_f2 rs = (gm2(GM_PIO2_HI) - t) + gm2(GM_PIO2_LO);
    const gm_f2 kc = gm2_kcos(rc);
    const gm_f2 ks = gm2_ksin(rs);
    return mid ? ks : (lo ? kc : -kc);
}
// gm_sch_axis_acos for the sample in two frames; `rare` gets the lanes' "recompute with the general routine" flags
template <int AXIS>
__device__ __forceinline__ gm_f2 gm2_sch_axis_acos(gm_f2 vx, gm_f2 vy, gm_f2 vz, gm_i2 &rare) {
    const gm_f2 s2 = (vx * vx + vy * vy) + vz * vz;
    const gm_f2 prod = AXIS == 0 ? vx : (AXIS == 1 ? vy : vz);
    rare |= ~((s2 >= gm2(0x1p-40f)) & (s2 <= gm2(0x1p+40f)));
    const gm_f2 c = gm2_div_core(prod, gm2_sqrt_core(s2));
    const gm_f2 co = gm2_cosf(gm2_acosf_cores(c));
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
