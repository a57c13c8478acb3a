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
__device__ __forceinline__ float gm_min_image(float dx, float L, int &bad) {
    const float half = L / 2.0f;
    int it = 0;
    while (dx > half) {
        dx -= L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return dx; }
    }
    it = 0;
    while (dx < -half) {
        dx += L;
        if (++it > GM_MI_MAX_ITER) { bad = 1; return dx; }
    }
    return dx;
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

__device__ __forceinline__ float gm_acosf(float x) {
    const float ax = __builtin_fabsf(x);
    if (!(ax <= 1.0f)) return __builtin_nanf("");
    // both argument reductions share the polynomial: z = x^2 (|x| <= 1/2) or (1-|x|)/2
    const bool small = ax <= 0.5f;
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float r = z * gm_asin_r(z);
    if (small) return GM_PIO2_HI - (x - (GM_PIO2_LO - x * r));
    const float s = __builtin_sqrtf(z);
    const float c = (s > 0.0f) ? __builtin_fmaf(-s, s, z) / (s + s) : 0.0f;
    float w = __builtin_fmaf(s, r, c);
    if (x > 0.0f) return 2.0f * (s + w);
    w = w - GM_PIO2_LO;
    return 2.0f * (GM_PIO2_HI - (s + w));
}

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
__device__ __forceinline__ float gm_cosf(float t) {
    if (t < GM_PIO4) return gm_kcos(t);
    if (t <= GM_3PIO4) return gm_ksin((GM_PIO2_HI - t) + GM_PIO2_LO);
    if (t != t) return t;
    return -gm_kcos((GM_PI_HI - t) + GM_PI_LO);
}

// nalgebra angle + calc_sch.  n2 = |normal| is precomputed on the host with the same f32 sequence.
__device__ __forceinline__ float gm_calc_sch(float vx, float vy, float vz, float nx, float ny, float nz,
                                             float n2) {
    const float prod = (vx * nx + vy * ny) + vz * nz;
    const float n1 = __builtin_sqrtf((vx * vx + vy * vy) + vz * vz);
    float angle = 0.0f;
    if (!(n1 == 0.0f || n2 == 0.0f)) {
        float c = prod / (n1 * n2);
        if (c < -1.0f) c = -1.0f;
        else if (c > 1.0f) c = 1.0f;
        angle = gm_acosf(c);
    }
    const float co = gm_cosf(angle);
    return (1.5f * co * co) - 0.5f;
}

// round(f64(S) * 1e6) as i64 — S is in [-0.5, 1] or NaN here, so the tick fits 32 bits.
// f64::round is half-away-from-zero; NaN -> 0 (Rust `as i64`).
__device__ __forceinline__ int gm_tick(float s) {
    const double t = (double)s * 1000000.0;
    const double r = __builtin_round(t);
    return (r != r) ? 0 : (int)r;
}
