/*
 * gorder_oracle.c — CPU restatement (plain C99 + pthreads) of the per-frame order-parameter path of
 * VachaLab/gorder v1.4.1.  Every function cites the reference file:line it follows.
 *
 * TEST INFRASTRUCTURE ONLY (see gorder_oracle.h).  Not part of the product path.
 *
 * Pinning status: see DESIGN.md §"Oracle".  The reference is Rust and cannot be built here (no
 * cargo/rustc, un-vendored crates), so this restatement is pinned by the reference's own known-
 * answer tests and golden output files (tests/golden/, tests/test_oracle_*.py).
 * Arithmetic that lives in third-party crates absent from /root/reference (groan_rs 0.11.2,
 * nalgebra 0.34.0, statistical 1.0.0) is restated from their documented behaviour; each such
 * function is marked [3rd-party].
 *
 * All geometry is f32 with NO fused multiply-add (build with -ffp-contract=off): rustc never
 * contracts a*b+c.  fmaf() appears only in MIRROR-mode trig, which restates the device kernels.
 */
#define _GNU_SOURCE
#include "gorder_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================================
 * scalar building blocks
 * ========================================================================================== */

#define MI_MAX_ITER 8 /* see GORDER_ERR_BOX_RANGE in gorder_hip.h */

/* [3rd-party] groan_rs minimum image of a 1-D displacement: shift by whole box lengths until the
 * value lies in [-L/2, L/2].  Used by Vector3D::vector_to (pbc.rs:378-385) and ::distance
 * (pbc.rs:354-356).  `*bad` is raised instead of spinning for ever on a degenerate input. */
static inline float min_image(float dx, float L, int *bad) {
    const float half = L / 2.0f;
    int it = 0;
    while (dx > half) {
        dx -= L;
        if (++it > MI_MAX_ITER) { *bad = 1; return dx; }
    }
    it = 0;
    while (dx < -half) {
        dx += L;
        if (++it > MI_MAX_ITER) { *bad = 1; return dx; }
    }
    return dx;
}

/* [3rd-party] groan_rs Vector3D::wrap: into [0, L] per dimension (pbc.rs:388-390). */
static inline float wrap1(float x, float L, int *bad) {
    int it = 0;
    while (x > L) {
        x -= L;
        if (++it > MI_MAX_ITER) { *bad = 1; return x; }
    }
    it = 0;
    while (x < 0.0f) {
        x += L;
        if (++it > MI_MAX_ITER) { *bad = 1; return x; }
    }
    return x;
}

/* PBC3D::vector_to (pbc.rs:378-385) / NoPBC::vector_to (pbc.rs:188-190) */
static inline int vector_to(const float *p1, const float *p2, const float *box, int pbc, float *out) {
    int bad = 0;
    for (int d = 0; d < 3; d++) {
        float v = p2[d] - p1[d];
        out[d] = pbc ? min_image(v, box[d], &bad) : v;
    }
    return bad;
}

int gorder_oracle_vector_to(const float p1[3], const float p2[3], const float box[3], int handle_pbc,
                            float out[3]) {
    return vector_to(p1, p2, box, handle_pbc, out);
}

/* [3rd-party] nalgebra dot / norm of a 3-vector: (a0*b0 + a1*b1) + a2*b2, sqrt of the same. */
static inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float norm3(const float *a) { return sqrtf(dot3(a, a)); }

/* ---- MIRROR-mode trig.  Rounds 1-3: a restatement of the device's own polynomial kernels.  Since round 4 the device
 * computes acos / cos / sin the way glibc does (gorder_amd/csrc/gm_math.h), and these are the same restatements of
 * glibc 2.28 - 2.40's algorithms — sysdeps/ieee754/flt-32/e_acosf.c (fdlibm), s_cosf.c / s_sinf.c / sincosf.h (2018:
 * reduce_fast + sinf_poly in double) — written from the published algorithms, on the domains the path can produce
 * (acos on [-1, 1], cos and sin on [0, pi]).  On a host whose libm IS such a glibc they equal acosf / cosf / sinf bit for
 * bit (tests/test_oracle_kat.py checks that over the whole domains), i.e. MIRROR = LIBM there; on a host with another
 * libm MIRROR still says what the DEVICE computes. */
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float gorder_oracle_mirror_acosf(float x) {
    static const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
        pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
        pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
        qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    float z, p, q, r, w, s, c, df;
    const int32_t hx = (int32_t)f_bits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;      /* |x| = 1 */
    if (ix > 0x3f800000) return (x - x) / (x - x);                          /* |x| > 1, NaN */
    if (ix < 0x3f000000) {                                                  /* |x| < 0.5 */
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;                     /* |x| <= 2^-57 */
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                                                           /* x < -0.5 */
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    z = (one - x) * 0.5f;                                                   /* x > 0.5 */
    s = sqrtf(z);
    df = bits_f(f_bits(s) & 0xfffff000u);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0f * (df + w);
}

/* sincosf.h: sinf_poly (even n: the sine, odd n: the cosine; `neg`: glibc's second table = negated cosine coefficients) */
static inline float m_sinf_poly(double x, double x2, int neg, int n) {
    static const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                        c4 = 0x1.99343027bf8c3p-16, s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7,
                        s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = s2 + x2 * s3, x7 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x7 * t1);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double x4 = x2 * x2, t2 = sg * c3 + x2 * (sg * c4), t1 = sg * c0 + x2 * (sg * c1), x6 = x4 * x2, c = t1 + x4 * (sg * c2);
    return (float)(c + x6 * t2);
}
static inline double m_reduce_fast(double x, int *np) {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * 0x1.921FB54442D18p0;
}
static inline uint32_t m_abstop12(float x) { return (f_bits(x) >> 20) & 0x7ffu; }
static const double m_sign[4] = {1.0, -1.0, -1.0, 1.0};
/* valid for t in [0, pi] (the range of acos; glibc's own range of this branch is |t| < 120); NaN propagates */
float gorder_oracle_mirror_cosf(float t) {
    double x = t;
    int n;
    if (t != t) return t;
    if (m_abstop12(t) < m_abstop12(0x1.921FB6p-1f)) {
        if (m_abstop12(t) < m_abstop12(0x1p-12f)) return 1.0f;
        return m_sinf_poly(x, x * x, 0, 1);
    }
    x = m_reduce_fast(x, &n);
    return m_sinf_poly(x * m_sign[n & 3], x * x, (n & 2) != 0, n ^ 1);
}
float gorder_oracle_mirror_sinf(float t) {
    double x = t;
    int n;
    if (t != t) return t;
    if (m_abstop12(t) < m_abstop12(0x1.921FB6p-1f)) {
        if (m_abstop12(t) < m_abstop12(0x1p-12f)) return t;
        return m_sinf_poly(x, x * x, 0, 0);
    }
    x = m_reduce_fast(x, &n);
    return m_sinf_poly(x * m_sign[n & 3], x * x, (n & 2) != 0, n);
}
/* batch forms for the tests: fn 0 acos, 1 cos, 2 sin of the floats with bit patterns first + i * stride; which = 0 the
 * restatement above, 1 the host's libm */
void gorder_oracle_trig_batch(int fn, int which, uint32_t first_bits, uint32_t stride, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; i++) {
        const float x = bits_f(first_bits + i * stride);
        out[i] = fn == 0 ? (which ? acosf(x) : gorder_oracle_mirror_acosf(x))
               : fn == 1 ? (which ? cosf(x) : gorder_oracle_mirror_cosf(x))
                         : (which ? sinf(x) : gorder_oracle_mirror_sinf(x));
    }
}

/* [3rd-party] nalgebra Matrix::angle, reached through groan_rs Vector3D::angle (mod.rs:79):
 * 0 if either norm is 0, else acos(clamp(a.b / (|a||b|), -1, 1)); clamp lets NaN through.
 * `*cosine` receives the clamped cosine (1 for the zero-norm case, = cos(0)). */
static inline float angle3(const float *a, const float *b, int trig, float *cosine) {
    const float prod = dot3(a, b);
    const float n1 = norm3(a), n2 = norm3(b);
    if (n1 == 0.0f || n2 == 0.0f) { if (cosine) *cosine = 1.0f; return 0.0f; }
    float c = prod / (n1 * n2);
    if (c < -1.0f) c = -1.0f;
    else if (c > 1.0f) c = 1.0f;
    if (cosine) *cosine = c;
    return trig == GORDER_ORACLE_TRIG_MIRROR ? gorder_oracle_mirror_acosf(c) : acosf(c);
}

/* calc_sch, src/analysis/mod.rs:78-82.
 * DIRECT restates the device library's default evaluation (gorder_amd/csrc/gm_math.h): P2 from the
 * squared cosine q = (v.n)^2 / (|v|^2 |n|^2), no acos -> cos round trip and no square root. */
static inline float calc_sch(const float *v, const float *n, int trig) {
    if (trig == GORDER_ORACLE_TRIG_DIRECT) {
        const float prod = dot3(v, n);
        const float s2 = dot3(v, v), n2sq = dot3(n, n);
        float q = (prod * prod) / (s2 * n2sq);
        if (q > 1.0f) q = 1.0f;
        if (s2 == 0.0f || n2sq == 0.0f) q = 1.0f;
        return (1.5f * q) - 0.5f;
    }
    const float angle = angle3(v, n, trig, NULL);
    const float co = trig == GORDER_ORACLE_TRIG_MIRROR ? gorder_oracle_mirror_cosf(angle) : cosf(angle);
    return (1.5f * co * co) - 0.5f;
}
float gorder_oracle_calc_sch(const float v[3], const float n[3], int trig_mode) {
    return calc_sch(v, n, trig_mode);
}

/* OrderValue::from(f32), order.rs:21-26: (value as f64 * 1e6).round() as i64.
 * f64::round = half away from zero; `as i64` saturates and maps NaN to 0. */
int64_t gorder_oracle_tick(float s) {
    const double t = round((double)s * 1000000.0);
    if (t != t) return 0;
    if (t >= 9223372036854775807.0) return INT64_MAX;
    if (t <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)t;
}

/* AnalysisOrder::calc_order (order.rs:101-107) with OrderValue / usize (order.rs:34-42, truncating
 * integer division) and f32::from(OrderValue) (order.rs:28-32). */
float gorder_oracle_calc_order(int64_t sum, uint64_t n, uint64_t min_samples) {
    if (n < min_samples || n == 0) return NAN;
    const int64_t q = sum / (int64_t)n;
    return (float)((double)q / 1000000.0);
}

/* ---- centre of geometry -------------------------------------------------------------------
 * [3rd-party] groan_rs System::group_get_center (pbc.rs:269-271) / iterator get_center
 * (pbc.rs:305-308): "Refined Bai-Breen" (CHANGELOG.md:46).  Restated as: wrap each position, map
 * to an angle per dimension, average cos/sin, take the circular mean; then refine with the plain
 * mean of the minimum-image displacements from that estimate and wrap.  NoPBC: arithmetic mean
 * (group_get_center_naive, pbc.rs:103).  Summation in atom-index order, f32.
 * Only the SIGN of (head - centre) along the normal feeds the results (leaflets.rs:725-731), so
 * last-bit differences from the crate cannot change a flag except for a lipid sitting within
 * ~1e-6 nm of the centre plane; pinned through the leaflet counts of aaorder.rs:268-350. */
/* the periodic image of x nearest to `ref`: x shifted by whole box lengths while |x - ref| > L/2 */
static inline float nearest_image(float x, float ref, float L, int *bad) {
    const float half = L / 2.0f;
    float d = x - ref;
    int it = 0;
    while (d > half) { d -= L; x -= L; if (++it > MI_MAX_ITER) { *bad = 1; return x; } }
    it = 0;
    while (d < -half) { d += L; x += L; if (++it > MI_MAX_ITER) { *bad = 1; return x; } }
    return x;
}

static int center_of(const float *xyz, const uint32_t *idx, uint32_t n, const float *box, int pbc,
                     float *out) {
    int bad = 0;
    if (n == 0) { out[0] = out[1] = out[2] = NAN; return 0; }
    if (!pbc) {
        float s[3] = {0, 0, 0};
        for (uint32_t i = 0; i < n; i++) {
            const float *p = xyz + 3 * (size_t)idx[i];
            s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
        }
        for (int d = 0; d < 3; d++) out[d] = s[d] / (float)n;
        return 0;
    }
    const float two_pi = 6.2831855f;
    float est[3];
    for (int d = 0; d < 3; d++) {
        const float scaling = two_pi / box[d];
        float sc = 0.0f, ss = 0.0f;
        for (uint32_t i = 0; i < n; i++) {
            float c = wrap1(xyz[3 * (size_t)idx[i] + d], box[d], &bad);
            const float theta = c * scaling;
            sc += cosf(theta);
            ss += sinf(theta);
        }
        const float th = atan2f(-ss, -sc) + 3.1415927f;
        est[d] = th / scaling;
    }
    /* refinement: every atom is moved to its image nearest to the estimate, then the plain centre of
     * those positions is taken (sum of ABSOLUTE positions, f32, atom order) and wrapped.  This form —
     * rather than est + mean(displacement), which differs in the last bit — is pinned by the golden
     * aa_order_sphere_dynamic.yaml (tests_aa.rs:3322-3346): in frame 49 one C217-H17R bond of POPE lies
     * 2 ulp from the surface of the sphere around the centre of residue 1, and only this form leaves it
     * outside as the reference does. */
    /* The image is written as the atom's own coordinate shifted by whole box lengths (x -+ L), not as
     * est + displacement: for a compact group both are the atom's coordinate itself, bit for bit, but this form
     * does not carry the last bits of the estimate into the result — the estimate only picks the image. */
    for (int d = 0; d < 3; d++) {
        float acc = 0.0f;
        for (uint32_t i = 0; i < n; i++) acc += nearest_image(xyz[3 * (size_t)idx[i] + d], est[d], box[d], &bad);
        out[d] = wrap1(acc / (float)n, box[d], &bad);
    }
    return bad;
}
int gorder_oracle_center(const float *xyz, const uint32_t *idx, uint32_t n, const float box[3],
                         int handle_pbc, float out[3]) {
    return center_of(xyz, idx, n, box, handle_pbc, out);
}

/* ---- united-atom hydrogen construction, uaorder.rs:947-1104 ------------------------------- */
static const float TETRAHEDRAL_ANGLE = 1.910633f;      /* uaorder.rs:35 */
static const float TETRAHEDRAL_ANGLE_HALF = 0.9553165f; /* uaorder.rs:37 */
static const float BOND_LENGTH = 0.109f;                /* uaorder.rs:39 */
static const float CH3_ANGLE = 2.0943952f;              /* uaorder.rs:41 */

static inline void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
/* [3rd-party] nalgebra Unit::new_normalize / groan_rs to_unit: component / norm */
static inline void unit3(const float *a, float *o) {
    const float n = norm3(a);
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}
/* [3rd-party] nalgebra Rotation3::from_axis_angle(unit axis, angle) applied to v
 * (groan_rs Vector3D::rotate = matrix * vector, column-accumulated). */
static void rotate_axis_angle_mode(const float *u, float angle, const float *v, float *o, int mirror) {
    if (angle == 0.0f) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; return; }
    const float ux = u[0], uy = u[1], uz = u[2];
    const float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    /* mirror: the device's own sin / cos kernels (angle in [0, pi]); else libm like Rust's f32::sin_cos */
    const float s = mirror ? gorder_oracle_mirror_sinf(angle) : sinf(angle);
    const float c = mirror ? gorder_oracle_mirror_cosf(angle) : cosf(angle);
    const float omc = 1.0f - c;
    const float m11 = sqx + (1.0f - sqx) * c;
    const float m12 = ux * uy * omc - uz * s;
    const float m13 = ux * uz * omc + uy * s;
    const float m21 = ux * uy * omc + uz * s;
    const float m22 = sqy + (1.0f - sqy) * c;
    const float m23 = uy * uz * omc - ux * s;
    const float m31 = ux * uz * omc - uy * s;
    const float m32 = uy * uz * omc + ux * s;
    const float m33 = sqz + (1.0f - sqz) * c;
    o[0] = (m11 * v[0] + m12 * v[1]) + m13 * v[2];
    o[1] = (m21 * v[0] + m22 * v[1]) + m23 * v[2];
    o[2] = (m31 * v[0] + m32 * v[1]) + m33 * v[2];
}
static void rotate_axis_angle(const float *u, float angle, const float *v, float *o) {
    rotate_axis_angle_mode(u, angle, v, o, 0);
}
/* [3rd-party] groan_rs Vector3D::shift(direction, distance): move along the normalised direction;
 * then PBCHandler::wrap (pbc.rs:388-390; no-op for NoPBC, pbc.rs:193) */
static int shift_wrap(const float *target, const float *dir, const float *box, int pbc, float *h) {
    float u[3];
    int bad = 0;
    unit3(dir, u);
    for (int d = 0; d < 3; d++) {
        h[d] = target[d] + u[d] * BOND_LENGTH;
        if (pbc) h[d] = wrap1(h[d], box[d], &bad);
    }
    return bad;
}

/* trig: GORDER_ORACLE_TRIG_LIBM = the reference's arithmetic (acosf / sinf / cosf of the host libm, like Rust's f32
 * methods); any other mode restates the DEVICE's kernels for the one data-dependent angle of the construction (the
 * unsaturated CH), so that the device's hydrogens can be compared bit for bit. */
static int predict_hydrogens_mode(uint32_t kind, const float pos[4][3], const float box[3], int pbc, int trig,
                                  float out[3][3]) {
    int bad = 0;
    if (kind == GORDER_UA_CH3) { /* uaorder.rs:947-981; indices helper1,target,helper2 */
        const float *h1 = pos[0], *t = pos[1], *h2 = pos[2];
        float th1[3], th2[3], axis[3], ua[3], hv1[3], nth1[3], hv[3];
        bad |= vector_to(t, h1, box, pbc, th1);
        bad |= vector_to(t, h2, box, pbc, th2);
        cross3(th2, th1, axis);
        unit3(axis, ua);
        rotate_axis_angle(ua, TETRAHEDRAL_ANGLE, th1, hv1);
        bad |= shift_wrap(t, hv1, box, pbc, out[0]);
        unit3(th1, nth1);
        rotate_axis_angle(nth1, CH3_ANGLE, hv1, hv);
        bad |= shift_wrap(t, hv, box, pbc, out[1]);
        rotate_axis_angle(nth1, -CH3_ANGLE, hv1, hv);
        bad |= shift_wrap(t, hv, box, pbc, out[2]);
        return bad ? -1 : 3;
    }
    if (kind == GORDER_UA_CH2) { /* uaorder.rs:985-1020 */
        const float *h1 = pos[0], *t = pos[1], *h2 = pos[2];
        float a[3], b[3], th1[3], th2[3], pn[3], diff[3], ra[3], rv[3], ura[3], hv[3];
        bad |= vector_to(t, h1, box, pbc, a);
        bad |= vector_to(t, h2, box, pbc, b);
        unit3(a, th1);
        unit3(b, th2);
        cross3(th2, th1, pn);
        for (int d = 0; d < 3; d++) diff[d] = th1[d] - th2[d];
        unit3(diff, ra);
        cross3(pn, ra, rv);
        unit3(ra, ura); /* Unit::new_normalize(*rot_axis) normalises the already-unit axis again */
        rotate_axis_angle(ura, TETRAHEDRAL_ANGLE_HALF, rv, hv);
        bad |= shift_wrap(t, hv, box, pbc, out[0]);
        rotate_axis_angle(ura, -TETRAHEDRAL_ANGLE_HALF, rv, hv);
        bad |= shift_wrap(t, hv, box, pbc, out[1]);
        return bad ? -1 : 2;
    }
    if (kind == GORDER_UA_CH1_UNSAT) { /* uaorder.rs:1024-1045 */
        const float *h1 = pos[0], *t = pos[1], *h2 = pos[2];
        float th1[3], th2[3], axis[3], ua[3], hv[3];
        bad |= vector_to(t, h1, box, pbc, th1);
        bad |= vector_to(t, h2, box, pbc, th2);
        const int mirror = trig != GORDER_ORACLE_TRIG_LIBM;
        const float gamma = angle3(th1, th2, mirror ? GORDER_ORACLE_TRIG_MIRROR : GORDER_ORACLE_TRIG_LIBM, NULL);
        cross3(th1, th2, axis);
        unit3(axis, ua);
        rotate_axis_angle_mode(ua, 3.14159265358979323846f - (gamma / 2.0f), th2, hv, mirror);
        bad |= shift_wrap(t, hv, box, pbc, out[0]);
        return bad ? -1 : 1;
    }
    if (kind == GORDER_UA_CH1_SAT) { /* uaorder.rs:1087-1104; indices h1,h2,h3,target */
        const float *t = pos[3];
        float a[3], th1[3], th2[3], th3[3], hv[3];
        bad |= vector_to(t, pos[0], box, pbc, a); unit3(a, th1);
        bad |= vector_to(t, pos[1], box, pbc, a); unit3(a, th2);
        bad |= vector_to(t, pos[2], box, pbc, a); unit3(a, th3);
        for (int d = 0; d < 3; d++) hv[d] = -((th1[d] + th2[d]) + th3[d]);
        bad |= shift_wrap(t, hv, box, pbc, out[0]);
        return bad ? -1 : 1;
    }
    return -2;
}
/* ---- GORDER_FLAG_UA_FAST_NORMALISE: restatement of the DEVICE's tolerance-bounded construction
 * (gorder_amd/csrc/kernels_extras.h: ua_fast_rsqrt, PbcFast, ua_carbon_fast), operation for operation — every step
 * there is an IEEE mul / add / fma, rint or floor, so the bits are reproducible here.  Not the reference's arithmetic:
 * the LIBM mode never takes this path; tools/ua_fast_fidelity.py measures one against the other. */
static inline float fast_rsqrt(float x) {
    union { float f; int32_t i; } u;
    u.f = x;
    u.i = 0x5f375a86 - (u.i >> 1);
    float y = u.f;
    const float hx = 0.5f * x;
    y = y * fmaf(-(hx * y), y, 1.5f);
    y = y * fmaf(-(hx * y), y, 1.5f);
    y = y * fmaf(-(hx * y), y, 1.5f);
    return y;
}
typedef struct { const float *box; float inv[3]; int pbc; int slow; int need_pos; float hu[3]; } fast_ctx;
/* Rotations by Rodrigues' formula with given sine and cosine (the constants of the construction are evaluated once on
 * the host with libm — sinf / cosf of the same literals — exactly as the device's ExtraArgs carries them), not by
 * nalgebra's rotation matrix.  v rotated about the unit axis u PERPENDICULAR to it: v c + (u x v) s */
static void rotate_perp(const float *u, float s, float c, const float *v, float *o) {
    float w[3];
    cross3(u, v, w);
    for (int d = 0; d < 3; d++) o[d] = fmaf(w[d], s, v[d] * c);
}
/* general Rodrigues: v c + (u x v) s + u (u.v)(1 - c) */
static void rotate_rod(const float *u, float s, float c, const float *v, float *o) {
    float w[3];
    cross3(u, v, w);
    const float k = fmaf(u[2], v[2], fmaf(u[1], v[1], u[0] * v[0])) * (1.0f - c);
    for (int d = 0; d < 3; d++) o[d] = fmaf(u[d], k, fmaf(w[d], s, v[d] * c));
}
static inline float fast_mi(fast_ctx *c, float d, int k) {
    if (!c->pbc) return d;
    const float q = rintf(d * c->inv[k]);
    if (fabsf(q) > 1.0f) c->slow = 1;
    return fmaf(-c->box[k], q, d);
}
static inline float fast_wr(fast_ctx *c, float x, int k) {
    if (!c->pbc) return x;
    const float q = floorf(x * c->inv[k]);
    if (fabsf(q) > 1.0f) c->slow = 1;
    return fmaf(-c->box[k], q, x);
}
static inline void fast_to(fast_ctx *c, const float *p1, const float *p2, float *o) {
    for (int d = 0; d < 3; d++) o[d] = fast_mi(c, p2[d] - p1[d], d);
}
static inline float fast_rnorm(fast_ctx *c, const float *a) {
    const float s2 = fmaf(a[2], a[2], fmaf(a[1], a[1], a[0] * a[0]));
    if (!(s2 >= 0x1p-40f && s2 <= 0x1p+40f)) c->slow = 1;
    return fast_rsqrt(s2);
}
static inline void fast_unit(fast_ctx *c, const float *a, float *o) {
    const float r = fast_rnorm(c, a);
    o[0] = a[0] * r; o[1] = a[1] * r; o[2] = a[2] * r;
}
static inline void fast_shift_wrap(fast_ctx *c, const float *t, const float *dir, float *h) {
    const float r = fast_rnorm(c, dir) * BOND_LENGTH;
    for (int d = 0; d < 3; d++) {
        c->hu[d] = fmaf(dir[d], r, t[d]);
        /* the wrapped hydrogen is only consumed through the bond position (ordermaps, geometry selection): a run
         * without them neither wraps nor looks at how many box lengths a wrap would take */
        h[d] = c->need_pos ? fast_wr(c, c->hu[d], d) : c->hu[d];
    }
}
/* -> number of hydrogens, their positions and the vectors target -> H; *slow = 1: the device re-evaluates this carbon
 * with the reference's literal loops (predict_hydrogens_mode + vector_to), and so must the caller */
static int predict_hydrogens_fast(uint32_t kind, const float pos[4][3], const float box[3], int pbc, int need_pos,
                                  float out[3][3], float vec[3][3], int *slow) {
    fast_ctx c;
    c.box = box; c.pbc = pbc; c.slow = 0; c.need_pos = need_pos;
    for (int d = 0; d < 3; d++) c.inv[d] = pbc ? 1.0f / box[d] : 1.0f;      /* k_inv_box */
    const float *t = kind == GORDER_UA_CH1_SAT ? pos[3] : pos[1];
    int nh;
    float hus[3][3];
    if (kind == GORDER_UA_CH3) {
        float th1[3], th2[3], axis[3], ua[3], hv1[3], n1[3], hv[3];
        fast_to(&c, t, pos[0], th1);
        fast_to(&c, t, pos[2], th2);
        cross3(th2, th1, axis);
        fast_unit(&c, axis, ua);
        rotate_perp(ua, sinf(TETRAHEDRAL_ANGLE), cosf(TETRAHEDRAL_ANGLE), th1, hv1);   /* ua _|_ th1 */
        fast_shift_wrap(&c, t, hv1, out[0]); memcpy(hus[0], c.hu, sizeof(c.hu));
        fast_unit(&c, th1, n1);
        rotate_rod(n1, sinf(CH3_ANGLE), cosf(CH3_ANGLE), hv1, hv);
        fast_shift_wrap(&c, t, hv, out[1]); memcpy(hus[1], c.hu, sizeof(c.hu));
        rotate_rod(n1, -sinf(CH3_ANGLE), cosf(CH3_ANGLE), hv1, hv);
        fast_shift_wrap(&c, t, hv, out[2]); memcpy(hus[2], c.hu, sizeof(c.hu));
        nh = 3;
    } else if (kind == GORDER_UA_CH2) {
        float a[3], b[3], th1[3], th2[3], pn[3], diff[3], ra[3], rv[3], hv[3];
        fast_to(&c, t, pos[0], a);
        fast_to(&c, t, pos[2], b);
        fast_unit(&c, a, th1);
        fast_unit(&c, b, th2);
        cross3(th2, th1, pn);
        for (int d = 0; d < 3; d++) diff[d] = th1[d] - th2[d];
        fast_unit(&c, diff, ra);                 /* (the unit axis is not normalised a second time) */
        cross3(pn, ra, rv);
        rotate_perp(ra, sinf(TETRAHEDRAL_ANGLE_HALF), cosf(TETRAHEDRAL_ANGLE_HALF), rv, hv);   /* ra _|_ rv */
        fast_shift_wrap(&c, t, hv, out[0]); memcpy(hus[0], c.hu, sizeof(c.hu));
        rotate_perp(ra, -sinf(TETRAHEDRAL_ANGLE_HALF), cosf(TETRAHEDRAL_ANGLE_HALF), rv, hv);
        fast_shift_wrap(&c, t, hv, out[1]); memcpy(hus[1], c.hu, sizeof(c.hu));
        nh = 2;
    } else if (kind == GORDER_UA_CH1_UNSAT) {
        float th1[3], th2[3], axis[3], ua[3], hv[3];
        fast_to(&c, t, pos[0], th1);
        fast_to(&c, t, pos[2], th2);
        const float gamma = angle3(th1, th2, GORDER_ORACLE_TRIG_MIRROR, NULL);
        const float ang = 3.14159265358979323846f - (gamma / 2.0f);
        cross3(th1, th2, axis);
        fast_unit(&c, axis, ua);
        if (ang == 0.0f) { hv[0] = th2[0]; hv[1] = th2[1]; hv[2] = th2[2]; }
        else rotate_perp(ua, gorder_oracle_mirror_sinf(ang), gorder_oracle_mirror_cosf(ang), th2, hv);   /* ua _|_ th2 */
        fast_shift_wrap(&c, t, hv, out[0]); memcpy(hus[0], c.hu, sizeof(c.hu));
        nh = 1;
    } else if (kind == GORDER_UA_CH1_SAT) {
        float a[3], th1[3], th2[3], th3[3], hv[3];
        fast_to(&c, t, pos[0], a); fast_unit(&c, a, th1);
        fast_to(&c, t, pos[1], a); fast_unit(&c, a, th2);
        fast_to(&c, t, pos[2], a); fast_unit(&c, a, th3);
        for (int d = 0; d < 3; d++) hv[d] = -((th1[d] + th2[d]) + th3[d]);
        fast_shift_wrap(&c, t, hv, out[0]); memcpy(hus[0], c.hu, sizeof(c.hu));
        nh = 1;
    } else {
        return -2;
    }
    for (int k = 0; k < nh; k++) {
        for (int d = 0; d < 3; d++) vec[k][d] = hus[k][d] - t[d];      /* from the hydrogen before it is wrapped */
    }
    *slow = c.slow;
    return nh;
}
int gorder_oracle_predict_hydrogens_fast(uint32_t kind, const float pos[4][3], const float box[3], int pbc,
                                         float out[3][3], float vec[3][3], int *slow) {
    return predict_hydrogens_fast(kind, pos, box, pbc, 1, out, vec, slow);
}

int gorder_oracle_predict_hydrogens(uint32_t kind, const float pos[4][3], const float box[3],
                                    int pbc, float out[3][3]) {
    return predict_hydrogens_mode(kind, pos, box, pbc, GORDER_ORACLE_TRIG_LIBM, out);
}

/* ---- geometry selection ----------------------------------------------------------------------
 * Shape construction: CuboidAnalysis / CylinderAnalysis / SphereAnalysis::construct_shape
 * (geometry.rs:328-357, 422-451, 507-514) with get_infinite_span (pbc.rs:236-240, 392-396): an
 * unbounded extent anchors at 0 (PBC) or f32::MIN (NoPBC).  [3rd-party] groan_rs Rectangular /
 * Cylinder / Sphere ::inside: offsets from the anchor are wrapped into [0, L] (PBC) and must not exceed
 * the extent; radial tests use the minimum-image distance, strict `<`. */
/* ---- dynamic membrane normals ------------------------------------------------------------------
 * DynamicMembraneNormal::calculate_normal (normal.rs:160-199): cloud of "NormalHeads" atoms inside
 * Sphere(reference, radius) — PBC3D::get_heads_cloud (pbc.rs:321-350: minimum-image distance, points made
 * whole as reference + vector_to) or NoPBC::get_heads_cloud (pbc.rs:142-161: naive) — then
 * membrane_normal_from_cloud (normal.rs:421-458): demean, SVD, last right singular vector, to_unit;
 * fewer than 3 points -> DynamicNormalError::NotEnoughPoints.
 *
 * nalgebra's f32 SVD cannot be restated bit for bit (its source is not under /root/reference).  The
 * direction it returns is the eigenvector of the smallest eigenvalue of sum (p - c)(p - c)^T; here that
 * 3x3 matrix is accumulated in f64 from the f32 minimum-image vectors d = vector_to(reference, p)
 * (translation does not change it) and diagonalised by cyclic Jacobi rotations in f64.  The sign of a
 * singular vector is arbitrary and P2 does not depend on it; convention here: last non-zero component
 * positive.  Pinned only through the reference's 4-decimal goldens (the "dynamic" files of tests/golden/expected). */
static void sym3_smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22,
                                      double out[3]) {
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-34 * dia)) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                const int r = 3 - p - q;                       /* the third index */
                const double apq = a[p][q], arp = a[r][p], arq = a[r][q];
                a[p][p] = a[p][p] - t * apq;
                a[q][q] = a[q][q] + t * apq;
                a[p][q] = a[q][p] = 0.0;
                a[r][p] = a[p][r] = c * arp - sn * arq;
                a[r][q] = a[q][r] = sn * arp + c * arq;
                for (int k = 0; k < 3; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    out[0] = v[0][m]; out[1] = v[1][m]; out[2] = v[2][m];
    const double lead = out[2] != 0.0 ? out[2] : (out[1] != 0.0 ? out[1] : out[0]);
    if (lead < 0.0) { out[0] = -out[0]; out[1] = -out[1]; out[2] = -out[2]; }
}

/* normal[4] = (nx, ny, nz, number of cloud points); NaN normal when fewer than 3 points */
static int dynamic_normal(const float *xyz, const uint32_t *cloud, uint32_t n_cloud, uint32_t head, float radius,
                          const float *box, int pbc, float *normal, uint64_t *err_index) {
    const float *ref = xyz + 3 * (size_t)head;
    if (ref[0] != ref[0]) { *err_index = head; return GORDER_ERR_UNDEFINED_POSITION; }
    int bad = 0;
    uint32_t n = 0;
    double s[3] = {0, 0, 0}, ss[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < n_cloud; i++) {
        const float *p = xyz + 3 * (size_t)cloud[i];
        float d[3];
        bad |= vector_to(ref, p, box, pbc, d);
        if (!(sqrtf(dot3(d, d)) < radius)) continue;           /* Sphere::inside: distance < radius */
        if (p[0] != p[0]) { *err_index = cloud[i]; return GORDER_ERR_UNDEFINED_POSITION; }
        n++;
        for (int k = 0; k < 3; k++) s[k] += (double)d[k];
        ss[0] += (double)d[0] * d[0]; ss[1] += (double)d[0] * d[1]; ss[2] += (double)d[0] * d[2];
        ss[3] += (double)d[1] * d[1]; ss[4] += (double)d[1] * d[2]; ss[5] += (double)d[2] * d[2];
    }
    normal[3] = (float)n;
    if (n < 3) { normal[0] = normal[1] = normal[2] = NAN; return bad ? GORDER_ERR_BOX_RANGE : GORDER_OK; }
    const double inv = 1.0 / (double)n;
    double e[3];
    sym3_smallest_eigenvector(ss[0] - s[0] * s[0] * inv, ss[1] - s[0] * s[1] * inv, ss[2] - s[0] * s[2] * inv,
                              ss[3] - s[1] * s[1] * inv, ss[4] - s[1] * s[2] * inv, ss[5] - s[2] * s[2] * inv, e);
    const float f[3] = {(float)e[0], (float)e[1], (float)e[2]};
    const float len = norm3(f);                                /* Vector3D::to_unit */
    normal[0] = f[0] / len; normal[1] = f[1] / len; normal[2] = f[2] / len;
    return bad ? GORDER_ERR_BOX_RANGE : GORDER_OK;
}
int gorder_oracle_dynamic_normal(const float *xyz, const uint32_t *cloud, uint32_t n_cloud, uint32_t head,
                                 float radius, const float box[3], int handle_pbc, float normal[4]) {
    uint64_t e = 0;
    return dynamic_normal(xyz, cloud, n_cloud, head, radius, box, handle_pbc, normal, &e);
}

typedef struct { float pos[3]; float size[3]; float radius, height; } o_shape;

static int make_shape(const gorder_geometry_t *g, const float *ref, const float *box, int pbc, o_shape *sh) {
    int bad = 0;
    const float unbounded_anchor = pbc ? 0.0f : -3.40282347e+38f;
    memset(sh, 0, sizeof(*sh));
    for (int d = 0; d < 3; d++) sh->pos[d] = ref[d];
    if (g->kind == GORDER_GEOM_CUBOID) {
        const float *dims[3] = {g->xdim, g->ydim, g->zdim};
        for (int d = 0; d < 3; d++) {
            if (dims[d][0] == -INFINITY && dims[d][1] == INFINITY) { sh->pos[d] = unbounded_anchor; sh->size[d] = INFINITY; }
            else { sh->pos[d] = ref[d] + dims[d][0]; sh->size[d] = dims[d][1] - dims[d][0]; }
        }
    } else if (g->kind == GORDER_GEOM_CYLINDER) {
        const int o = (int)g->orientation;
        sh->radius = g->radius;
        if (g->span[0] == -INFINITY && g->span[1] == INFINITY) { sh->pos[o] = unbounded_anchor; sh->height = INFINITY; }
        else { sh->pos[o] = ref[o] + g->span[0]; sh->height = g->span[1] - g->span[0]; }
    } else {
        sh->radius = g->radius;
    }
    if (pbc) for (int d = 0; d < 3; d++) sh->pos[d] = wrap1(sh->pos[d], box[d], &bad);
    return bad;
}

static inline int inside_shape(const gorder_geometry_t *g, const o_shape *sh, const float *p, const float *box,
                               int pbc, int *bad) {
    int in = 1;
    if (g->kind == GORDER_GEOM_CUBOID) {
        for (int d = 0; d < 3; d++) {
            float e = p[d] - sh->pos[d];
            if (pbc) { e = wrap1(e, box[d], bad); in = in && (e <= sh->size[d]); }
            else in = in && (e >= 0.0f) && (e <= sh->size[d]);
        }
    } else if (g->kind == GORDER_GEOM_CYLINDER) {
        const int o = (int)g->orientation, a = (o + 1) % 3, b = (o + 2) % 3;
        float da = p[a] - sh->pos[a], db = p[b] - sh->pos[b], e = p[o] - sh->pos[o];
        if (pbc) { da = min_image(da, box[a], bad); db = min_image(db, box[b], bad); e = wrap1(e, box[o], bad); }
        in = (sqrtf(da * da + db * db) < sh->radius) && (pbc ? 1 : (e >= 0.0f)) && (e <= sh->height);
    } else if (g->kind == GORDER_GEOM_SPHERE) {
        float d[3];
        for (int k = 0; k < 3; k++) { d[k] = p[k] - sh->pos[k]; if (pbc) d[k] = min_image(d[k], box[k], bad); }
        in = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) < sh->radius;
    }
    return in ^ (g->invert ? 1 : 0);
}

/* ---- timewise statistics ------------------------------------------------------------------ */
/* TimeWiseData::estimate_error, timewise.rs:191-231; [3rd-party] statistical::standard_deviation
 * = sqrt(sum((x-mean)^2)/(n-1)) in f32. */
float gorder_oracle_estimate_error(const int64_t *sums, const uint64_t *counts, uint64_t n_frames,
                                   uint64_t n_blocks) {
    if (n_frames == 0 || n_blocks < 2) return NAN;
    const uint64_t block_size = n_frames / n_blocks;
    if (block_size == 0) return NAN;
    int64_t *bs = (int64_t *)calloc(n_blocks, sizeof(int64_t));
    uint64_t *bn = (uint64_t *)calloc(n_blocks, sizeof(uint64_t));
    for (uint64_t i = 0; i < n_frames; i++) {
        const uint64_t b = i / block_size;
        if (b < n_blocks) { bs[b] += sums[i]; bn[b] += counts[i]; }
    }
    float *o = (float *)malloc(n_blocks * sizeof(float));
    float result;
    int nan = 0;
    for (uint64_t b = 0; b < n_blocks; b++) {
        if (bn[b] == 0) { nan = 1; break; }
        o[b] = (float)((double)(bs[b] / (int64_t)bn[b]) / 1000000.0);
    }
    if (nan) result = NAN;
    else {
        float mean = 0.0f;
        for (uint64_t b = 0; b < n_blocks; b++) mean += o[b];
        mean = mean / (float)n_blocks;
        float var = 0.0f;
        for (uint64_t b = 0; b < n_blocks; b++) { const float d = mean - o[b]; var += d * d; }
        result = sqrtf(var / (float)(n_blocks - 1));
    }
    free(bs); free(bn); free(o);
    return result;
}
/* TimeWiseData::prefix_average, timewise.rs:259-274 */
void gorder_oracle_prefix_average(const int64_t *sums, const uint64_t *counts, uint64_t n_frames,
                                  float *out) {
    int64_t s = 0;
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_frames; i++) {
        s += sums[i];
        n += counts[i];
        out[i] = n == 0 ? NAN : (float)((double)(s / (int64_t)n) / 1000000.0);
    }
}

/* ============================================================================================
 * the analysis state ("SystemTopology", topology/mod.rs:34-65)
 * ========================================================================================== */

typedef struct {
    uint32_t n_molecules, n_bond_types, n_ua_atoms, n_methyls;
    uint32_t *bonds;      /* [n_bond_types][n_molecules][2] */
    uint32_t *ua_kind;    /* [n_ua_atoms] */
    uint32_t **ua_idx;    /* [n_ua_atoms] -> [n_molecules][4] */
    uint32_t *ua_slot0;   /* first accumulator slot of each ua atom */
    uint32_t *heads;      /* [n_molecules] or NULL */
    uint32_t *methyls;    /* [n_molecules][n_methyls] or NULL */
    uint32_t *normal_heads; /* [n_molecules] or NULL (dynamic membrane normals) */
    uint32_t slot0;       /* first accumulator slot of this molecule type */
    uint32_t mol0;        /* index of first molecule in the global molecule numbering */
} o_moltype;

typedef struct {
    int64_t *sums;     /* [3][n_acc] */
    uint64_t *counts;  /* [3][n_acc] */
    int64_t *map_sums; /* [3][n_acc][ntiles] or NULL */
    uint64_t *map_counts;
} o_acc;

struct gorder_oracle_handle {
    uint32_t n_atoms, n_mt, n_acc, n_mol_total;
    o_moltype *mt;
    int pbc, trig, n_threads, timewise;
    int ua_fast;              /* GORDER_FLAG_UA_FAST_NORMALISE and a non-libm mode: restate the device's fast construction */
    float normal[3];
    gorder_leaflets_t lf;
    uint32_t *membrane;
    gorder_ordermap_t om;
    uint32_t nx, ny;
    gorder_geometry_t geom;
    uint32_t *geom_group;
    gorder_dynamic_normal_t dyn;
    uint32_t *dyn_cloud;
    float *last_normals;      /* [n_mol_total][4] (nx, ny, nz, n_points) of the last analysed frame */
    float *manual_normals;    /* [manual_frames][n_mol_total][3] for the next submit (ManualMembraneNormal, normal.rs:266-300) */
    uint32_t manual_frames;
    o_acc acc;
    /* leaflets: flags of the most recent assignment (AssignedLeaflets::local, leaflets.rs:1371-1380) */
    uint8_t *flags;
    float *flag_dist;
    int have_flags;
    uint64_t flags_frame;
    /* timewise: grows by one row [3][n_acc] per analysed frame */
    int64_t *tw_sums;
    uint64_t *tw_counts;
    uint64_t tw_cap;
    uint64_t n_frames;
    uint64_t err_index;
};

static void acc_alloc(const gorder_oracle_handle *h, o_acc *a) {
    a->sums = (int64_t *)calloc(3 * (size_t)h->n_acc, sizeof(int64_t));
    a->counts = (uint64_t *)calloc(3 * (size_t)h->n_acc, sizeof(uint64_t));
    a->map_sums = NULL;
    a->map_counts = NULL;
    if (h->om.enabled) {
        const size_t n = 3 * (size_t)h->n_acc * h->nx * h->ny;
        a->map_sums = (int64_t *)calloc(n, sizeof(int64_t));
        a->map_counts = (uint64_t *)calloc(n, sizeof(uint64_t));
    }
}
static void acc_free(o_acc *a) {
    free(a->sums); free(a->counts); free(a->map_sums); free(a->map_counts);
}

/* [3rd-party] groan_rs GridMap::new: tiles centred at span_min + k*bin, n = round(span/bin) + 1
 * (confirmed from fixture data). */
static uint32_t gridmap_n(float lo, float hi, float bin) {
    const float n = roundf((hi - lo) / bin);
    return n < 0.0f ? 0u : (uint32_t)n + 1u;
}
/* [3rd-party] GridMap::get_mut_at: nearest tile centre; None outside the grid. */
static inline int gridmap_index(float x, float lo, float bin, uint32_t n) {
    const float k = roundf((x - lo) / bin);
    if (!(k >= 0.0f) || !(k < (float)n)) return -1;
    return (int)k;
}

/* The device's tile index under GORDER_FLAG_UA_FAST_NORMALISE (gorder_amd/csrc/kernels_bonds.h: grid_index_fast), operation
 * for operation: one fma with the reciprocal of the bin and a floor.  Not the reference's arithmetic. */
static inline int gridmap_index_fast(float x, float lo, float inv_bin, uint32_t n) {
    const float k = floorf(fmaf(x - lo, inv_bin, 0.5f));
    if (!(k >= 0.0f) || !(k < (float)n)) return -1;
    return (int)k;
}

uint32_t gorder_oracle_n_accumulators(const gorder_oracle_handle *h) { return h->n_acc; }
uint32_t gorder_oracle_ordermap_dims(const gorder_oracle_handle *h, uint32_t *nx, uint32_t *ny) {
    if (nx) *nx = h->nx;
    if (ny) *ny = h->ny;
    return h->om.enabled ? h->nx * h->ny : 0;
}
uint64_t gorder_oracle_last_error_index(const gorder_oracle_handle *h) { return h->err_index; }

static void *dup_mem(const void *p, size_t n) {
    if (!p || n == 0) return NULL;
    void *q = malloc(n);
    memcpy(q, p, n);
    return q;
}

#define gorder_oracle_destroy_partial gorder_oracle_destroy
int gorder_oracle_create(const gorder_tables_t *t, int trig_mode, int n_threads,
                         gorder_oracle_handle **out) {
    if (!t || !out || t->n_atoms == 0) return GORDER_ERR_INVALID_ARGUMENT;
    gorder_oracle_handle *h = (gorder_oracle_handle *)calloc(1, sizeof(*h));
    h->n_atoms = t->n_atoms;
    h->n_mt = t->n_molecule_types;
    h->pbc = t->handle_pbc != 0;
    h->trig = trig_mode;
    /* the reference-faithful mode ignores the flag: it IS the reference's arithmetic */
    h->ua_fast = (t->flags & GORDER_FLAG_UA_FAST_NORMALISE) && trig_mode != GORDER_ORACLE_TRIG_LIBM;
    h->n_threads = n_threads < 1 ? 1 : n_threads;
    h->timewise = t->timewise != 0;
    memcpy(h->normal, t->normal, sizeof(h->normal));
    h->lf = t->leaflets;
    h->membrane = (uint32_t *)dup_mem(t->leaflets.membrane, sizeof(uint32_t) * t->leaflets.n_membrane);
    h->lf.membrane = h->membrane;
    h->geom = t->geometry;
    h->geom_group = (uint32_t *)dup_mem(t->geometry.group, sizeof(uint32_t) * t->geometry.n_group);
    h->geom.group = h->geom_group;
    if (h->geom.kind != GORDER_GEOM_NONE && h->pbc && h->geom.reference == GORDER_GEOMREF_POINT &&
        !(h->geom.structure_box[0] > 0.0f && h->geom.structure_box[1] > 0.0f && h->geom.structure_box[2] > 0.0f)) {
        free(h->geom_group); free(h->membrane); free(h);
        return GORDER_ERR_INVALID_ARGUMENT;
    }
    if (h->geom.kind != GORDER_GEOM_NONE && h->geom.reference == GORDER_GEOMREF_BOX_CENTER && !h->pbc) {
        free(h->geom_group); free(h->membrane); free(h);
        return GORDER_ERR_INVALID_ARGUMENT;   /* NoPBC::get_box_center panics, pbc.rs:243-245 */
    }
    h->dyn = t->dynamic_normal;
    h->dyn_cloud = (uint32_t *)dup_mem(t->dynamic_normal.cloud, sizeof(uint32_t) * t->dynamic_normal.n_cloud);
    h->dyn.cloud = h->dyn_cloud;
    h->om = t->ordermap;
    if (h->om.enabled) {
        h->nx = gridmap_n(h->om.span_x[0], h->om.span_x[1], h->om.bin[0]);
        h->ny = gridmap_n(h->om.span_y[0], h->om.span_y[1], h->om.bin[1]);
    }
    h->mt = (o_moltype *)calloc(h->n_mt ? h->n_mt : 1, sizeof(o_moltype));
    uint32_t slot = 0, mol = 0;
    for (uint32_t m = 0; m < h->n_mt; m++) {
        const gorder_moltype_t *s = &t->molecule_types[m];
        o_moltype *d = &h->mt[m];
        d->n_molecules = s->n_molecules;
        d->n_bond_types = s->n_bond_types;
        d->n_ua_atoms = s->n_ua_atoms;
        d->n_methyls = s->n_methyls;
        d->slot0 = slot;
        d->mol0 = mol;
        d->bonds = (uint32_t *)dup_mem(s->bonds, sizeof(uint32_t) * 2 * (size_t)s->n_bond_types * s->n_molecules);
        slot += s->n_bond_types;
        if (s->n_ua_atoms) {
            d->ua_kind = (uint32_t *)calloc(s->n_ua_atoms, sizeof(uint32_t));
            d->ua_idx = (uint32_t **)calloc(s->n_ua_atoms, sizeof(uint32_t *));
            d->ua_slot0 = (uint32_t *)calloc(s->n_ua_atoms, sizeof(uint32_t));
            for (uint32_t a = 0; a < s->n_ua_atoms; a++) {
                const uint32_t k = s->ua_atoms[a].kind;
                d->ua_kind[a] = k;
                d->ua_idx[a] = (uint32_t *)dup_mem(s->ua_atoms[a].indices, sizeof(uint32_t) * 4 * (size_t)s->n_molecules);
                d->ua_slot0[a] = slot;
                slot += (k == GORDER_UA_CH3) ? 3 : (k == GORDER_UA_CH2) ? 2 : 1;
            }
        }
        d->heads = (uint32_t *)dup_mem(s->heads, sizeof(uint32_t) * s->n_molecules);
        d->methyls = (uint32_t *)dup_mem(s->methyls, sizeof(uint32_t) * (size_t)s->n_molecules * s->n_methyls);
        d->normal_heads = (uint32_t *)dup_mem(s->normal_heads, sizeof(uint32_t) * s->n_molecules);
        if (h->dyn.enabled && !d->normal_heads) { gorder_oracle_destroy_partial(h); return GORDER_ERR_INVALID_ARGUMENT; }
        mol += s->n_molecules;
    }
    h->n_acc = slot;
    h->n_mol_total = mol;
    acc_alloc(h, &h->acc);
    h->flags = (uint8_t *)calloc(mol ? mol : 1, 1);
    h->flag_dist = (float *)calloc(mol ? mol : 1, sizeof(float));
    h->last_normals = (float *)calloc(4 * (size_t)(mol ? mol : 1), sizeof(float));
    *out = h;
    return GORDER_OK;
}

void gorder_oracle_destroy(gorder_oracle_handle *h) {
    if (!h) return;
    for (uint32_t m = 0; m < h->n_mt; m++) {
        o_moltype *d = &h->mt[m];
        free(d->bonds);
        for (uint32_t a = 0; a < d->n_ua_atoms; a++) free(d->ua_idx[a]);
        free(d->ua_kind); free(d->ua_idx); free(d->ua_slot0); free(d->heads); free(d->methyls); free(d->normal_heads);
    }
    free(h->mt); free(h->membrane); free(h->geom_group); free(h->dyn_cloud); free(h->last_normals); free(h->manual_normals); acc_free(&h->acc);
    free(h->flags); free(h->flag_dist); free(h->tw_sums); free(h->tw_counts);
    free(h);
}

/* check_box, common.rs:186-198.  A frame's box arrives as a 3x3 matrix (XTC layout). */
static int check_box(const float *b9, float *box3) {
    int all_nan = 1;
    for (int i = 0; i < 9; i++) if (b9[i] == b9[i]) all_nan = 0;
    if (all_nan) return GORDER_ERR_UNDEFINED_BOX;
    if (b9[1] != 0.0f || b9[2] != 0.0f || b9[3] != 0.0f || b9[5] != 0.0f || b9[6] != 0.0f || b9[7] != 0.0f)
        return GORDER_ERR_NOT_ORTHOGONAL_BOX;
    box3[0] = b9[0]; box3[1] = b9[4]; box3[2] = b9[8];
    if (box3[0] == 0.0f && box3[1] == 0.0f && box3[2] == 0.0f) return GORDER_ERR_ZERO_BOX;
    if (!(box3[0] > 0.0f) || !(box3[1] > 0.0f) || !(box3[2] > 0.0f)) return GORDER_ERR_BOX_RANGE;
    return GORDER_OK;
}

/* should_assign, leaflets.rs:435-441 */
static inline int should_assign(uint32_t frequency, uint64_t frame) {
    return frequency == 0 ? frame == 0 : (frame % frequency) == 0;
}

/* ---- leaflet assignment of one frame (molecule.rs:61-70 + leaflets.rs:444-498, 1406-1435) ---- */
typedef struct { uint32_t *cell_start; uint32_t *cell_atoms; uint32_t ncx, ncy; } o_grid;

static int assign_leaflets(gorder_oracle_handle *h, const float *xyz, const float *box, uint8_t *flags,
                           float *dist) {
    const uint32_t dim = h->lf.normal_dim;
    int bad = 0;
    if (h->lf.method == GORDER_LEAFLETS_GLOBAL) {
        /* SystemLeafletClassification::run, leaflets.rs:186-197 */
        float center[3];
        bad |= center_of(xyz, h->membrane, h->lf.n_membrane, box, h->pbc, center);
        if (center[0] != center[0] || center[1] != center[1] || center[2] != center[2])
            return GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER;
        for (uint32_t m = 0; m < h->n_mt; m++) {
            const o_moltype *mt = &h->mt[m];
            for (uint32_t i = 0; i < mt->n_molecules; i++) {
                /* common_identify_leaflet, leaflets.rs:711-732 */
                const float hp = xyz[3 * (size_t)mt->heads[i] + dim];
                const float d = h->pbc ? min_image(hp - center[dim], box[dim], &bad) : hp - center[dim];
                dist[mt->mol0 + i] = d;
                flags[mt->mol0 + i] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (h->lf.flip ? 1 : 0));
            }
        }
    } else if (h->lf.method == GORDER_LEAFLETS_LOCAL) {
        /* PBC3D::calc_local_membrane_centers, pbc.rs:273-318 (NoPBC: pbc.rs:107-139).
         * Every membrane atom whose in-plane (PBC) distance from the head is < radius belongs to the
         * cylinder; the CellGrid of the reference only prunes the search.  Brute force over a 2-D
         * bucket grid here; members are visited in atom-index order. */
        const int a = (dim + 1) % 3, b = (dim + 2) % 3;
        const float r = h->lf.radius;
        uint32_t *members = (uint32_t *)malloc(sizeof(uint32_t) * (h->lf.n_membrane ? h->lf.n_membrane : 1));
        for (uint32_t m = 0; m < h->n_mt; m++) {
            const o_moltype *mt = &h->mt[m];
            for (uint32_t i = 0; i < mt->n_molecules; i++) {
                const float *hp = xyz + 3 * (size_t)mt->heads[i];
                uint32_t nm = 0;
                for (uint32_t k = 0; k < h->lf.n_membrane; k++) {
                    const float *p = xyz + 3 * (size_t)h->membrane[k];
                    float da = p[a] - hp[a], db = p[b] - hp[b];
                    if (h->pbc) { da = min_image(da, box[a], &bad); db = min_image(db, box[b], &bad); }
                    if (sqrtf(da * da + db * db) < r) members[nm++] = h->membrane[k];
                }
                float center[3];
                bad |= center_of(xyz, members, nm, box, h->pbc, center);
                if (nm == 0 || center[0] != center[0] || center[1] != center[1] || center[2] != center[2]) {
                    h->err_index = mt->heads[i];
                    free(members);
                    return GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER;
                }
                const float d = h->pbc ? min_image(hp[dim] - center[dim], box[dim], &bad) : hp[dim] - center[dim];
                dist[mt->mol0 + i] = d;
                flags[mt->mol0 + i] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (h->lf.flip ? 1 : 0));
            }
        }
        free(members);
    } else if (h->lf.method == GORDER_LEAFLETS_INDIVIDUAL) {
        /* IndividualClassification::identify_leaflet, leaflets.rs:777-801 */
        for (uint32_t m = 0; m < h->n_mt; m++) {
            const o_moltype *mt = &h->mt[m];
            for (uint32_t i = 0; i < mt->n_molecules; i++) {
                const float hp = xyz[3 * (size_t)mt->heads[i] + dim];
                float total = 0.0f;
                for (uint32_t k = 0; k < mt->n_methyls; k++) {
                    const float mp = xyz[3 * (size_t)mt->methyls[(size_t)i * mt->n_methyls + k] + dim];
                    total += h->pbc ? min_image(hp - mp, box[dim], &bad) : hp - mp;
                }
                dist[mt->mol0 + i] = total;
                flags[mt->mol0 + i] = (uint8_t)((total >= 0.0f ? 0 : 1) ^ (h->lf.flip ? 1 : 0));
            }
        }
    }
    return bad ? GORDER_ERR_BOX_RANGE : GORDER_OK;
}

/* ---- one sample: BondLike::add_order, bond.rs:184-215 (+ Map::add_order, ordermap.rs:100-113) */
static inline void add_order(const gorder_oracle_handle *h, o_acc *a, int64_t *tw_s, uint64_t *tw_n,
                             uint32_t slot, float sch, const float *pos, int leaflet /* -1 none */, int fast_bin) {
    const int64_t tick = gorder_oracle_tick(sch);
    const uint32_t n_acc = h->n_acc;
    int tile = -1;
    if (h->om.enabled) {
        float x, y;
        switch (h->om.plane) { /* Plane::projection2plane, input/ordermap.rs:44-50 */
            case 0: x = pos[0]; y = pos[1]; break;
            case 1: x = pos[0]; y = pos[2]; break;
            default: x = pos[2]; y = pos[1]; break;
        }
        const int ix = fast_bin ? gridmap_index_fast(x, h->om.span_x[0], 1.0f / h->om.bin[0], h->nx)
                                : gridmap_index(x, h->om.span_x[0], h->om.bin[0], h->nx);
        const int iy = fast_bin ? gridmap_index_fast(y, h->om.span_y[0], 1.0f / h->om.bin[1], h->ny)
                                : gridmap_index(y, h->om.span_y[0], h->om.bin[1], h->ny);
        if (ix >= 0 && iy >= 0) tile = ix * (int)h->ny + iy;
    }
    for (int pass = 0; pass < 2; pass++) {
        const int which = pass == 0 ? 0 : (leaflet < 0 ? -1 : 1 + leaflet);
        if (which < 0) break;
        const size_t k = (size_t)which * n_acc + slot;
        a->sums[k] += tick;
        a->counts[k] += 1;
        if (tw_s) { tw_s[k] += tick; tw_n[k] += 1; }
        if (tile >= 0) {
            const size_t t = k * ((size_t)h->nx * h->ny) + (size_t)tile;
            a->map_sums[t] += tick;
            a->map_counts[t] += 1;
        }
    }
}

/* ---- one frame: analyze_frame (common.rs:201-235) -> MoleculeTypes::analyze_frame
 * (molecule.rs:54-95) -> BondType::analyze_frame (bond.rs:396-446) / UAAtom::analyze_frame
 * (uaorder.rs:400-437).  `flags` = leaflet assignment that applies to this frame. */
static int analyze_frame_orders(const gorder_oracle_handle *h, o_acc *a, int64_t *tw_s, uint64_t *tw_n,
                                const float *xyz, const float *box, const uint8_t *flags,
                                uint64_t *err_index, float *normals_out /* [n_mol_total][4] scratch */,
                                const float *manual /* [n_mol_total][3] normals of this frame or NULL */) {
    int bad = 0;
    const int lf = h->lf.method != GORDER_LEAFLETS_NONE;
    const int geom = h->geom.kind != GORDER_GEOM_NONE;
    o_shape shape;
    if (geom) {   /* GeometrySelection::init_reference, geometry.rs:192-210 */
        float ref[3] = {h->geom.point[0], h->geom.point[1], h->geom.point[2]};
        const float *shape_box = box;
        if (h->geom.reference == GORDER_GEOMREF_BOX_CENTER) { ref[0] = box[0] / 2.0f; ref[1] = box[1] / 2.0f; ref[2] = box[2] / 2.0f; }
        else if (h->geom.reference == GORDER_GEOMREF_GROUP) bad |= center_of(xyz, h->geom_group, h->geom.n_group, box, h->pbc, ref);
        else shape_box = h->geom.structure_box;   /* fixed point: built once, with the structure's box (:194) */
        bad |= make_shape(&h->geom, ref, shape_box, h->pbc, &shape);
    }
    /* dynamic normals: the reference computes a molecule's normal lazily once per frame (OnceCell,
     * normal.rs:145-158); computing all of them up front gives the same values — only the
     * NotEnoughPoints error must wait until a sample of that molecule is really accumulated */
    float *dyn = NULL;
    if (manual) {             /* ManualMembraneNormal::get_normal(frame, molecule): the vector the host supplied */
        dyn = normals_out;
        for (uint32_t i = 0; i < h->n_mol_total; i++) {
            dyn[4 * (size_t)i + 0] = manual[3 * (size_t)i + 0];
            dyn[4 * (size_t)i + 1] = manual[3 * (size_t)i + 1];
            dyn[4 * (size_t)i + 2] = manual[3 * (size_t)i + 2];
            dyn[4 * (size_t)i + 3] = 3.0f;
        }
    } else if (h->dyn.enabled) {
        dyn = normals_out;
        for (uint32_t m = 0; m < h->n_mt; m++)
            for (uint32_t i = 0; i < h->mt[m].n_molecules; i++) {
                const int st = dynamic_normal(xyz, h->dyn_cloud, h->dyn.n_cloud, h->mt[m].normal_heads[i], h->dyn.radius,
                                              box, h->pbc, dyn + 4 * (size_t)(h->mt[m].mol0 + i), err_index);
                if (st == GORDER_ERR_UNDEFINED_POSITION) return st;
                if (st != GORDER_OK) bad = 1;
            }
    }
    for (uint32_t m = 0; m < h->n_mt; m++) {
        const o_moltype *mt = &h->mt[m];
        for (uint32_t bt = 0; bt < mt->n_bond_types; bt++) {
            const uint32_t *bonds = mt->bonds + 2 * (size_t)bt * mt->n_molecules;
            for (uint32_t i = 0; i < mt->n_molecules; i++) {
                const float *p1 = xyz + 3 * (size_t)bonds[2 * i];
                const float *p2 = xyz + 3 * (size_t)bonds[2 * i + 1];
                if (p1[0] != p1[0]) { *err_index = bonds[2 * i]; return GORDER_ERR_UNDEFINED_POSITION; }
                if (p2[0] != p2[0]) { *err_index = bonds[2 * i + 1]; return GORDER_ERR_UNDEFINED_POSITION; }
                float v[3], mid[3];
                bad |= vector_to(p1, p2, box, h->pbc, v);
                for (int d = 0; d < 3; d++) mid[d] = p1[d] + v[d] / 2.0f; /* bond.rs:422 */
                if (geom && !inside_shape(&h->geom, &shape, mid, box, h->pbc, &bad)) continue; /* bond.rs:424-426 */
                const float *normal = h->normal;
                if (dyn) {                                       /* bond.rs:429-431 */
                    normal = dyn + 4 * (size_t)(mt->mol0 + i);
                    if (normal[3] < 3.0f) { *err_index = (uint64_t)normal[3]; return GORDER_ERR_DYNAMIC_NORMAL; }
                }
                const float sch = calc_sch(v, normal, h->trig);
                add_order(h, a, tw_s, tw_n, mt->slot0 + bt, sch, mid, lf ? flags[mt->mol0 + i] : -1, 0);
            }
        }
        for (uint32_t ua = 0; ua < mt->n_ua_atoms; ua++) {
            const uint32_t kind = mt->ua_kind[ua];
            const int ti = kind == GORDER_UA_CH1_SAT ? 3 : 1;
            const int nidx = kind == GORDER_UA_CH1_SAT ? 4 : 3;
            for (uint32_t i = 0; i < mt->n_molecules; i++) {
                const uint32_t *ix = mt->ua_idx[ua] + 4 * (size_t)i;
                const float *normal = h->normal;
                if (dyn) {   /* uaorder.rs:412-413: fetched for every molecule, before the geometry test */
                    normal = dyn + 4 * (size_t)(mt->mol0 + i);
                    if (normal[3] < 3.0f) { *err_index = (uint64_t)normal[3]; return GORDER_ERR_DYNAMIC_NORMAL; }
                }
                float pos[4][3] = {{0}};
                for (int k = 0; k < nidx; k++) {
                    const float *p = xyz + 3 * (size_t)ix[k];
                    if (p[0] != p[0]) { *err_index = ix[k]; return GORDER_ERR_UNDEFINED_POSITION; }
                    pos[k][0] = p[0]; pos[k][1] = p[1]; pos[k][2] = p[2];
                }
                float hy[3][3], hv[3][3];
                int nh = -1, fast_ok = 0;
                if (h->ua_fast) {   /* the device's fast construction; a carbon it flags takes the literal path below */
                    int slow = 0;
                    nh = predict_hydrogens_fast(kind, pos, box, h->pbc, h->om.enabled || geom, hy, hv, &slow);
                    fast_ok = nh > 0 && !slow;
                }
                if (!fast_ok) nh = predict_hydrogens_mode(kind, pos, box, h->pbc, h->trig, hy);
                if (nh < 0) { bad = 1; continue; }
                for (int k = 0; k < nh; k++) {
                    /* UAAtom::calculate_sch, uaorder.rs:375-397: vec = target->H, pos = H + vec/2 (sic) */
                    float v[3], bp[3];
                    if (fast_ok) { v[0] = hv[k][0]; v[1] = hv[k][1]; v[2] = hv[k][2]; }
                    else bad |= vector_to(pos[ti], hy[k], box, h->pbc, v);
                    for (int d = 0; d < 3; d++) bp[d] = hy[k][d] + v[d] / 2.0f;
                    if (geom && !inside_shape(&h->geom, &shape, bp, box, h->pbc, &bad)) continue; /* uaorder.rs:388-390 */
                    const float sch = calc_sch(v, normal, h->trig);
                    add_order(h, a, tw_s, tw_n, mt->ua_slot0[ua] + (uint32_t)k, sch, bp,
                              lf ? flags[mt->mol0 + i] : -1, fast_ok);
                }
            }
        }
    }
    return bad ? GORDER_ERR_BOX_RANGE : GORDER_OK;
}

/* ---- fidelity of the fast united-atom construction (tools/ua_fast_fidelity.py): every virtual C-H sample of the given
 * frames evaluated twice — the reference's arithmetic (literal construction + libm acosf / cosf, the LIBM mode) and the
 * device's GORDER_FLAG_UA_FAST_NORMALISE arithmetic (fast construction + the squared cosine) — and compared one by one.
 *   hist[k]        samples whose tick differs by k (k = 15: by 15 or more)
 *   out[0] samples, out[1] carbons the fast path hands to the literal loops, out[2] samples whose bond position falls
 *   into another ordermap tile (0 without an ordermap), out[3] sum of signed tick differences + 2^62 (bias of the mean),
 *   out[4] samples that differ between the plain default (literal construction + squared cosine) and libm — the
 *   "5.9 %" of the default path, for comparison.  Static normal only. */
int gorder_oracle_ua_fast_fidelity(const gorder_oracle_handle *h, const float *xyz, const float *box9, uint32_t n_frames,
                                   uint64_t hist[16], uint64_t out[5]) {
    if (!h || !xyz || (!box9 && h->pbc)) return GORDER_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < 16; k++) hist[k] = 0;
    out[0] = out[1] = out[2] = out[4] = 0;
    int64_t bias = 0;
    for (uint32_t f = 0; f < n_frames; f++) {
        const float *x = xyz + 3 * (size_t)h->n_atoms * f;
        float box[3] = {0, 0, 0};
        if (h->pbc) { box[0] = box9[9 * (size_t)f]; box[1] = box9[9 * (size_t)f + 4]; box[2] = box9[9 * (size_t)f + 8]; }
        for (uint32_t m = 0; m < h->n_mt; m++) {
            const o_moltype *mt = &h->mt[m];
            for (uint32_t ua = 0; ua < mt->n_ua_atoms; ua++) {
                const uint32_t kind = mt->ua_kind[ua];
                const int ti = kind == GORDER_UA_CH1_SAT ? 3 : 1;
                const int nidx = kind == GORDER_UA_CH1_SAT ? 4 : 3;
                for (uint32_t i = 0; i < mt->n_molecules; i++) {
                    const uint32_t *ix = mt->ua_idx[ua] + 4 * (size_t)i;
                    float pos[4][3] = {{0}};
                    for (int k = 0; k < nidx; k++)
                        for (int d = 0; d < 3; d++) pos[k][d] = x[3 * (size_t)ix[k] + d];
                    float hy_ref[3][3], hy_def[3][3], hy_fast[3][3], v_fast[3][3];
                    int slow = 0;
                    const int nh = predict_hydrogens_mode(kind, pos, box, h->pbc, GORDER_ORACLE_TRIG_LIBM, hy_ref);
                    const int nh_d = predict_hydrogens_mode(kind, pos, box, h->pbc, GORDER_ORACLE_TRIG_DIRECT, hy_def);
                    const int nh_f = predict_hydrogens_fast(kind, pos, box, h->pbc, 1, hy_fast, v_fast, &slow);
                    if (nh < 0 || nh_f != nh || nh_d != nh) continue;
                    if (slow) out[1]++;
                    for (int k = 0; k < nh; k++) {
                        float v[3], vd[3], vf[3], bp[3], bpf[3];
                        vector_to(pos[ti], hy_ref[k], box, h->pbc, v);
                        vector_to(pos[ti], hy_def[k], box, h->pbc, vd);
                        if (slow) { vf[0] = vd[0]; vf[1] = vd[1]; vf[2] = vd[2]; }
                        else { vf[0] = v_fast[k][0]; vf[1] = v_fast[k][1]; vf[2] = v_fast[k][2]; }
                        const float *hf = slow ? hy_def[k] : hy_fast[k];
                        for (int d = 0; d < 3; d++) { bp[d] = hy_ref[k][d] + v[d] / 2.0f; bpf[d] = hf[d] + vf[d] / 2.0f; }
                        const int64_t t_ref = gorder_oracle_tick(calc_sch(v, h->normal, GORDER_ORACLE_TRIG_LIBM));
                        const int64_t t_def = gorder_oracle_tick(calc_sch(vd, h->normal, GORDER_ORACLE_TRIG_DIRECT));
                        const int64_t t_fast = gorder_oracle_tick(calc_sch(vf, h->normal, GORDER_ORACLE_TRIG_DIRECT));
                        int64_t dlt = t_fast - t_ref;
                        bias += dlt;
                        if (dlt < 0) dlt = -dlt;
                        hist[dlt > 15 ? 15 : dlt]++;
                        if (t_def != t_ref) out[4]++;
                        out[0]++;
                        if (h->om.enabled) {
                            int tile[2];
                            for (int w = 0; w < 2; w++) {
                                const float *p = w ? bpf : bp;
                                float px, py;
                                switch (h->om.plane) {
                                    case 0: px = p[0]; py = p[1]; break;
                                    case 1: px = p[0]; py = p[2]; break;
                                    default: px = p[2]; py = p[1]; break;
                                }
                                /* (w = 1: the fast construction's position AND its tile index) */
                                const int gx = w ? gridmap_index_fast(px, h->om.span_x[0], 1.0f / h->om.bin[0], h->nx)
                                                 : gridmap_index(px, h->om.span_x[0], h->om.bin[0], h->nx);
                                const int gy = w ? gridmap_index_fast(py, h->om.span_y[0], 1.0f / h->om.bin[1], h->ny)
                                                 : gridmap_index(py, h->om.span_y[0], h->om.bin[1], h->ny);
                                tile[w] = (gx >= 0 && gy >= 0) ? gx * (int)h->ny + gy : -1;
                            }
                            if (tile[0] != tile[1]) out[2]++;
                        }
                    }
                }
            }
        }
    }
    out[3] = (uint64_t)(bias + ((int64_t)1 << 62));
    return GORDER_OK;
}

/* ---- batch driver: restates groan_rs traj_iter_map_reduce as used at common.rs:283-339:
 * thread t analyses frames t, t+n, t+2n, ... with its own accumulator clone
 * (ParallelTrajData::initialize, topology/mod.rs:274-277), the clones are folded in thread order
 * (SystemTopology::reduce, topology/mod.rs:257-272).  Leaflet flags of assignment frames are
 * resolved first, which is what the reference's shared map + spin-wait (leaflets.rs:1523-1577)
 * achieves. */
typedef struct {
    gorder_oracle_handle *h;
    const float *xyz, *box;
    const uint8_t *frame_flags; /* [n_frames][n_mol_total] or NULL */
    uint32_t n_frames, tid, nthr;
    uint32_t passes;               /* gorder_oracle_submit_passes: walk the thread's frames this many times (>= 1) */
    o_acc acc;
    int64_t *tw_s; uint64_t *tw_n; /* batch timewise rows (shared, disjoint rows per thread) */
    const float *manual;           /* manual normals of the batch or NULL */
    int status; uint64_t err_index; uint32_t err_frame;
    int range_bad;   /* GORDER_ERR_BOX_RANGE met (the library's own range check): reported only if nothing else fails */
} o_job;

static void *worker(void *arg) {
    o_job *j = (o_job *)arg;
    gorder_oracle_handle *h = j->h;
    const size_t row = 3 * (size_t)h->n_acc;
    const int use_normals = h->dyn.enabled || j->manual;
    float *normals = use_normals ? (float *)malloc(4 * sizeof(float) * (size_t)(h->n_mol_total ? h->n_mol_total : 1)) : NULL;
    for (uint32_t pass = 0; pass < j->passes && j->status == GORDER_OK; pass++)
    for (uint32_t f = j->tid; f < j->n_frames; f += j->nthr) {
        float box3[3] = {0, 0, 0};
        if (h->pbc) {
            const int st = check_box(j->box + 9 * (size_t)f, box3);
            if (st == GORDER_ERR_BOX_RANGE) { j->range_bad = 1; continue; }   /* a frame the reference could not walk */
            if (st != GORDER_OK) { j->status = st; j->err_frame = f; break; }
        }
        const int st = analyze_frame_orders(
            h, &j->acc, j->tw_s ? j->tw_s + row * f : NULL, j->tw_n ? j->tw_n + row * f : NULL,
            j->xyz + 3 * (size_t)h->n_atoms * f, box3,
            j->frame_flags ? j->frame_flags + (size_t)h->n_mol_total * f : NULL, &j->err_index, normals,
            j->manual ? j->manual + 3 * (size_t)h->n_mol_total * f : NULL);
        if (st == GORDER_ERR_BOX_RANGE) j->range_bad = 1;
        else if (st != GORDER_OK) { j->status = st; j->err_frame = f; break; }
        if (normals && f + 1 == j->n_frames) memcpy(h->last_normals, normals, 4 * sizeof(float) * (size_t)h->n_mol_total);
    }
    free(normals);
    return NULL;
}

static int oracle_submit(gorder_oracle_handle *h, const float *xyz, const float *box,
                         const uint64_t *frame_index, uint32_t n_frames, uint32_t passes);
int gorder_oracle_submit(gorder_oracle_handle *h, const float *xyz, const float *box,
                         const uint64_t *frame_index, uint32_t n_frames) {
    return oracle_submit(h, xyz, box, frame_index, n_frames, 1);
}
/* Benchmark aid (bench.py's cpu_baseline): the same batch analysed `passes` times by the SAME worker threads — every
 * thread walks its frames t, t + n, ... `passes` times into its accumulator clone, one ordered reduce at the end —, so
 * that a timing does not consist of thread start-up (the reference keeps its threads for the whole trajectory,
 * groan_rs traj_iter_map_reduce; common.rs:283-339).  Sums and counts come out `passes` times those of one pass; the
 * frame counter advances by n_frames only; not meant for timewise rows. */
int gorder_oracle_submit_passes(gorder_oracle_handle *h, const float *xyz, const float *box,
                                const uint64_t *frame_index, uint32_t n_frames, uint32_t passes) {
    return oracle_submit(h, xyz, box, frame_index, n_frames, passes < 1 ? 1 : passes);
}
static int oracle_submit(gorder_oracle_handle *h, const float *xyz, const float *box,
                         const uint64_t *frame_index, uint32_t n_frames, uint32_t passes) {
    if (!h || !xyz || (!box && h->pbc) || !frame_index) return GORDER_ERR_INVALID_ARGUMENT;
    if (n_frames == 0) return GORDER_OK;
    const size_t row = 3 * (size_t)h->n_acc;
    uint8_t *frame_flags = NULL;
    int status = GORDER_OK;
    /* The error a batch reports is the one the reference's single-threaded walk meets first: frame by frame, inside a
     * frame the box check, the leaflet assignment, then the order loop (common.rs:201-235, molecule.rs:54-95).  The
     * leaflet pass below runs ahead of the order pass, so an error it finds in frame f only stands if the order pass
     * over the frames before f is clean. */
    int lf_status = GORDER_OK, range_bad = 0;
    uint32_t lf_err_frame = n_frames;
    uint64_t lf_err_index = 0;
    /* pass 1: leaflet flags per frame (sequential; molecule.rs:61-70) */
    if (h->lf.method != GORDER_LEAFLETS_NONE) {
        frame_flags = (uint8_t *)malloc((size_t)h->n_mol_total * n_frames);
        for (uint32_t f = 0; f < n_frames; f++) {
            const uint64_t g = frame_index[f];
            if (h->lf.method != GORDER_LEAFLETS_MANUAL && should_assign(h->lf.frequency, g)) {
                float box3[3] = {0, 0, 0};
                if (h->pbc) {
                    lf_status = check_box(box + 9 * (size_t)f, box3);
                    if (lf_status == GORDER_ERR_BOX_RANGE) { lf_status = GORDER_OK; range_bad = 1; box3[0] = box3[1] = box3[2] = 1.0f; }
                    if (lf_status != GORDER_OK) { lf_err_frame = f; break; }
                }
                lf_status = assign_leaflets(h, xyz + 3 * (size_t)h->n_atoms * f, box3, h->flags, h->flag_dist);
                if (lf_status == GORDER_ERR_BOX_RANGE) { lf_status = GORDER_OK; range_bad = 1; }
                if (lf_status != GORDER_OK) { lf_err_frame = f; lf_err_index = h->err_index; break; }
                h->have_flags = 1;
                h->flags_frame = g;
            }
            if (!h->have_flags) { status = GORDER_ERR_LEAFLETS_NOT_PRIMED; goto done; }
            memcpy(frame_flags + (size_t)h->n_mol_total * f, h->flags, h->n_mol_total);
        }
    }
    /* timewise rows for this batch */
    int64_t *tw_s = NULL; uint64_t *tw_n = NULL;
    if (h->timewise) {
        if (h->n_frames + n_frames > h->tw_cap) {
            h->tw_cap = (h->n_frames + n_frames) * 2;
            h->tw_sums = (int64_t *)realloc(h->tw_sums, h->tw_cap * row * sizeof(int64_t));
            h->tw_counts = (uint64_t *)realloc(h->tw_counts, h->tw_cap * row * sizeof(uint64_t));
        }
        tw_s = h->tw_sums + row * h->n_frames;
        tw_n = h->tw_counts + row * h->n_frames;
        memset(tw_s, 0, row * n_frames * sizeof(int64_t));
        memset(tw_n, 0, row * n_frames * sizeof(uint64_t));
    }
    {
        const uint32_t n_run = lf_err_frame;   /* frames the order pass may look at (all of them without a leaflet error) */
        uint32_t nthr = (uint32_t)h->n_threads;
        if (nthr > n_run) nthr = n_run;
        uint32_t err_frame = n_frames;
        o_job *jobs = (o_job *)calloc(nthr, sizeof(o_job));
        pthread_t *th = (pthread_t *)calloc(nthr, sizeof(pthread_t));
        for (uint32_t t = 0; t < nthr; t++) {
            jobs[t].h = h; jobs[t].xyz = xyz; jobs[t].box = box; jobs[t].frame_flags = frame_flags;
            jobs[t].n_frames = n_run; jobs[t].tid = t; jobs[t].nthr = nthr; jobs[t].passes = passes;
            jobs[t].manual = (h->manual_frames == n_frames) ? h->manual_normals : NULL;
            jobs[t].tw_s = tw_s; jobs[t].tw_n = tw_n; jobs[t].status = GORDER_OK;
            acc_alloc(h, &jobs[t].acc);
            if (nthr > 1) pthread_create(&th[t], NULL, worker, &jobs[t]);
            else worker(&jobs[t]);
        }
        const size_t ntile = (size_t)h->nx * h->ny;
        for (uint32_t t = 0; t < nthr; t++) {
            if (nthr > 1) pthread_join(th[t], NULL);
            if (jobs[t].range_bad) range_bad = 1;
            if (jobs[t].status != GORDER_OK && (status == GORDER_OK || jobs[t].err_frame < err_frame)) {
                status = jobs[t].status;          /* the error of the earliest frame */
                err_frame = jobs[t].err_frame;
                h->err_index = jobs[t].err_index;
            }
            /* SystemTopology::add, topology/mod.rs:236-254 */
            for (size_t k = 0; k < row; k++) {
                if (__builtin_add_overflow(h->acc.sums[k], jobs[t].acc.sums[k], &h->acc.sums[k]))
                    status = GORDER_ERR_OVERFLOW; /* order.rs:44-60 panics */
                h->acc.counts[k] += jobs[t].acc.counts[k];
            }
            if (h->om.enabled)
                for (size_t k = 0; k < row * ntile; k++) {
                    h->acc.map_sums[k] += jobs[t].acc.map_sums[k];
                    h->acc.map_counts[k] += jobs[t].acc.map_counts[k];
                }
            acc_free(&jobs[t].acc);
        }
        free(jobs); free(th);
    }
    if (status == GORDER_OK && lf_status != GORDER_OK) { status = lf_status; h->err_index = lf_err_index; }
    if (status == GORDER_OK && range_bad) status = GORDER_ERR_BOX_RANGE;
    if (status == GORDER_OK) h->n_frames += n_frames;
done:
    h->manual_frames = 0;   /* manual normals apply to one submit call */
    free(frame_flags);
    return status;
}

int gorder_oracle_prime_leaflets(gorder_oracle_handle *h, const float *xyz, const float *box,
                                 uint64_t frame_index) {
    float box3[3] = {0, 0, 0};
    if (h->pbc) {
        const int st = check_box(box, box3);
        if (st != GORDER_OK) return st;
    }
    const int st = assign_leaflets(h, xyz, box3, h->flags, h->flag_dist);
    if (st == GORDER_OK) { h->have_flags = 1; h->flags_frame = frame_index; }
    return st;
}

int gorder_oracle_set_manual_leaflets(gorder_oracle_handle *h, const uint8_t *flags, uint64_t frame_index) {
    for (uint32_t i = 0; i < h->n_mol_total; i++) h->flags[i] = (uint8_t)((flags[i] & 1) ^ (h->lf.flip ? 1 : 0));
    h->have_flags = 1;
    h->flags_frame = frame_index;
    return GORDER_OK;
}

int gorder_oracle_finish(gorder_oracle_handle *h, int64_t *sums, uint64_t *counts, int64_t *map_sums,
                         uint64_t *map_counts, uint64_t *n_frames_analyzed) {
    const size_t row = 3 * (size_t)h->n_acc;
    if (sums) memcpy(sums, h->acc.sums, row * sizeof(int64_t));
    if (counts) memcpy(counts, h->acc.counts, row * sizeof(uint64_t));
    if (h->om.enabled) {
        const size_t n = row * h->nx * h->ny;
        if (map_sums) memcpy(map_sums, h->acc.map_sums, n * sizeof(int64_t));
        if (map_counts) memcpy(map_counts, h->acc.map_counts, n * sizeof(uint64_t));
    }
    if (n_frames_analyzed) *n_frames_analyzed = h->n_frames;
    return GORDER_OK;
}

int gorder_oracle_timewise(gorder_oracle_handle *h, int64_t *tw_sums, uint64_t *tw_counts,
                           uint64_t capacity_frames) {
    if (!h->timewise) return GORDER_ERR_INVALID_ARGUMENT;
    if (capacity_frames < h->n_frames) return GORDER_ERR_INVALID_ARGUMENT;
    const size_t row = 3 * (size_t)h->n_acc;
    memcpy(tw_sums, h->tw_sums, row * h->n_frames * sizeof(int64_t));
    memcpy(tw_counts, h->tw_counts, row * h->n_frames * sizeof(uint64_t));
    return GORDER_OK;
}

int gorder_oracle_set_normals(gorder_oracle_handle *h, const float *normals, uint32_t n_frames) {
    if (!h || !normals || n_frames == 0) return GORDER_ERR_INVALID_ARGUMENT;
    const size_t n = 3 * (size_t)n_frames * h->n_mol_total;
    h->manual_normals = (float *)realloc(h->manual_normals, n * sizeof(float));
    memcpy(h->manual_normals, normals, n * sizeof(float));
    h->manual_frames = n_frames;
    return GORDER_OK;
}

int gorder_oracle_normals(gorder_oracle_handle *h, float *normals, uint32_t *n_points) {
    if (!h || !h->dyn.enabled) return GORDER_ERR_INVALID_ARGUMENT;
    for (uint32_t m = 0; m < h->n_mol_total; m++) {
        if (normals) memcpy(normals + 3 * (size_t)m, h->last_normals + 4 * (size_t)m, 3 * sizeof(float));
        if (n_points) n_points[m] = (uint32_t)h->last_normals[4 * (size_t)m + 3];
    }
    return GORDER_OK;
}

int gorder_oracle_leaflets(gorder_oracle_handle *h, uint8_t *flags, float *distances,
                           uint64_t *assignment_frame) {
    if (!h->have_flags) return GORDER_ERR_LEAFLETS_NOT_PRIMED;
    if (flags) memcpy(flags, h->flags, h->n_mol_total);
    if (distances) memcpy(distances, h->flag_dist, sizeof(float) * h->n_mol_total);
    if (assignment_frame) *assignment_frame = h->flags_frame;
    return GORDER_OK;
}
