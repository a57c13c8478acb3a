/*
 * gorder_oracle.h — CPU restatement of gorder's per-frame order-parameter path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gorder_amd/, include/) links, imports or
 * executes this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * there only as the checker / reported baseline.
 *
 * The oracle consumes the same plain-C tables as the HIP library (include/gorder_hip.h) so that the
 * parity tests are symmetric, but shares no code with it.
 */
#ifndef GORDER_ORACLE_H
#define GORDER_ORACLE_H

#include "../include/gorder_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How acos / cos of `calc_sch` (src/analysis/mod.rs:78-82) are evaluated:
 *   LIBM   — host libm acosf/cosf: what the Rust reference itself calls on linux-gnu
 *            (f32::acos -> acosf, f32::cos -> cosf).  This is the reference-faithful mode.
 *   DIRECT — cos(theta) = the clamped cosine itself, no acos -> cos round trip: restates the device
 *            library's default mode (include/gorder_hip.h, gorder_flags_t) for EQUALITY checks.
 *   MIRROR — the oracle's own restatement (in portable C, fmaf only) of the polynomial kernels the
 *            device code uses, so that device-vs-oracle i64 sums can be compared for EQUALITY. */
enum { GORDER_ORACLE_TRIG_LIBM = 0, GORDER_ORACLE_TRIG_MIRROR = 1, GORDER_ORACLE_TRIG_DIRECT = 2 };

typedef struct gorder_oracle_handle gorder_oracle_handle;

int gorder_oracle_create(const gorder_tables_t *tables, int trig_mode, int n_threads,
                         gorder_oracle_handle **out);
void gorder_oracle_destroy(gorder_oracle_handle *h);
uint32_t gorder_oracle_n_accumulators(const gorder_oracle_handle *h);
uint32_t gorder_oracle_ordermap_dims(const gorder_oracle_handle *h, uint32_t *nx, uint32_t *ny);

/* xyz [n_frames][n_atoms][3], box [n_frames][3][3], frame_index [n_frames]; host memory */
int gorder_oracle_submit(gorder_oracle_handle *h, const float *xyz, const float *box,
                         const uint64_t *frame_index, uint32_t n_frames);
/* benchmark aid: the same batch walked `passes` times by the same worker threads (sums come out `passes` times as large) */
int gorder_oracle_submit_passes(gorder_oracle_handle *h, const float *xyz, const float *box,
                                const uint64_t *frame_index, uint32_t n_frames, uint32_t passes);
int gorder_oracle_prime_leaflets(gorder_oracle_handle *h, const float *xyz, const float *box,
                                 uint64_t frame_index);
int gorder_oracle_set_manual_leaflets(gorder_oracle_handle *h, const uint8_t *flags,
                                      uint64_t frame_index);
int gorder_oracle_finish(gorder_oracle_handle *h, int64_t *sums, uint64_t *counts,
                         int64_t *map_sums, uint64_t *map_counts, uint64_t *n_frames_analyzed);
int gorder_oracle_timewise(gorder_oracle_handle *h, int64_t *tw_sums, uint64_t *tw_counts,
                           uint64_t capacity_frames);
/* flags of the most recent assignment frame + the signed distances they were derived from */
/* manual membrane normals [n_frames][n_mol_total][3] for the next submit call (see gorder_hip_set_normals) */
int gorder_oracle_set_normals(gorder_oracle_handle *h, const float *normals, uint32_t n_frames);
/* dynamic membrane normals of the last analysed frame (see gorder_hip_normals) */
int gorder_oracle_normals(gorder_oracle_handle *h, float *normals, uint32_t *n_points);
/* one normal: normal[4] = (nx, ny, nz, number of cloud points) — normal.rs:160-199, 421-458 */
int gorder_oracle_dynamic_normal(const float *xyz, const uint32_t *cloud, uint32_t n_cloud, uint32_t head,
                                 float radius, const float box[3], int handle_pbc, float normal[4]);
int gorder_oracle_leaflets(gorder_oracle_handle *h, uint8_t *flags, float *distances,
                           uint64_t *assignment_frame);
uint64_t gorder_oracle_last_error_index(const gorder_oracle_handle *h);

/* ---- scalar building blocks, exported for known-answer tests ------------------------------- */
/* groan_rs Vector3D::vector_to (assumed semantics) ; out = shortest p1->p2 */
int gorder_oracle_vector_to(const float p1[3], const float p2[3], const float box[3], int handle_pbc,
                            float out[3]);
/* calc_sch (mod.rs:78-82) */
float gorder_oracle_calc_sch(const float v[3], const float n[3], int trig_mode);
/* OrderValue::from(f32) (order.rs:21-26) */
int64_t gorder_oracle_tick(float s);
/* AnalysisOrder::calc_order (order.rs:101-107): f32::from(sum / n), NaN if n < min_samples */
float gorder_oracle_calc_order(int64_t sum, uint64_t n, uint64_t min_samples);
/* mirror-mode primitives */
float gorder_oracle_mirror_acosf(float x);
float gorder_oracle_mirror_cosf(float x);
float gorder_oracle_mirror_sinf(float x);   /* x in [0, pi] */
/* fn 0 acos, 1 cos, 2 sin of the floats with bit patterns first_bits + i * stride (i < n); which = 0 the restatement of
 * glibc's algorithms above (= what the device computes), 1 the host's libm */
void gorder_oracle_trig_batch(int fn, int which, uint32_t first_bits, uint32_t stride, uint32_t n, float *out);
/* UA hydrogen construction (uaorder.rs:947-1104); pos = [4][3] in the order of `indices`;
 * out = [n_h][3]; returns n_h */
/* the device's GORDER_FLAG_UA_FAST_NORMALISE construction restated (hydrogens, vectors target -> H, and whether the
 * device would re-evaluate the carbon with the literal loops) — test / fidelity tooling */
int gorder_oracle_predict_hydrogens_fast(uint32_t kind, const float pos[4][3], const float box[3], int pbc,
                                         float out[3][3], float vec[3][3], int *slow);
/* per-sample comparison of the fast construction with the reference's arithmetic (see the definition) */
int gorder_oracle_ua_fast_fidelity(const gorder_oracle_handle *h, const float *xyz, const float *box9, uint32_t n_frames,
                                   uint64_t hist[16], uint64_t out[5]);
int gorder_oracle_predict_hydrogens(uint32_t kind, const float pos[4][3], const float box[3],
                                    int handle_pbc, float out[3][3]);
/* TimeWiseData::estimate_error (timewise.rs:191-231) and prefix_average (:259-274) */
float gorder_oracle_estimate_error(const int64_t *sums, const uint64_t *counts, uint64_t n_frames,
                                   uint64_t n_blocks);
void gorder_oracle_prefix_average(const int64_t *sums, const uint64_t *counts, uint64_t n_frames,
                                  float *out);
/* PBC-aware centre of geometry (groan_rs refined Bai-Breen, assumed semantics) */
int gorder_oracle_center(const float *xyz, const uint32_t *idx, uint32_t n, const float box[3],
                         int handle_pbc, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
