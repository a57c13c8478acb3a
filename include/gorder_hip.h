/*
 * gorder_hip.h — C ABI of the MI355X (gfx950) lipid-order engine.
 *
 * This is the drop-in boundary for ONE path of VachaLab/gorder: the per-frame
 * order-parameter computation that gorder runs inside
 *   groan_rs `System::traj_iter_map_reduce(.., body = analyze_frame, ..)`
 *   (call sites  src/analysis/common.rs:283-339, body  src/analysis/common.rs:201-235,
 *    data type  `SystemTopology`  src/analysis/topology/mod.rs:34-65,
 *    map/reduce trait impl  src/analysis/topology/mod.rs:256-278).
 *
 * Mapping of the reference's interface onto this ABI
 *   SystemTopology::new  (topology/mod.rs:70-118)        -> gorder_hip_create
 *   ParallelTrajData::initialize (topology/mod.rs:274-277) -> one handle per GPU/rank; the caller
 *                                                            passes GLOBAL frame indices to submit
 *   analyze_frame (common.rs:201-235), called per frame   -> gorder_hip_submit_{device,host}, called
 *                                                            per BATCH of frames (AoS xyz as decoded)
 *   SystemTopology::add / reduce (topology/mod.rs:236-272) -> gorder_hip_finish returns raw i64 sums +
 *                                                            u64 counts which add element-wise
 *                                                            (order.rs:160-176, ordermap.rs:116-138);
 *                                                            on a multi-GPU node one RCCL all-reduce
 *                                                            (gorder_hip_allreduce, or the host's own
 *                                                            over gorder_hip_accumulators_device())
 *   read_trajectory (common.rs:239-342)                    -> gorder_hip_run_trajectory
 *   Result<(), AnalysisError> (errors.rs:121-168)          -> gorder_status_t + gorder_hip_last_error_index
 *
 * Plain pointers and sizes only.  No allocation ownership crosses the boundary: every input
 * buffer stays owned by the caller, every output buffer is caller-allocated.
 * A handle is thread-compatible (one handle per host thread / GPU / stream), exactly like one
 * `SystemTopology` clone per thread in the reference.
 */
#ifndef GORDER_HIP_H
#define GORDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "gorder_xtc.h"   /* gorder_xtc_frame_t for gorder_hip_xtc_decode */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------------------------
 * 1..7 mirror the AnalysisError variants the per-frame path can raise (src/errors.rs:121-141).
 * >= 100 are errors of this library (no counterpart in the reference). */
typedef enum {
    GORDER_OK = 0,
    GORDER_ERR_UNDEFINED_BOX = 1,                   /* AnalysisError::UndefinedBox            errors.rs:124 */
    GORDER_ERR_NOT_ORTHOGONAL_BOX = 2,              /* AnalysisError::NotOrthogonalBox        errors.rs:128 */
    GORDER_ERR_ZERO_BOX = 3,                        /* AnalysisError::ZeroBox                 errors.rs:132 */
    GORDER_ERR_UNDEFINED_POSITION = 4,              /* AnalysisError::UndefinedPosition(idx)  errors.rs:136 */
    GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER = 5,  /* ...::InvalidGlobalMembraneCenter       errors.rs:139 */
    GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER = 6,   /* ...::InvalidLocalMembraneCenter(idx)   errors.rs:143 */
    GORDER_ERR_DYNAMIC_NORMAL = 7,                  /* ...::DynamicNormalError(NotEnoughPoints(n)) errors.rs:155,
                                                       172-177; gorder_hip_last_error_index = n */
    GORDER_ERR_INVALID_ARGUMENT = 100,
    GORDER_ERR_DEVICE = 101,          /* a HIP runtime call failed; see gorder_hip_last_error_message */
    GORDER_ERR_NO_DEVICE = 102,       /* no gfx950 device visible: the product path has NO CPU fallback */
    GORDER_ERR_BOX_RANGE = 103,       /* a box edge is <= 0 / NaN, or a coordinate lies so far outside
                                         the box that the reference's `while` minimum-image loop would not
                                         terminate in 8 iterations (the reference would spin) */
    GORDER_ERR_LEAFLETS_NOT_PRIMED = 104, /* first submitted frame is not an assignment frame and no
                                             earlier assignment is known: call gorder_hip_prime_leaflets */
    GORDER_ERR_OVERFLOW = 105,        /* a batch was refused because frames x molecules would reach 2^63 / 1e6: an i64
                                         order sum could then overflow (the reference panics, order.rs:44-60) */
    GORDER_ERR_TRAJECTORY_FORMAT = 106 /* a corrupt or truncated trajectory frame, met by gorder_hip_xtc_decode on the device or
                                         by the host reader inside gorder_hip_run_trajectory (the reference: a read error
                                         of the trajectory iterator, common.rs:248) */
} gorder_status_t;

/* ---- leaflets -------------------------------------------------------------------------------- */
typedef enum {
    GORDER_LEAFLETS_NONE = 0,
    GORDER_LEAFLETS_GLOBAL = 1,      /* leaflets.rs:171-205 + :711-732 */
    GORDER_LEAFLETS_LOCAL = 2,       /* leaflets.rs:661-675 + pbc.rs:273-318 + :711-732 */
    GORDER_LEAFLETS_INDIVIDUAL = 3,  /* leaflets.rs:777-801 */
    GORDER_LEAFLETS_MANUAL = 4       /* host supplies flags per assignment frame (leaflets.rs:820-860);
                                        Leaflet encoding Upper=0, Lower=1 (lib.rs:416-422) */
} gorder_leaflet_method_t;

typedef struct {
    uint32_t method;        /* gorder_leaflet_method_t */
    uint32_t normal_dim;    /* 0=x 1=y 2=z : `membrane_normal: Dimension` of the classifier */
    uint32_t frequency;     /* 0 = Frequency::Once, n>=1 = Frequency::Every(n); this is the REAL
                               frequency = input frequency * step (leaflets.rs:157-163) */
    uint32_t flip;          /* leaflets.rs:68-73 */
    float radius;           /* Local: cylinder radius in nm (leaflets.rs:389-418) */
    uint32_t n_membrane;    /* Global/Local: size of group "Membrane" */
    const uint32_t *membrane; /* atom indices (into the submitted coordinate frame) */
} gorder_leaflets_t;

/* ---- ordermaps (src/analysis/ordermap.rs:40-113, input/ordermap.rs:34-50) --------------------- */
typedef struct {
    uint32_t enabled;
    uint32_t plane;         /* 0=xy, 1=xz, 2=yz (yz projects to (z, y), input/ordermap.rs:48) */
    float span_x[2];        /* resolved span; GridSpan::Auto => (0, box) of the STRUCTURE file */
    float span_y[2];
    float bin[2];
} gorder_ordermap_t;

/* ---- molecule types ---------------------------------------------------------------------------
 * United-atom carbon kinds (uaorder.rs:448-452). */
typedef enum {
    GORDER_UA_CH1_SAT = 1,   /* indices = helper1, helper2, helper3, target  (uaorder.rs:1050-1055) */
    GORDER_UA_CH2 = 2,       /* indices = helper1, target, helper2, -        (uaorder.rs:911-915)  */
    GORDER_UA_CH3 = 3,
    GORDER_UA_CH1_UNSAT = 4
} gorder_ua_kind_t;

typedef struct {
    uint32_t kind;             /* gorder_ua_kind_t; number of virtual C-H bonds = 1,2,3,1 */
    const uint32_t *indices;   /* [n_molecules][4] */
} gorder_ua_atom_t;

typedef struct {
    uint32_t n_molecules;
    /* AA / CG (BondType, topology/bond.rs:220-246): */
    uint32_t n_bond_types;
    const uint32_t *bonds;     /* [n_bond_types][n_molecules][2] atom indices, first < second
                                  (bond.rs:300-304, 338-344) */
    /* UA (UAOrderAtoms, topology/uatom.rs): */
    uint32_t n_ua_atoms;
    const gorder_ua_atom_t *ua_atoms;
    /* leaflet classifier inputs (leaflets.rs:575-590, 744-775): */
    const uint32_t *heads;     /* [n_molecules] or NULL */
    uint32_t n_methyls;        /* equal for all molecules of a type (leaflets.rs:760-770) */
    const uint32_t *methyls;   /* [n_molecules][n_methyls] or NULL */
    /* dynamic membrane normal (normal.rs:145-158, get_reference_head): */
    const uint32_t *normal_heads; /* [n_molecules] the molecule's atom of group "NormalHeads", or NULL */
} gorder_moltype_t;

/* ---- dynamic membrane normals (src/analysis/normal.rs:160-199, 421-458; pbc.rs:142-161, 321-350) ----
 * Per frame and molecule: the cloud of "NormalHeads" atoms within `radius` (3-D, minimum image) of the
 * molecule's own head, made whole around it; the normal is the direction of least variance of the cloud
 * (last right singular vector of the demeaned cloud).  Fewer than 3 points: GORDER_ERR_DYNAMIC_NORMAL,
 * raised only if a sample of that molecule is actually accumulated (the reference computes normals lazily,
 * after the geometry test, bond.rs:424-431).  With `enabled` the static `normal` of the tables is unused. */
typedef struct {
    uint32_t enabled;
    float radius;
    uint32_t n_cloud;
    const uint32_t *cloud;     /* atom indices of group "NormalHeads" (every atom the heads query selects) */
} gorder_dynamic_normal_t;

/* ---- geometry selection (src/analysis/geometry.rs:24-136, 181-210; input/geometry.rs) ----------
 * Only samples whose bond position lies inside (or, inverted, outside) the shape are accumulated
 * (bond.rs:424-426, uaorder.rs:388-390).  The shape is re-anchored every frame (geometry.rs:192-210). */
typedef enum { GORDER_GEOM_NONE = 0, GORDER_GEOM_CUBOID = 1, GORDER_GEOM_CYLINDER = 2, GORDER_GEOM_SPHERE = 3 } gorder_geom_kind_t;
typedef enum { GORDER_GEOMREF_POINT = 0, GORDER_GEOMREF_BOX_CENTER = 1, GORDER_GEOMREF_GROUP = 2 } gorder_geom_ref_t;
typedef struct {
    uint32_t kind;          /* gorder_geom_kind_t */
    uint32_t invert;
    uint32_t reference;     /* gorder_geom_ref_t: fixed point / box centre / centre of geometry of a group */
    float point[3];
    uint32_t n_group;
    const uint32_t *group;  /* atom indices of group "GeomReference" */
    float xdim[2], ydim[2], zdim[2];   /* cuboid extents relative to the reference; {-inf, +inf} = unbounded */
    float radius;           /* cylinder, sphere */
    float span[2];          /* cylinder: extent along its axis relative to the reference; {-inf, +inf} = unbounded */
    uint32_t orientation;   /* cylinder axis 0/1/2 */
    float structure_box[3]; /* box of the STRUCTURE file: a shape anchored at a fixed point is built (and its
                               anchor wrapped) once, at setup, with that box (geometry.rs:296-311 + :194);
                               box-centre / group references are rebuilt with each frame's box */
} gorder_geometry_t;

/* How P2(cos theta) of calc_sch (mod.rs:78-82) is evaluated.
 * Default (0): from the SQUARED cosine, q = (v.n)^2 / (|v|^2 |n|^2), S = 1.5 q - 0.5: one IEEE division, no square
 *   root, no acos -> cos round trip.  The reference evaluates `angle = acos(clamp(v.n / (|v||n|)))` and then
 *   `angle.cos()` in f32 — mathematically the same number plus <= 2 ulp of rounding noise.  Measured against the
 *   reference pipeline with glibc's acosf / cosf (tools/trig_fidelity.c): 5.9 % of samples move by one 1e-6 tick, never
 *   more, mean shift 2.6e-10 — inside the 1e-6 parity tolerance but NOT bit-identical to the reference.  The squared
 *   form also has half the exponent range (|v| below ~1e-19 nm or above ~1e19 nm is no longer the reference's value).
 * GORDER_FLAG_TRIG_ACOS_COS: evaluate the literal acos -> cos round trip like the reference, with acos and cos computed
 *   BY GLIBC'S ALGORITHMS (restatements of glibc 2.28 - 2.40's acosf and cosf, gm_math.h; rounds 1-3 used own polynomial
 *   kernels, 1.6 % of the ticks one off).  Rust's f32::acos / f32::cos are the platform libm's acosf / cosf: against a
 *   reference built on such a glibc the order sums of this mode are the reference's integers — EQUAL to the oracle's LIBM
 *   mode in the tests, the device's acos / cos / sin compared with the host's over their whole domains
 *   (gorder_hip_selftest_trig).  The price is the f64 polynomial of glibc's cosf: 51 % of the HBM roofline on the
 *   256-lipid membrane where the default runs at 80 %.
 *   The one data-dependent angle of the united-atom construction (unsaturated CH: acos, sin, cos, uaorder.rs:1024-1045)
 *   goes through the same restatements in EVERY mode.
 * GORDER_FLAG_UA_FAST_NORMALISE (united atoms only, opt-in, default off): the virtual-hydrogen construction
 *   (uaorder.rs:947-1104) with tolerance-bounded arithmetic instead of the reference's operation sequence — a / |a| as
 *   a * rsqrt(|a|^2) (integer seed + three Newton steps, relative error ~1e-7 where the reference's two roundings leave
 *   6e-8), no second normalisation of the CH2 rotation axis (uaorder.rs:990-1003), minimum image / wrap as
 *   d - L rint(d / L) and x - L floor(x / L), the ordermap tile of a virtual C-H bond as floor((x - x0) * (1 / bin) + 1/2)
 *   by one fma instead of the rounded IEEE quotient (a position within an ulp or two of the line between two tiles may
 *   land on the other side: 0 of 522 036 on the reference's membrane, 5 of 1 015 808 on the synthetic one).  Every
 *   operation is an IEEE mul / add / fma, restated by the oracle's FAST mode: device and oracle sums stay EQUAL.  Against the reference-faithful (libm) oracle every order parameter
 *   stays within one 1e-6 tick; single samples move more often than with the default (a hydrogen position is
 *   target + 0.109 nm * unit vector rounded to the grid of the target's coordinates, so an ulp in the unit vector
 *   flips that rounding in ~1 % of the components): measured in profiles/r04_ua_fast_fidelity.json.  Not with
 *   GORDER_FLAG_TRIG_ACOS_COS (gorder_hip_create refuses the pair).  The default path is bit-unchanged. */
typedef enum {
    GORDER_FLAG_TRIG_ACOS_COS = 1u,
    GORDER_FLAG_UA_FAST_NORMALISE = 2u
} gorder_flags_t;

typedef struct {
    uint32_t n_atoms;           /* atoms per submitted frame (the decoded "Master" group, common.rs:283-304) */
    uint32_t n_molecule_types;
    const gorder_moltype_t *molecule_types;
    int32_t handle_pbc;         /* 1 = PBC3D (pbc.rs:257-460), 0 = NoPBC (pbc.rs:98-253) */
    float normal[3];            /* static membrane normal (normal.rs:75-87, input/axis.rs:49-58) */
    gorder_leaflets_t leaflets;
    gorder_ordermap_t ordermap;
    int32_t timewise;           /* 1 = keep per-frame partial sums (estimate_error; timewise.rs:130-186) */
    int32_t device;             /* HIP device ordinal */
    uint32_t flags;             /* gorder_flags_t */
    gorder_geometry_t geometry; /* kind = GORDER_GEOM_NONE: every sample counts (geometry.rs:215-284) */
    gorder_dynamic_normal_t dynamic_normal;
} gorder_tables_t;

/* Accumulator slots are numbered in reference iteration order: molecule type major, then bond
 * type (AA/CG) or (united atom, hydrogen) (UA).  n_acc = number of slots.  All result arrays are
 * laid out [3][n_acc] with the leading index 0=total, 1=upper, 2=lower (bond.rs:233-246). */

typedef struct gorder_hip_handle gorder_hip_handle;

int gorder_hip_create(const gorder_tables_t *tables, gorder_hip_handle **out);
void gorder_hip_destroy(gorder_hip_handle *h);

/* number of accumulator slots / ordermap tiles (0 if maps are off) */
uint32_t gorder_hip_n_accumulators(const gorder_hip_handle *h);
uint32_t gorder_hip_ordermap_dims(const gorder_hip_handle *h, uint32_t *nx, uint32_t *ny);

/* Run the launches of this handle on a caller-owned hipStream_t (e.g. PyTorch's current stream).
 * NULL selects the handle's own stream. */
int gorder_hip_set_stream(gorder_hip_handle *h, void *hip_stream);

/* Analyse a batch of frames whose coordinates are ALREADY in HBM.
 *   d_xyz  [n_frames][n_atoms][3] f32 (AoS, exactly as an XTC decoder emits them), 16-byte aligned
 *   d_box  [n_frames][3][3]       f32 box matrix rows (GROMACS/XTC order); must be diagonal
 *   frame_index [n_frames] (HOST) global frame indices = `SystemTopology::frame` (topology/mod.rs:141-144)
 * Asynchronous on the handle's stream; errors raised on the device surface at the next
 * gorder_hip_synchronize / gorder_hip_finish. */
int gorder_hip_submit_device(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                             const uint64_t *frame_index, uint32_t n_frames);

/* Same, from host memory (pinned or pageable).  Double-buffered: the batch is copied on a copy stream into one of
 * two internal device buffers while the kernels of the previous batch may still be running; the call returns once
 * the copy has left the host buffer (the caller may refill it at once), NOT once the kernels are done.  A host that
 * decodes batch k+1 between two calls therefore overlaps decoding, the PCIe copy and the kernels. */
int gorder_hip_submit_host(gorder_hip_handle *h, const float *xyz, const float *box,
                           const uint64_t *frame_index, uint32_t n_frames);

/* ---- trajectory driver: `read_trajectory` (src/analysis/common.rs:239-342) --------------------------------------
 * Reads one trajectory (XTC, TRR or multi-frame GRO; several files = one concatenated trajectory with the duplicate
 * boundary frames dropped), applies the time window and the step, and feeds the frames to the handle batch by batch
 * through a pipeline: `n_threads` host threads decode batch k+2 into pinned memory while batch k+1 is copied on a
 * copy stream and batch k is analysed (see gorder_amd/csrc/trajectory_driver.h).  The k-th analysed frame gets the
 * global frame index first_frame_index + k * step (SystemTopology::frame, topology/mod.rs:141-144).
 * Returns after the last batch has been analysed (synchronised); results are read with gorder_hip_finish.
 * The first error — of the reader or of the analysis — ends the run (common.rs:248). */
typedef struct {
    const char *const *paths;
    uint32_t n_paths;
    const uint32_t *group;       /* atoms of the file that make up a submitted frame, in order (the "Master" group,
                                    common.rs:283-304); NULL = every atom */
    uint32_t n_group;
    float begin_ps, end_ps;      /* inclusive window on the frame time; end_ps < 0 = to the end */
    uint32_t step;               /* analyse every step-th frame of the window (>= 1) */
    uint32_t n_threads;          /* decoder / copying threads; 0 = the hardware threads, at most 16 */
    uint32_t batch_frames;       /* frames per device batch; 0 = about 128 MB of coordinates (1 GiB with device_decode) */
    uint64_t first_frame_index;  /* 0 for a whole trajectory; a rank that reads a later window passes where it starts */
    uint32_t device_decode;      /* 1: XTC files are decompressed ON THE DEVICE — the host threads only copy the compressed
                                    blocks into pinned memory (gorder_xtc_pack_window), the copy engine moves those
                                    (about a third of the decoded bytes) and gorder_hip_xtc_decode unpacks them (k_xtc_scan:
                                    a wave per frame finds where every 256-atom chunk starts; k_xtc_chunks: a lane per
                                    chunk decodes).  Same coordinates bit for bit.  A run with a TRR or GRO file in it, or of frames
                                    so large that fewer than 512 fit a 4-GiB batch (about 700 000 analysed atoms; a launch
                                    takes 0.6 us per atom whatever its size), uses the host decoder.  When the analysed
                                    atoms end before the frame does, only the leading part of every compressed block the
                                    decoder needs is copied (learned from the first batches; a frame that needs more is
                                    decoded by the host).  0: host decoder threads */
    uint32_t shard_index;        /* SURVEY 8e, contiguous frame shards: with shard_count = n > 1 the call first counts the F frames */
    uint32_t shard_count;        /* the window selects (headers only), then analyses frames [i F / n, (i + 1) F / n) of them,
                                    numbered as in the whole trajectory (first_frame_index + k * step for the k-th selected
                                    frame).  Every rank passes the same trajectory and its own i; gorder_hip_allreduce (or the
                                    host's own reduction) then gives SystemTopology::reduce's result.  With a leaflet frequency
                                    other than every frame a shard that begins between two assignment frames reads the one
                                    frame it depends on itself and primes the handle with it (gorder_hip_prime_leaflets).
                                    0 or 1: the whole trajectory */
    uint32_t reserved;
} gorder_trajectory_t;

typedef struct {
    uint64_t n_frames;               /* frames analysed */
    uint64_t n_batches;
    uint64_t bytes_h2d;              /* coordinate + box bytes copied to the device */
    double seconds_total;            /* wall time of the call */
    double seconds_decode;           /* wall time inside the decoder (overlaps with copies and kernels) */
    double seconds_reader_stalled;   /* the reader waited for a free staging slot: copies / kernels are the bottleneck */
    double seconds_gpu_starved;      /* the submitter waited for a decoded batch: the decoder is the bottleneck */
    uint32_t batch_frames, decoder_threads;   /* what was used */
    uint32_t device_decode;          /* 1: the frames were decompressed on the device */
    uint32_t frames_decoded_by_host; /* device route: frames of which too short a leading part had been copied (see
                                        gorder_xtc_pack_window_ex) and which the host decoded after all */
    uint64_t shard_first, shard_frames_total;   /* with shards: ordinal of this shard's first frame among the F selected; F */
    double seconds_setup;            /* time spent allocating the pinned and device staging buffers (all but the first slot's
                                        share overlaps with reading) */
} gorder_trajectory_stats_t;

int gorder_hip_run_trajectory(gorder_hip_handle *h, const gorder_trajectory_t *trajectory,
                              gorder_trajectory_stats_t *stats /* may be NULL */);
/* The pinned host buffers and device buffers of gorder_hip_run_trajectory (with device_decode: 4 slots of a pinned
 * blob + its device copy + 1 GiB of coordinates) stay with the handle after the call, so that the next trajectory
 * analysed with it does not pay for pinning again (~0.1 s per GB); a run of a different shape (route, batch size)
 * replaces them, gorder_hip_destroy frees them — and so does this call, for a host that wants the memory back. */
void gorder_hip_release_staging(gorder_hip_handle *h);

/* Decompress XTC frames on the device (the decoding half of groan_rs' GroupXtcReader, common.rs:283-304):
 * `d_blob` / `d_frames` are device copies of what gorder_xtc_pack_window produced (d_blob 64-byte aligned, blob_bytes >= 64), `d_slot_of`
 * [n_atoms_file] maps a file atom to its place in the output frame or -1 (NULL: every atom, in order), `n_stop` is
 * the number of atoms to go through (gorder_xtc_n_atoms_needed), `d_xyz` [n_frames][n_atoms_out][3] receives exactly
 * what gorder_xtc_next would have written, bit for bit.  Asynchronous on the handle's stream; a corrupt frame is
 * reported as GORDER_ERR_TRAJECTORY_FORMAT by the next synchronising call.  Two kernels: k_xtc_scan (a wave per frame walks the
 * group headers, 64 groups a step, and leaves a checkpoint per 256 atoms) and k_xtc_chunks (a lane per chunk decodes from
 * its checkpoint); a frame table with widths no writer produces (more than 72 bits per atom, a field of more than 32, none at
 * all) is a format error too.  2.5 ms per 3 566 frames of 25 088 atoms. */
int gorder_hip_xtc_decode(gorder_hip_handle *h, const uint8_t *d_blob, uint64_t blob_bytes,
                          const gorder_xtc_frame_t *d_frames, uint32_t n_frames, uint32_t n_atoms_file,
                          const int32_t *d_slot_of, uint32_t n_stop, float *d_xyz, uint32_t n_atoms_out);

/* Compute (only) the leaflet assignment of ONE frame that precedes a rank's frame range
 * (replaces the cross-thread spin-wait of leaflets.rs:1529-1565). */
int gorder_hip_prime_leaflets(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                              uint64_t frame_index);

/* Manual assignment (GORDER_LEAFLETS_MANUAL): flags[n_molecules_total] for the assignment frame
 * that applies from `frame_index` on; molecules ordered molecule type major. */
int gorder_hip_set_manual_leaflets(gorder_hip_handle *h, const uint8_t *flags, uint64_t frame_index);

int gorder_hip_synchronize(gorder_hip_handle *h);

/* Synchronise and copy the accumulators out.  Leaves them intact (callers may keep submitting).
 *   sums   [3][n_acc] i64 : sum of round(f64(S) * 1e6)            (order.rs:13-26)
 *   counts [3][n_acc] u64 : n_samples                              (order.rs:178-188)
 * Optional (NULL to skip):
 *   map_sums / map_counts [3][n_acc][nx*ny]  (x-major, y inner; ordermap.rs:100-113)
 * n_frames_analyzed: `total_frames` (topology/mod.rs:141-144). */
int gorder_hip_finish(gorder_hip_handle *h, int64_t *sums, uint64_t *counts, int64_t *map_sums,
                      uint64_t *map_counts, uint64_t *n_frames_analyzed);

/* Per-frame partial sums (timewise.rs:130-186) of the frames submitted so far, in submission order:
 *   tw_sums / tw_counts [n_frames_analyzed][3][n_acc].  Requires tables.timewise = 1. */
int gorder_hip_timewise(gorder_hip_handle *h, int64_t *tw_sums, uint64_t *tw_counts,
                        uint64_t capacity_frames);

/* Leaflet flags of the most recent assignment frame, [n_molecules_total] (Upper=0, Lower=1). */
int gorder_hip_leaflets(gorder_hip_handle *h, uint8_t *flags, uint64_t *assignment_frame);
/* Signed distances (nm) behind those flags, [n_molecules_total] (leaflets.rs:725, 796). */
int gorder_hip_leaflet_distances(gorder_hip_handle *h, float *distances);

/* Manual membrane normals (MembraneNormal::Manual, ManualMembraneNormal::get_normal, normal.rs:266-300): the host
 * resolves the normals file and hands over, before a submit call, one vector per frame of that batch and per
 * molecule — normals [n_frames][n_molecules_total][3] (host memory, copied; any length, calc_sch normalises).
 * They replace the static / dynamic normal for exactly the next gorder_hip_submit_* call, whose n_frames must
 * match. */
int gorder_hip_set_normals(gorder_hip_handle *h, const float *normals, uint32_t n_frames);

/* Dynamic membrane normals of the LAST submitted frame: normals [n_molecules_total][3] (unit vectors; NaN
 * for a molecule whose cloud had fewer than 3 points), n_points [n_molecules_total] the cloud sizes
 * (what NormalsStorage keeps per frame, normal.rs:460-520).  Requires tables.dynamic_normal.enabled. */
int gorder_hip_normals(gorder_hip_handle *h, float *normals, uint32_t *n_points);

/* Device pointer + element count of the packed 64-bit accumulator block, n_u64 = 4 * n_acc + 1 words:
 *   { i64 sum_total[n_acc], i64 sum_upper[n_acc], u64 count_total[n_acc], u64 count_upper[n_acc], u64 total_frames }
 * (lower = total - upper: every sample is upper or lower, bond.rs:199-213; gorder_hip_finish derives it), so that
 * a host can issue ONE RCCL all-reduce (ncclInt64 / ncclSum) over it — the multi-GPU form of
 * SystemTopology::reduce (topology/mod.rs:256-272).  Ordermaps are NOT part of the block: see
 * gorder_hip_export_maps / gorder_hip_allreduce. */
int gorder_hip_accumulators_device(gorder_hip_handle *h, void **d_ptr, uint64_t *n_u64);

/* ---- multi-GPU reduce inside the library (for hosts without a collective library of their own) -------------------
 * gorder_hip_allreduce sums, in place and across the ranks of an initialised RCCL communicator (`ncclComm_t`, passed
 * as void *), everything SystemTopology::add sums (topology/mod.rs:236-254): the packed accumulator block (order sums,
 * counts, total_frames) and, with ordermaps, the maps — ONE group of ncclAllReduce(ncclInt64, ncclSum) calls on the
 * handle's stream, over xGMI on an MI355X node.  Call it once, after the rank's last batch; every rank then reads the
 * whole-trajectory result with gorder_hip_finish.  Per-frame timewise rows and leaflet flags are per-rank data and are
 * not touched (the host concatenates them by frame index).  RCCL is bound at the first call (librccl.so.1).
 * The communicator can come from the host's own RCCL calls or from the two helpers below:
 *   rank 0: gorder_hip_comm_unique_id(id) -> the host ships the 128 bytes to the other ranks (any channel) ->
 *   every rank: gorder_hip_comm_create(handle, id, n_ranks, rank, &comm) on its handle's device. */
int gorder_hip_comm_unique_id(uint8_t id[128]);
int gorder_hip_comm_create(gorder_hip_handle *h, const uint8_t id[128], int n_ranks, int rank, void **comm_out);
void gorder_hip_comm_destroy(void *comm);
int gorder_hip_allreduce(gorder_hip_handle *h, void *nccl_comm);

/* Forget everything accumulated so far (sums, counts, maps, timewise rows, frame count, leaflet carry, error
 * state): the handle is a fresh `SystemTopology` on the same tables.  Stream-ordered. */
int gorder_hip_reset(gorder_hip_handle *h);

/* Ordermaps for the same reduction: copies the folded maps, i64 sums and u64 counts laid out
 * [3][n_acc][nx*ny] (n_u64 = 3 * n_acc * nx * ny words each, see gorder_hip_ordermap_dims), into caller-owned
 * device buffers (e.g. two torch.int64 tensors) that the host all-reduces like the accumulator block
 * (Map::add, ordermap.rs:116-138).  Stream-ordered after everything submitted so far; synchronises. */
int gorder_hip_export_maps(gorder_hip_handle *h, void *d_sums, void *d_counts, uint64_t n_u64);

/* Make the handle accumulate into caller-owned device memory (>= n_u64 words, 8-byte aligned; e.g.
 * a torch.int64 tensor that the host then hands to torch.distributed.all_reduce = RCCL).  The
 * current contents of the handle's accumulators are copied over. */
int gorder_hip_bind_accumulators(gorder_hip_handle *h, void *d_ptr, uint64_t n_u64);

/* Payload of the last UndefinedPosition / InvalidLocalMembraneCenter error (atom index). */
uint64_t gorder_hip_last_error_index(const gorder_hip_handle *h);
/* The frame (SystemTopology::frame, i.e. the frame_index the host passed) in which the last device error was raised:
 * the first error of the run in trajectory order — batches in the order of submission, inside a batch the order of the
 * reference's sequential walk (common.rs:201-235, 248). */
uint64_t gorder_hip_last_error_frame(const gorder_hip_handle *h);
const char *gorder_hip_last_error_message(const gorder_hip_handle *h);
const char *gorder_hip_strerror(int status);

/* Device time of the submits since the last call with reset != 0 (ms, HIP events on the stream the handle launches on) and
 * their number.  The first call switches the timing on; submits before it are not timed.  A submit is timed as a chain of
 * segments, one per kernel group it queues — the leaflet kernels ("k_leaflets_global_contig"; "k_local_build",
 * "k_local_rowprefix", "k_local_flags_rows", "k_local_flags_todo" per 256-frame slab; ...), "k_dyn_cov + k_dyn_eigen",
 * "k_geom_shapes", the order kernels ("k_bonds_tiled", "k_ua_extras", "k_bonds_tiled_maps", ...), "k_map_accumulate",
 * "k_bonds_direct", "k_batch_end" —; *ms is the sum over all segments, i.e. the WHOLE step on the device. */
int gorder_hip_kernel_time(gorder_hip_handle *h, double *ms, uint64_t *launches, int reset);
/* Group `index` of the same measurement (in order of first appearance since the last reset): its name, the device time
 * of its segments and their number.  GORDER_ERR_INVALID_ARGUMENT past the last group.  The times of all groups add up to
 * gorder_hip_kernel_time's *ms.  Call before gorder_hip_kernel_time(..., reset = 1). */
int gorder_hip_kernel_time_group(gorder_hip_handle *h, uint32_t index, const char **name, double *ms, uint64_t *segments);
/* The group names since the last reset joined by " + ", e.g. "k_bonds_tiled + k_batch_end" ("" before the first timed
 * submit); valid until the next submit or reset. */
const char *gorder_hip_kernel_time_names(const gorder_hip_handle *h);

/* Introspection for tests / DESIGN.md: how the bond table was tiled. */
typedef struct {
    uint32_t n_tiles;
    uint32_t block_threads;
    uint32_t max_window_atoms;
    uint32_t n_direct_items;     /* samples that did not fit an LDS window (direct-gather kernel) */
    uint32_t frames_per_stage;
    uint32_t lds_bytes;
    uint32_t map_staged;         /* 1: ordermap samples go through the staging buffer + k_map_accumulate (LDS) */
    uint32_t map_lds_bytes;      /* packed map of one accumulator slot (x2 with leaflets) = k_map_accumulate's LDS */
    uint32_t leaflets_one_read;  /* 1: global leaflets can be assigned from the order kernel's own read of the frame (the
                                  * membrane group is one range of atoms the tiles' windows cover, every head in it) */
} gorder_hip_plan_t;
int gorder_hip_plan(const gorder_hip_handle *h, gorder_hip_plan_t *plan);
/* Same, without touching a device (host logic only); *selfcheck = 0 when every sample of the
 * tables is covered exactly once by the plan. */
int gorder_hip_plan_tables(const gorder_tables_t *tables, gorder_hip_plan_t *plan, int *selfcheck);

/* Global leaflets assigned every frame, nothing but the order parameters asked for: from the second batch on the library
 * reads every frame ONCE — the order kernel routes each molecule by the last assignment and sums the membrane group's
 * normal coordinate on the way, the exact centres follow from the sums, and the few (frame, molecule) pairs whose side
 * was mispredicted are moved afterwards (GORDER_HIP_NO_SPECULATE=1 keeps the two-kernel path; so does a
 * membrane group that is not one range of atoms the tiles can cover).  Results are the two-kernel path's.
 * out[0] = batches that ran this way, out[1] = mispredicted (frame, molecule) pairs moved, out[2] = frames whose centre
 * the sums could not vouch for (they took the exact kernel), out[3] = 1 while the handle still speculates (it stops
 * when a batch leaves more than 1/8 of its frames to the exact kernel or mispredicts more than 1/16 of its pairs).
 * Waits for the handle's stream. */
int gorder_hip_speculation_stats(gorder_hip_handle *h, uint64_t out[4]);

/* Local leaflets in a periodic box: a kernel of its own (k_local_decide) first tries every head against a bound on what the
 * atoms in the ring of cells the cylinder cuts can change about its side — from per-cell sums made without sorting the
 * atoms (k_local_sums) —, and a frame whose heads it all decides skips the cell list and the pass that looks at those
 * atoms.  A submit that finds the majority of its frames left open — a membrane
 * that undulates by more than the water around it allows — sends the next 16 submits down the atom-by-atom pass alone.
 * The sides are the reference's either way.  (Device memory: a handle with local leaflets holds up to 4 GiB of cell-list
 * scratch — 2 048 assignment frames a launch group, fewer for membranes that need more than 2 MB a frame;
 * GORDER_HIP_LOCAL_SLAB=n caps the frames.)  out[0] = submits that ran the bound kernel, out[1] = submits that paused it,
 * out[2] / out[3] = frames left open / frames seen in the last report read back.  (GORDER_HIP_LOCAL_NO_DECIDE=1: never.)
 * Waits for the handle's stream. */
int gorder_hip_local_decide_stats(gorder_hip_handle *h, uint64_t out[4]);

/* Diagnostic: the kernels replace the IEEE division and square root by their Newton cores inside guarded operand
 * ranges (gm_div_core / gm_sqrt_core in gm_math.h; DESIGN.md, K1 arithmetic).  This runs both forms on `n` pseudo-random
 * operand sets on the device — divisors and radicands log-uniform over the guarded range [2^-40, 2^40], numerators of
 * either sign from 2^-100 to 2^60 and exact zeros — and returns how many results differ in any bit:
 * mismatches[0] division, mismatches[1] square root and the acos kernel with the cores inside against the same kernel
 * with IEEE operations (arguments over all of [-1, 1]).  Both must be 0. */
int gorder_hip_selftest_arithmetic(int device, uint64_t n, uint64_t seed, uint64_t mismatches[2]);
/* Diagnostic: the device's acos (fn 0; 1: with the division / square-root cores inside), cos (2) and sin (3) — restatements
 * of glibc's acosf / cosf / sinf algorithms, gm_math.h — of the `n` floats with bit patterns first_bits + i * stride, into
 * the HOST array `out`: to be compared with the host's libm (tests/test_parity_gpu.py does, over all of [-1, 1] and
 * [0, pi] in strides). */
int gorder_hip_selftest_trig(int device, int fn, uint32_t first_bits, uint32_t stride, uint32_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* GORDER_HIP_H */
