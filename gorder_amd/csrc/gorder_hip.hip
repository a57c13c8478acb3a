// gorder_hip.hip — MI355X (gfx950) lipid-order engine: host side + C ABI; the kernels live in kernels_*.h.
//
// Path accelerated (reference file:line under /root/reference/src/analysis):
//   analyze_frame                      common.rs:201-235
//   MoleculeTypes::analyze_frame       topology/molecule.rs:54-95
//   BondType::analyze_frame            topology/bond.rs:396-446      -> k_bonds_tiled / k_bonds_direct
//   BondLike::add_order                topology/bond.rs:184-215         (register accumulators)
//   calc_sch / vector_to               mod.rs:78-82, pbc.rs:378-385     (gm_math.h)
//   OrderValue / AnalysisOrder         order.rs:13-66, 178-188          (i64 ticks, u64 counts)
//   check_box                          common.rs:186-198             -> k_batch_end (k_check_box for a priming frame)
//   SystemLeafletClassification::run   leaflets.rs:171-205           -> k_leaflets_global
//   common_identify_leaflet            leaflets.rs:711-732              (same kernel)
//   IndividualClassification           leaflets.rs:777-801           -> k_leaflets_individual
//   LocalClassification + local centres leaflets.rs:661-675, pbc.rs:273-318 -> k_local_{build,rowprefix,decide,flags_rows,flags_todo} (k_local_{bin,scan,scatter,flags}: very large membranes, no box)
//   should_assign / get_assigned       leaflets.rs:435-441, 1437-1472   (host: assignment-row table)
//   SystemTopology::add / reduce       topology/mod.rs:236-272          (integer sums: order-free)
//
// Built for gfx950 only.  No CPU fallback: without a device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <deque>
#include <vector>

#include "../../include/gorder_hip.h"
#include "gm_math.h"
#include "plan.h"

#pragma clang fp contract(off)

using gorder::DirectItem;
using gorder::Item;
using gorder::kBlock;
using gorder::Plan;
using gorder::Tile;

#include "kernels_common.h"
#include "kernels_bonds.h"
#include "kernels_extras.h"
#include "kernels_leaflets.h"
#include "kernels_normals.h"
#include "kernels_xtc.h"

// ============================================================================================
// host side
// ============================================================================================

struct gorder_hip_handle {
    // staging buffers of gorder_hip_run_trajectory, kept between calls (trajectory_driver.h: TrajCache)
    void *traj_cache = nullptr;
    void (*traj_cache_free)(gorder_hip_handle *) = nullptr;
    Plan plan;
    gorder_tables_t tables{};           // scalar copy (pointers inside are NOT kept)
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // device tables
    Tile *d_tiles = nullptr;
    Item *d_items = nullptr;
    uint32_t *d_tile_slots = nullptr;
    DirectItem *d_direct = nullptr;
    Tile *d_ua_tiles = nullptr;
    gorder::UaItem *d_ua_items = nullptr;
    uint32_t *d_ua_tile_slots = nullptr;
    // ordermaps [3][n_acc][nx*ny] and timewise rows [cap][3][n_acc]
    uint32_t map_nx = 0, map_ny = 0;
    unsigned long long *d_map_sums = nullptr, *d_map_cnts = nullptr;   // [3][n_acc][nx*ny], folded
    unsigned long long *d_map_packed = nullptr;   // [1 or 2][n_acc][nx*ny], what the kernels add into
    unsigned long long *d_map_rec = nullptr;      // united-atom staging for k_map_accumulate (see ExtraArgs::map_rec)
    size_t map_rec_cap = 0;
    gorder::MapRun *d_ua_runs = nullptr, *d_runs = nullptr;
    uint32_t *d_ua_run_begin = nullptr, *d_run_begin = nullptr;
    Item *d_items_by_slot = nullptr;
    uint32_t *d_ua_item_run = nullptr, *d_item_run = nullptr;
    uint4 *d_lgrid = nullptr;       // per slab frame: the cell grid of the local-leaflet kernels
    LocalRowPre *d_lrowpre = nullptr;   // per slab frame: prefix sums along the rows of cells (k_local_rowprefix)
    LocalEdge *d_ledge = nullptr;       // the same cells' 16-byte entries for the bound of k_local_flags_rows
    float4 *d_lfinfo = nullptr;     // per slab frame: extrema of the membrane's normal coordinate, finite flag
    // k_local_decide pays when it decides whole frames; a membrane whose heads it leaves open (one that undulates by more than
    // the water around it allows) is better off without it: every submit that runs it reports {frames left open, frames}
    // to pinned memory, and a submit that finds the majority open sends the next kDecidePause submits down the rows kernel
    // alone.  (Either way the sides are the reference's: the choice is about time only.)
    static constexpr uint32_t kDecidePause = 16;
    uint32_t *d_lsummary = nullptr, *h_lsummary = nullptr;
    hipEvent_t lsummary_written = nullptr;
    bool lsummary_pending = false;
    uint32_t decide_pause = 0;
    uint64_t decide_submits = 0, decide_paused_submits = 0;
    uint32_t *d_lneed = nullptr;    // [local_slab + 1] heads of each slab frame k_local_decide left to k_local_flags_rows; any of the slab
    uint2 *d_ltodo = nullptr;       // {count}, then the (slab frame, head) pairs left to the general passes
    size_t map_lds_bytes = 0;
    bool map_staged = false;       // the packed map of one slot fits LDS: stage + accumulate instead of one atomic per sample
    uint64_t map_pending = 0;      // upper bound of the samples one packed word may hold since the last fold
    uint64_t map_fold_limit = kMapFoldLimit;   // GORDER_HIP_MAP_FOLD_LIMIT lowers it (tests)
    uint32_t map_max_mol = 1;      // most molecules of one type = most samples per tile and frame
    unsigned long long *d_tw_sums = nullptr, *d_tw_cnts = nullptr;
    uint64_t tw_cap = 0;
    ExtraArgs extra{};
    uint32_t *d_geom_group = nullptr;
    float *d_shapes = nullptr;
    float *d_inv_box = nullptr;      // GORDER_FLAG_UA_FAST_NORMALISE: 1 / box edge per frame of the batch
    size_t inv_box_cap = 0;
    // dynamic membrane normals: cloud + per-molecule heads, cell-list scratch (dyn_slab frames), normals of the batch
    uint32_t dyn_slab = 4, local_slab = 4;
    bool local_halo = false;           // local leaflets: rows of cells with a halo + prefix sums (k_local_flags_rows)
    size_t local_rec_stride = 0;       // records per slab frame the local-leaflet record arrays hold
    XtcCheckpoint *d_xtc_cp = nullptr; // gorder_hip_xtc_decode: where the chunks of the frames start (k_xtc_scan -> k_xtc_chunks)
    size_t xtc_cp_cap = 0;
    bool dyn = false;
    uint32_t *d_dyn_cloud = nullptr, *d_dyn_heads = nullptr;
    uint32_t *d_dyn_cell_of = nullptr, *d_dyn_count = nullptr;
    float *d_dyn_rec = nullptr;
    float4 *d_dyn_normals = nullptr;
    DynCov *d_dyn_cov = nullptr;       // per (slab frame, molecule): the sums of k_dyn_cov for k_dyn_eigen
    size_t dyn_normals_cap = 0;
    std::vector<float> last_normals;   // [n_mol_total][4] of the last submitted frame
    // manual membrane normals (ManualMembraneNormal, normal.rs:266-300): handed over per batch by the host
    std::vector<float> manual_normals; // [frames][n_mol_total][4] (nx, ny, nz, 3) for the NEXT submit
    uint32_t manual_frames = 0;
    bool manual_active = false;        // this batch's samples read d_dyn_normals filled from manual_normals
    size_t shapes_cap = 0;
    uint32_t *d_err = nullptr;
    unsigned long long *d_acc = nullptr;   // [4][n_acc] + total_frames
    unsigned long long *d_rep = nullptr;   // [n_rep][4][n_acc] (see k_fold_replicas)
    uint32_t n_rep = 32;
    bool rep_dirty = false;
    bool acc_external = false;
    size_t acc_words = 0;
    // leaflets
    uint32_t *d_heads = nullptr, *d_membrane = nullptr, *d_methyl_begin = nullptr, *d_methyl_atoms = nullptr;
    uint8_t *d_aflags = nullptr;
    size_t aflags_rows = 0;
    float *d_adist = nullptr;
    // Local leaflets scratch (sized for local_slab assignment frames)
    uint32_t *d_lcell_of = nullptr, *d_lcell_count = nullptr, *d_lcell_fill = nullptr;
    float *d_ltrig = nullptr;
    uint32_t *d_arow = nullptr, *d_aframes = nullptr;
    // what those two hold right now: equal batches (same length, same assignment pattern) skip the upload
    std::vector<uint32_t> up_arow, up_aframes;
    const uint32_t *up_arow_at = nullptr, *up_aframes_at = nullptr;
    size_t arow_cap = 0, aframes_cap = 0;
    bool have_assignment = false;
    uint64_t assignment_frame = 0;
    // one read for global leaflets + order parameters (Plan::spec_ok; k_bonds_tiled<..., MOM> + k_spec_*)
    bool spec_enabled = false;         // the plan allows it and GORDER_HIP_NO_SPECULATE is not set
    bool spec_now = false;             // this batch's order kernel is the MOM variant
    uint2 *d_own = nullptr;
    float4 *d_mom = nullptr;
    size_t mom_cap = 0;
    float *d_head_z = nullptr;
    size_t head_z_cap = 0;
    uint32_t *d_own_head_begin = nullptr;
    uint2 *d_own_heads = nullptr;
    float *d_spec_center = nullptr;
    uint8_t *d_spec_ok = nullptr;
    size_t spec_frames_cap = 0;
    uint32_t *d_spec_counters = nullptr;           // two pairs (batches alternate): [0] mispredicted (frame, molecule) pairs, [1] frames left to the exact kernel
    // pinned copies of the last speculative batches' counters, looked at (without waiting) when a later batch is submitted
    static constexpr uint32_t kSpecRing = 8;
    uint32_t *h_spec_counters = nullptr;           // [kSpecRing][2]
    hipEvent_t spec_counters_copied[kSpecRing] = {};
    uint64_t spec_ring_frames[kSpecRing] = {};     // frames of the batch in that slot (0: nothing pending)
    uint32_t *d_spec_mol_begin = nullptr;
    SpecSample *d_spec_samples = nullptr;
    uint64_t spec_batches = 0, spec_fixed = 0, spec_exact_frames = 0;   // statistics (gorder_hip_speculation_stats)
    // host staging for submit_host: two device buffers, filled on a copy stream while the kernels of the other one run
    float *d_stage_xyz[2] = {nullptr, nullptr}, *d_stage_box[2] = {nullptr, nullptr};
    size_t stage_xyz_cap[2] = {0, 0}, stage_box_cap[2] = {0, 0};
    hipEvent_t stage_copied[2] = {nullptr, nullptr}, stage_computed[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    int stage_next = 0;
    hipStream_t copy_stream = nullptr;
    float n2 = 1.0f, n2sq = 1.0f;
    int axis = -1;   // 0/1/2 when the static normal is exactly that unit axis (kernel specialisation)
    int frames_per_stage = kFramesPerStage;   // G = 4 frames staged per pass (8 was measured at 26 % and spilled: removed in round 4)
    bool membrane_is_frame = false;            // the membrane group is every atom of the frame, in order
    bool use_gather = false;                   // GORDER_HIP_KERNEL=gather: L1-gather kernel instead of LDS staging
    uint32_t wg_capacity = 256u * 6u;          // co-resident workgroups of the tiled kernel on this device
    uint32_t wg_target = 0;                    // GORDER_HIP_WG_TARGET: force the workgroup count aimed at
    uint32_t lw = 0;
    size_t lds_bytes = 0;
    uint64_t n_frames = 0;
    uint64_t err_index = 0;
    std::string err_msg;
    // kernel timing (gorder_hip_kernel_time / _group): OFF until the host asks for it once.  Everything a handle queues
    // runs in order on ONE stream, so a submit is a chain of labelled segments between events E0 [leaflet kernels] E1
    // [order kernels] E2 ... En: timing_mark(label) closes the open segment with a new event and opens the next.  The
    // events are a fixed ring, drained (oldest segment first) when it runs low — nothing grows with the trajectory.
    static constexpr uint32_t kTimingEvents = 1024;
    struct TimingSeg { uint32_t label, ev_a, ev_b; };
    bool timing_on = false;
    hipEvent_t timing_ev[kTimingEvents] = {};
    uint32_t timing_next_ev = 0;                  // ring position of the next event to record
    int timing_open_label = -1;                   // the segment being queued (-1: none)
    uint32_t timing_open_ev = 0;
    std::deque<TimingSeg> timing_pending;         // closed segments whose events have not been read yet
    std::vector<std::string> timing_labels;       // group names in order of first appearance since the last reset
    std::vector<double> timing_label_ms;
    std::vector<uint64_t> timing_label_n;
    uint64_t timing_launches = 0;                 // submits (chains) since the last reset
    std::string timed_kernels;                    // the labels joined by " + " (gorder_hip_kernel_time_names)
    // host copies behind the payload of gorder_hip_last_error_index
    std::vector<uint32_t> host_heads, host_dyn_heads;
    uint32_t *d_mol_slot0 = nullptr;
    uint64_t err_frame = 0;
    // frame indices of the batches submitted since the last clean synchronisation (ordinal -> SystemTopology::frame of
    // its frames), so that a device error names the frame of the TRAJECTORY; arithmetic progressions are kept as three
    // numbers, at most kBatchLog batches are remembered
    struct BatchLog { uint64_t ordinal, first, stride; std::vector<uint64_t> list; };
    static constexpr size_t kBatchLog = 4096;
    std::deque<BatchLog> batch_log;
    uint64_t n_submits = 0;
    const unsigned long long *decoder_key = nullptr;   // gorder_hip_run_trajectory: the slot's decoder record, for the next submit only
};

namespace {

int fail(gorder_hip_handle *h, int status, const std::string &msg) {
    if (h) h->err_msg = msg;
    return status;
}

#define HIP_TRY(h, expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(h, GORDER_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int upload(gorder_hip_handle *h, T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    if (src.empty()) return GORDER_OK;
    HIP_TRY(h, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GORDER_OK;
}

template <typename T>
int ensure(gorder_hip_handle *h, T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return GORDER_OK;
    if (*buf) HIP_TRY(h, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    const size_t n = need + need / 2;
    HIP_TRY(h, hipMalloc((void **)buf, n * sizeof(T)));
    *cap = n;
    return GORDER_OK;
}

bool should_assign(uint32_t frequency, uint64_t frame) {   // leaflets.rs:435-441
    return frequency == 0 ? frame == 0 : (frame % frequency) == 0;
}

// (slot, molecule, which atom) of an order sample -> atom index in the frame (the payload of UndefinedPosition,
// errors.rs:136).  Error path only: a linear walk over the plan.
uint32_t sample_atom(const Plan &p, uint32_t slot, uint32_t mol, uint32_t which) {
    for (const Tile &t : p.tiles)
        for (uint32_t q = 0; q < t.n_items; q++) {
            const Item &it = p.items[t.item0 + q];
            if (it.mol == mol && p.tile_slots[t.slot0 + it.lslot] == slot) return t.atom0 + (which ? it.lj : it.li);
        }
    for (const DirectItem &d : p.direct)
        if (d.mol == mol && d.slot == slot) return which ? d.j : d.i;
    for (const Tile &t : p.ua_tiles)
        for (uint32_t q = 0; q < t.n_items; q++) {
            const gorder::UaItem &it = p.ua_items[t.item0 + q];
            if (it.mol == mol && p.ua_tile_slots[t.slot0 + it.lslot0] == slot) return t.atom0 + it.l[which & 3u];
        }
    return 0;
}

// Decode the device's error key (kernels_common.h, raise_error): status, frame, payload.
int check_device_error(gorder_hip_handle *h) {
    unsigned long long rec[3] = {kErrNone, kErrNone, 0};
    HIP_TRY(h, hipMemcpy(rec, h->d_err, sizeof(rec), hipMemcpyDeviceToHost));
    // [1]: latched at the end of a batch; [0]: raised outside a batch (priming frame, stand-alone decode) and not committed yet
    const bool latched = rec[1] != kErrNone;
    const unsigned long long key = latched ? rec[1] : rec[0];
    if (key == kErrNone) { h->batch_log.clear(); return GORDER_OK; }
    const uint32_t code = (uint32_t)(key & 15u), detail = (uint32_t)(key >> 4) & 3u, mol = (uint32_t)(key >> 6) & 0x1ffffu;
    const uint32_t sample = (uint32_t)(key >> 23) & 1u, slot = (uint32_t)(key >> 24) & 0x3fffu;
    const uint32_t stage = (uint32_t)(key >> 38) & 3u, frame = (uint32_t)(key >> 40) & 0x7fffffu;
    const int status = code == 8u ? (int)GORDER_ERR_BOX_RANGE : (code == 9u ? (int)GORDER_ERR_TRAJECTORY_FORMAT : (int)code);
    h->err_frame = frame;
    char where[96];
    snprintf(where, sizeof(where), "frame %u of a batch", frame);
    if (latched)
        for (const auto &b : h->batch_log)
            if (b.ordinal == rec[2]) {
                if (b.list.empty()) h->err_frame = b.first + (uint64_t)frame * b.stride;
                else if (frame < b.list.size()) h->err_frame = b.list[frame];
                snprintf(where, sizeof(where), "frame %llu of the trajectory (frame %u of batch %llu)",
                         (unsigned long long)h->err_frame, frame, (unsigned long long)rec[2]);
                break;
            }
    h->err_index = 0;
    if (status == GORDER_ERR_UNDEFINED_POSITION) {
        if (stage == kStageTypes && sample) h->err_index = sample_atom(h->plan, slot, mol, detail);
        else if (mol < h->host_dyn_heads.size()) h->err_index = h->host_dyn_heads[mol];   // head of a dynamic-normal cloud
    } else if (status == GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER) {
        if (mol < h->host_heads.size()) h->err_index = h->host_heads[mol];                // leaflets.rs:661-675: the head's index
    } else if (status == GORDER_ERR_DYNAMIC_NORMAL) {
        h->err_index = detail;                                                             // NotEnoughPoints(n)
    }
    char buf[200];
    snprintf(buf, sizeof(buf), "device raised %s (payload %llu) in %s", gorder_hip_strerror(status),
             (unsigned long long)h->err_index, where);
    h->err_msg = buf;
    return status;
}

// ---- kernel timing: labelled segments between events on the handle's stream (see gorder_hip_handle::timing_ev) -----
int timing_drain_oldest(gorder_hip_handle *h) {
    const gorder_hip_handle::TimingSeg sg = h->timing_pending.front();
    float t = 0.0f;
    HIP_TRY(h, hipEventSynchronize(h->timing_ev[sg.ev_b]));
    HIP_TRY(h, hipEventElapsedTime(&t, h->timing_ev[sg.ev_a], h->timing_ev[sg.ev_b]));
    h->timing_label_ms[sg.label] += t;
    h->timing_label_n[sg.label] += 1;
    h->timing_pending.pop_front();
    return GORDER_OK;
}
// Close the open segment (if any) at a new event and, with a label, open the next one there.  No-op while timing is off.
int timing_mark(gorder_hip_handle *h, const char *label) {
    if (!h->timing_on) return GORDER_OK;
    if (!label && h->timing_open_label < 0) return GORDER_OK;
    // a pending segment holds at most two events of the ring: keep fewer than half of it pending
    while (h->timing_pending.size() >= gorder_hip_handle::kTimingEvents / 2u - 2u) {
        const int st = timing_drain_oldest(h);
        if (st != GORDER_OK) return st;
    }
    const uint32_t ev = h->timing_next_ev;
    h->timing_next_ev = (ev + 1u) % gorder_hip_handle::kTimingEvents;
    if (!h->timing_ev[ev]) HIP_TRY(h, hipEventCreate(&h->timing_ev[ev]));
    HIP_TRY(h, hipEventRecord(h->timing_ev[ev], h->stream));
    if (h->timing_open_label >= 0) h->timing_pending.push_back({(uint32_t)h->timing_open_label, h->timing_open_ev, ev});
    h->timing_open_label = -1;
    if (label) {
        size_t k = 0;
        while (k < h->timing_labels.size() && h->timing_labels[k] != label) k++;
        if (k == h->timing_labels.size()) {
            h->timing_labels.push_back(label);
            h->timing_label_ms.push_back(0.0);
            h->timing_label_n.push_back(0);
            if (!h->timed_kernels.empty()) h->timed_kernels += " + ";
            h->timed_kernels += label;
        }
        h->timing_open_label = (int)k;
        h->timing_open_ev = ev;
    }
    return GORDER_OK;
}
#define TIMING_MARK(h, label)                                   \
    do {                                                        \
        const int tm_ = timing_mark(h, label);                  \
        if (tm_ != GORDER_OK) return tm_;                       \
    } while (0)

bool env_flag(const char *name) {
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
}

// Launch the order kernels of one batch (internal).
// fold the replicas into the accumulator block (stream-ordered; cheap: 4 * n_acc threads)
int fold_replicas(gorder_hip_handle *h) {
    if (!h->rep_dirty || !h->d_rep) return GORDER_OK;
    const uint32_t n = 4u * h->plan.n_acc;
    hipLaunchKernelGGL(k_fold_replicas, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_acc, h->d_rep, h->n_rep, n);
    HIP_TRY(h, hipGetLastError());
    h->rep_dirty = false;
    return GORDER_OK;
}

// unpack the ordermap words the scatter kernels filled since the last fold (k_fold_maps)
int fold_maps(gorder_hip_handle *h) {
    if (!h->extra.maps || !h->map_pending) return GORDER_OK;
    const size_t n = (size_t)h->plan.n_acc * h->map_nx * h->map_ny;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fold_maps, dim3(blocks), dim3(256), 0, h->stream, h->d_map_packed, h->d_map_sums, h->d_map_cnts,
                       n, h->tables.leaflets.method != GORDER_LEAFLETS_NONE ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
    h->map_pending = 0;
    return GORDER_OK;
}

// the cell list of `ns` slab frames (k_local_build, or bin / scan / scatter for very large atom sets); `counts` are the
// lo.cell_count and lo.cell_fill words the three-kernel form needs zeroed
int launch_cell_list(gorder_hip_handle *h, const LocalArgs &lo, uint32_t ns, uint32_t n_list, void *counts, size_t count_bytes) {
    const bool three = env_flag("GORDER_HIP_LOCAL_THREE_KERNELS");   // A/B switch; also what membranes beyond kLocalBuildMax take
    TIMING_MARK(h, n_list <= kLocalBuildMax && !three ? "k_local_build" : "k_local_bin + k_local_scan + k_local_scatter");
    if (n_list <= kLocalBuildMax && !three) {
        // (a thread keeps a byte per atom it places: the kernel compiled for 2, 5 or 8 trips of eight atoms per thread)
        if (n_list <= 2u * 8192u) hipLaunchKernelGGL(k_local_build<2>, dim3(ns), dim3(1024), kLocalBuildLds, h->stream, lo);
        else if (n_list <= 5u * 8192u) hipLaunchKernelGGL(k_local_build<5>, dim3(ns), dim3(1024), kLocalBuildLds, h->stream, lo);
        else hipLaunchKernelGGL(k_local_build<kLocalBuildTrips>, dim3(ns), dim3(1024), kLocalBuildLds, h->stream, lo);
        return GORDER_OK;
    }
    HIP_TRY(h, hipMemsetAsync(counts, 0, count_bytes, h->stream));
    const dim3 ga((n_list + 255) / 256, ns);
    hipLaunchKernelGGL(k_local_bin, ga, dim3(256), 0, h->stream, lo);
    hipLaunchKernelGGL(k_local_scan, dim3(ns), dim3(1024), 0, h->stream, lo);
    hipLaunchKernelGGL(k_local_scatter, ga, dim3(256), 0, h->stream, lo);
    return GORDER_OK;
}

// normals of every molecule for the frames of this batch -> h->d_dyn_normals [n_frames][n_mol_total]
int run_dynamic_normals(gorder_hip_handle *h, const FrameArgs &a) {
    int st;
    const uint32_t n_mol = h->plan.n_mol_total;
    if ((st = ensure(h, &h->d_dyn_normals, &h->dyn_normals_cap, (size_t)a.n_frames * n_mol)) != GORDER_OK) return st;
    const gorder_dynamic_normal_t &dn = h->tables.dynamic_normal;
    const size_t ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
    LocalArgs lo{};
    lo.xyz = a.xyz; lo.box9 = a.box9; lo.n_atoms = a.n_atoms;
    lo.n_mol_total = n_mol; lo.heads = h->d_dyn_heads; lo.membrane = h->d_dyn_cloud; lo.n_membrane = dn.n_cloud;
    lo.dim = 2; lo.pbc = a.pbc; lo.radius = dn.radius; lo.radius_thr = local_radius_threshold(dn.radius);
    lo.halo = 0; lo.rec_stride = dn.n_cloud;
    lo.cell_of = h->d_dyn_cell_of; lo.trig = h->d_dyn_rec;
    lo.cell_count = h->d_dyn_count; lo.cell_fill = h->d_dyn_count + h->dyn_slab * (ncell + 1);
    lo.err = h->d_err; lo.aframes = nullptr; lo.write_dist_frame = -1;
    for (uint32_t done = 0; done < a.n_frames; done += h->dyn_slab) {
        const uint32_t ns = std::min(a.n_frames - done, h->dyn_slab);
        lo.frame0 = done;
        lo.n_slab = ns;
        if ((st = launch_cell_list(h, lo, ns, dn.n_cloud, h->d_dyn_count,
                                   (size_t)h->dyn_slab * (2 * ncell + 1) * sizeof(uint32_t))) != GORDER_OK) return st;
        TIMING_MARK(h, "k_dyn_cov + k_dyn_eigen");
        hipLaunchKernelGGL(k_dyn_cov, dim3((n_mol + 15) / 16, ns), dim3(256), 0, h->stream, lo, h->d_dyn_cov);
        hipLaunchKernelGGL(k_dyn_eigen, dim3((uint32_t)(((size_t)ns * n_mol + 255) / 256)), dim3(256), 0, h->stream, lo, h->d_dyn_cov,
                           h->d_dyn_normals);
    }
    HIP_TRY(h, hipGetLastError());
    // keep the last frame's normals for gorder_hip_normals (stream-ordered copy into pageable memory)
    h->last_normals.resize(4 * (size_t)n_mol);
    HIP_TRY(h, hipMemcpyAsync(h->last_normals.data(), h->d_dyn_normals + (size_t)(a.n_frames - 1) * n_mol,
                              4 * sizeof(float) * (size_t)n_mol, hipMemcpyDeviceToHost, h->stream));
    return GORDER_OK;
}

int launch_orders(gorder_hip_handle *h, FrameArgs &a) {
    const Plan &p = h->plan;
    const uint32_t n_tiles = (uint32_t)p.tiles.size();
    const bool extras = h->extra.maps || h->extra.tw || h->extra.geom_kind || h->dyn || h->manual_active;
    // (every kernel group of the batch opens a timing segment of its own: gorder_hip_kernel_time_group)
#define name(k) TIMING_MARK(h, k)
    if (h->dyn && !h->manual_active) {
        const int st2 = run_dynamic_normals(h, a);
        if (st2 != GORDER_OK) return st2;
    }
    if (h->extra.geom_kind) {
        int st2;
        if ((st2 = ensure(h, &h->d_shapes, &h->shapes_cap, (size_t)a.n_frames * 8)) != GORDER_OK) return st2;
        const gorder_geometry_t &ge = h->tables.geometry;
        GeomArgs ga{};
        ga.xyz = a.xyz; ga.box9 = a.box9; ga.n_atoms = a.n_atoms; ga.pbc = a.pbc;
        ga.kind = ge.kind; ga.reference = ge.reference; ga.orientation = ge.orientation;
        for (int d = 0; d < 3; d++) { ga.point[d] = ge.point[d]; ga.structure_box[d] = ge.structure_box[d]; }
        for (int d = 0; d < 2; d++) { ga.xdim[d] = ge.xdim[d]; ga.ydim[d] = ge.ydim[d]; ga.zdim[d] = ge.zdim[d]; ga.span[d] = ge.span[d]; }
        ga.radius = ge.radius; ga.group = h->d_geom_group; ga.n_group = ge.n_group;
        ga.shapes = h->d_shapes; ga.err = h->d_err;
        name("k_geom_shapes");
        hipLaunchKernelGGL(k_geom_shapes, dim3(a.n_frames), dim3(256), 0, h->stream, ga);
        HIP_TRY(h, hipGetLastError());
    }
    if (n_tiles && !extras) {
        // enough workgroups to fill 256 CUs x 8 blocks, frames split into chunks of whole stages
        // Cut the frame range into chunks of whole stages.  All workgroups do the same amount of work,
        // so the grid should be a whole number of co-resident rounds: exactly one round when the tiles
        // fit, otherwise many short rounds so that the last, partial one costs little.
        const uint32_t G = (uint32_t)h->frames_per_stage;
        const uint32_t n_stages = (a.n_frames + G - 1) / G;
        uint32_t target = h->wg_target ? h->wg_target : h->wg_capacity;
        if (!h->wg_target) target = 12u * h->wg_capacity;   // measured: ~8-12 short rounds beat one long round
        uint32_t n_chunks = std::max(1u, target / n_tiles);
        n_chunks = std::min(n_chunks, std::max(1u, n_stages / 4u));   // >= 4 stages per workgroup
        uint32_t fpc = ((n_stages + n_chunks - 1) / n_chunks) * G;
        n_chunks = (a.n_frames + fpc - 1) / fpc;
        a.frames_per_chunk = fpc;
        const uint64_t grid = (uint64_t)n_tiles * n_chunks;
        if (grid > 0x7fffffffull) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "batch too large");
        const bool ac = (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        const dim3 g((uint32_t)grid), b(kBlock);
#define GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, AX_)                                               \
        do {                                                                                                \
            if (LF_ && h->spec_now)                                                                         \
                hipLaunchKernelGGL((k_bonds_tiled<G_, NPF_, AC_, PBC_, LF_, AX_, LF_>), g, b, h->lds_bytes, h->stream, a, \
                                   a.xyz, a.box9, a.aflags, a.arow, h->d_tiles, h->d_items, h->d_tile_slots, n_tiles, h->lw); \
            else                                                                                            \
                hipLaunchKernelGGL((k_bonds_tiled<G_, NPF_, AC_, PBC_, LF_, AX_>), g, b, h->lds_bytes, h->stream, a,     \
                                   a.xyz, a.box9, a.aflags, a.arow, h->d_tiles, h->d_items, h->d_tile_slots, n_tiles,   \
                                   h->lw);                                                                  \
        } while (0)
#define GORDER_LAUNCH_TILED_V(G_, NPF_, AC_, PBC_, LF_)                                                    \
        do {                                                                                                \
            if (h->axis == 2) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 2);                           \
            else if (h->axis == 1) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 1);                      \
            else if (h->axis == 0) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 0);                      \
            else GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, -1);                                       \
        } while (0)
#define GORDER_LAUNCH_TILED(G_, NPF_)                                                                       \
        do {                                                                                                \
            const int v_ = (ac ? 4 : 0) | (a.pbc ? 2 : 0) | (a.leaflets ? 1 : 0);                           \
            switch (v_) {                                                                                   \
                case 0: GORDER_LAUNCH_TILED_V(G_, NPF_, false, false, false); break;                        \
                case 1: GORDER_LAUNCH_TILED_V(G_, NPF_, false, false, true); break;                         \
                case 2: GORDER_LAUNCH_TILED_V(G_, NPF_, false, true, false); break;                         \
                case 3: GORDER_LAUNCH_TILED_V(G_, NPF_, false, true, true); break;                          \
                case 4: GORDER_LAUNCH_TILED_V(G_, NPF_, true, false, false); break;                         \
                case 5: GORDER_LAUNCH_TILED_V(G_, NPF_, true, false, true); break;                          \
                case 6: GORDER_LAUNCH_TILED_V(G_, NPF_, true, true, false); break;                          \
                default: GORDER_LAUNCH_TILED_V(G_, NPF_, true, true, true); break;                          \
            }                                                                                               \
        } while (0)
#define GORDER_LAUNCH_GATHER_V(G_, AC_, PBC_, LF_)                                                         \
        hipLaunchKernelGGL((k_bonds_gather<G_, AC_, PBC_, LF_>), g, b, 0, h->stream, a, a.xyz, a.box9, a.aflags, \
                           a.arow, h->d_tiles, h->d_items, h->d_tile_slots, n_tiles)
#define GORDER_LAUNCH_GATHER(G_)                                                                            \
        do {                                                                                                \
            const int v_ = (ac ? 4 : 0) | (a.pbc ? 2 : 0) | (a.leaflets ? 1 : 0);                           \
            switch (v_) {                                                                                   \
                case 0: GORDER_LAUNCH_GATHER_V(G_, false, false, false); break;                             \
                case 1: GORDER_LAUNCH_GATHER_V(G_, false, false, true); break;                              \
                case 2: GORDER_LAUNCH_GATHER_V(G_, false, true, false); break;                              \
                case 3: GORDER_LAUNCH_GATHER_V(G_, false, true, true); break;                               \
                case 4: GORDER_LAUNCH_GATHER_V(G_, true, false, false); break;                              \
                case 5: GORDER_LAUNCH_GATHER_V(G_, true, false, true); break;                               \
                case 6: GORDER_LAUNCH_GATHER_V(G_, true, true, false); break;                               \
                default: GORDER_LAUNCH_GATHER_V(G_, true, true, true); break;                               \
            }                                                                                               \
        } while (0)
        name(h->use_gather ? "k_bonds_gather" : "k_bonds_tiled");
        if (h->use_gather) {
            GORDER_LAUNCH_GATHER(4);
        } else {
            // prefetch registers per thread: enough float4 for the widest window (64 threads stage a frame)
            if ((3u * p.max_window + 6u) / 4u <= 4u * 64u && !env_flag("GORDER_HIP_NPF5")) GORDER_LAUNCH_TILED(4, 4);
            else GORDER_LAUNCH_TILED(4, 5);
        }
#undef GORDER_LAUNCH_GATHER
#undef GORDER_LAUNCH_GATHER_V
#undef GORDER_LAUNCH_TILED_A
#undef GORDER_LAUNCH_TILED_V
#undef GORDER_LAUNCH_TILED
        HIP_TRY(h, hipGetLastError());
    }
    if (extras || !p.ua_tiles.empty()) {
        // scatter-bound modes: plain per-sample kernels (see "Extras" above)
        ExtraArgs e = h->extra;
        e.tw_sums = h->d_tw_sums; e.tw_cnts = h->d_tw_cnts; e.tw_row0 = h->n_frames;
        e.shapes = h->d_shapes;
        const bool ua_fast = (h->tables.flags & GORDER_FLAG_UA_FAST_NORMALISE) != 0 && !p.ua_tiles.empty();
        e.inv_box = nullptr;
        if (ua_fast && a.pbc) {      // 1 / box edge per frame, once per frame instead of once per lane and frame
            int st3;
            if ((st3 = ensure(h, &h->d_inv_box, &h->inv_box_cap, (size_t)a.n_frames * 8)) != GORDER_OK) return st3;
            hipLaunchKernelGGL(k_inv_box, dim3((4u * a.n_frames + 255u) / 256u), dim3(256), 0, h->stream, a.box9, a.n_frames, h->d_inv_box);
            e.inv_box = h->d_inv_box;
        }
        e.dyn = (h->dyn || h->manual_active) ? h->d_dyn_normals : nullptr;
        const bool ac = (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        // with ordermaps the frames go in sub-ranges short enough for the packed map words (k_fold_maps)
        uint32_t sub = e.maps ? (uint32_t)std::max<uint64_t>(1, (h->map_fold_limit - 1) / h->map_max_mol) : a.n_frames;
        const bool staged = e.maps && h->map_staged;
        const size_t rec_per_frame = std::max(p.ua_tiles.size() * 3u, extras ? p.tiles.size() : (size_t)0) * kBlock;   // staged words per frame
        if (staged) {   // at most 1 GiB of staging per sub-range
            sub = std::min<uint32_t>(sub, (uint32_t)std::max<size_t>(1, ((size_t)1 << 27) / rec_per_frame));
            sub = std::min(sub, a.n_frames);
            const int st2 = ensure(h, &h->d_map_rec, &h->map_rec_cap, rec_per_frame * (((size_t)sub + 15) / 16 * 16));
            if (st2 != GORDER_OK) return st2;
        }
        for (uint32_t lo = 0; lo < a.n_frames; lo += sub) {
            const uint32_t hi = std::min(a.n_frames, lo + sub), nf = hi - lo;
            if (e.maps) {
                const uint64_t cost = (uint64_t)nf * h->map_max_mol;
                if (h->map_pending + cost >= h->map_fold_limit) {
                    const int st = fold_maps(h);
                    if (st != GORDER_OK) return st;
                }
                h->map_pending += cost;
            }
            for (int pass = 0; pass < 2; pass++) {
                const uint32_t nt = pass == 0 ? (extras ? n_tiles : 0u) : (uint32_t)p.ua_tiles.size();
                if (!nt) continue;
                uint32_t n_chunks = std::max(1u, (h->wg_target ? h->wg_target : 8u * h->wg_capacity) / nt);
                n_chunks = std::min(n_chunks, nf);
                uint32_t fpc = (nf + n_chunks - 1) / n_chunks;
                if (staged) fpc = (fpc + 15u) / 16u * 16u;   // whole frame blocks (kRecFrames) and whole lines per workgroup
                // per-frame rows and nothing else, the default cosine, four frames per stage: the tiled kernel with the
                // stage's ticks as a second output (k_bonds_tiled_tw)
                // (with staged ordermaps as well: the same kernel writes the map words too)
                const bool tiled_tw = pass == 0 && e.tw && (!e.maps || staged) && !e.geom_kind && !e.dyn && !ac && !h->use_gather && h->d_item_run &&
                                      h->frames_per_stage == (int)kRecFrames && !env_flag("GORDER_HIP_TW_GATHER");
                if (tiled_tw) fpc = (fpc + kRecFrames - 1u) / kRecFrames * kRecFrames;       // whole stages
                n_chunks = (nf + fpc - 1) / fpc;
                FrameArgs b = a;
                b.frame0 = lo;
                b.n_frames = hi;
                b.frames_per_chunk = fpc;
                e.map_rec = staged ? h->d_map_rec : nullptr;
                e.item_run = pass == 0 ? h->d_item_run : h->d_ua_item_run;      // (bond tiles: k_bonds_tiled_tw only)
                e.rec_frame0 = lo;
                e.rec_stride = (nf + 15u) / 16u * 16u;
                const dim3 g(pass == 0 ? nt * n_chunks : (nt * n_chunks + 7u) / 8u * 8u), blk(kBlock);   // united atoms: see the XCD mapping in k_ua_extras
                if (pass == 0) {
                    const Item *items = (staged || tiled_tw) ? h->d_items_by_slot : h->d_items;
#define GORDER_LAUNCH_BONDS(AC, MO)                                                                               \
    hipLaunchKernelGGL((k_bonds_extras<AC, MO>), g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags, b.arow,       \
                       h->d_tiles, items, h->d_tile_slots, nt)
                    const bool maps_only = staged && !e.tw && !e.geom_kind && !e.dyn;
                    // maps and nothing else, the default cosine, four frames per stage: the tiled kernel with the staged
                    // words as a second output (k_bonds_tiled_maps)
                    const bool tiled_maps = maps_only && !ac && !h->use_gather && h->frames_per_stage == (int)kRecFrames &&
                                            !env_flag("GORDER_HIP_MAPS_GATHER");
#define GORDER_LAUNCH_TM_A(NPF_, PBC_, LF_, AX_)                                                                   \
    do {                                                                                                            \
        if (tiled_tw && !staged && LF_ && h->spec_now)                                                              \
            hipLaunchKernelGGL((k_bonds_tiled_tw<NPF_, PBC_, LF_, AX_, false, LF_>), g, blk, h->lds_bytes, h->stream, b, e, b.xyz, \
                               b.box9, b.aflags, b.arow, h->d_tiles, items, h->d_tile_slots, nt, h->lw);               \
        else if (tiled_tw && staged)                                                                                \
            hipLaunchKernelGGL((k_bonds_tiled_tw<NPF_, PBC_, LF_, AX_, true>), g, blk, h->lds_bytes, h->stream, b, e, b.xyz, \
                               b.box9, b.aflags, b.arow, h->d_tiles, items, h->d_tile_slots, nt, h->lw);               \
        else if (tiled_tw)                                                                                          \
            hipLaunchKernelGGL((k_bonds_tiled_tw<NPF_, PBC_, LF_, AX_, false>), g, blk, h->lds_bytes, h->stream, b, e, b.xyz, \
                               b.box9, b.aflags, b.arow, h->d_tiles, items, h->d_tile_slots, nt, h->lw);               \
        else                                                                                                        \
            hipLaunchKernelGGL((k_bonds_tiled_maps<NPF_, PBC_, LF_, AX_>), g, blk, h->lds_bytes, h->stream, b, e, b.xyz, \
                               b.box9, b.aflags, b.arow, h->d_tiles, items, h->d_tile_slots, nt, h->lw);               \
    } while (0)
#define GORDER_LAUNCH_TM_V(NPF_, PBC_, LF_)                                                                        \
    do {                                                                                                            \
        if (h->axis == 2) GORDER_LAUNCH_TM_A(NPF_, PBC_, LF_, 2);                                                   \
        else if (h->axis == 1) GORDER_LAUNCH_TM_A(NPF_, PBC_, LF_, 1);                                              \
        else if (h->axis == 0) GORDER_LAUNCH_TM_A(NPF_, PBC_, LF_, 0);                                              \
        else GORDER_LAUNCH_TM_A(NPF_, PBC_, LF_, -1);                                                               \
    } while (0)
#define GORDER_LAUNCH_TM(NPF_)                                                                                     \
    do {                                                                                                            \
        switch ((b.pbc ? 2 : 0) | (b.leaflets ? 1 : 0)) {                                                           \
            case 0: GORDER_LAUNCH_TM_V(NPF_, false, false); break;                                                  \
            case 1: GORDER_LAUNCH_TM_V(NPF_, false, true); break;                                                   \
            case 2: GORDER_LAUNCH_TM_V(NPF_, true, false); break;                                                   \
            default: GORDER_LAUNCH_TM_V(NPF_, true, true); break;                                                   \
        }                                                                                                           \
    } while (0)
                    name(tiled_maps ? "k_bonds_tiled_maps" : (tiled_tw ? "k_bonds_tiled_tw" : "k_bonds_extras"));
                    if (tiled_maps || tiled_tw) {
                        if ((3u * p.max_window + 6u) / 4u <= 4u * 64u && !env_flag("GORDER_HIP_NPF5")) GORDER_LAUNCH_TM(4);
                        else GORDER_LAUNCH_TM(5);
                    }
                    else if (maps_only) { if (ac) GORDER_LAUNCH_BONDS(true, true); else GORDER_LAUNCH_BONDS(false, true); }
                    else { if (ac) GORDER_LAUNCH_BONDS(true, false); else GORDER_LAUNCH_BONDS(false, false); }
#undef GORDER_LAUNCH_TM
#undef GORDER_LAUNCH_TM_V
#undef GORDER_LAUNCH_TM_A
#undef GORDER_LAUNCH_BONDS
                } else {
#define GORDER_LAUNCH_UA(AC, MODE)                                                                                \
    do {                                                                                                            \
        if (!AC && ua_fast)                                                                                         \
            hipLaunchKernelGGL((k_ua_extras_fast<MODE>), g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags,      \
                               b.arow, h->d_ua_tiles, h->d_ua_items, h->d_ua_tile_slots, nt, (const float *)e.inv_box); \
        else                                                                                                        \
            hipLaunchKernelGGL((k_ua_extras<AC, MODE>), g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags,        \
                               b.arow, h->d_ua_tiles, h->d_ua_items, h->d_ua_tile_slots, nt, (const float *)nullptr); \
    } while (0)
                    // staged ordermap samples and nothing else: the lean kernel
                    const bool maps_only = extras && staged && !e.tw && !e.geom_kind && !e.dyn;
                    name("k_ua_extras");
                    if (maps_only) { if (ac) GORDER_LAUNCH_UA(true, 1); else GORDER_LAUNCH_UA(false, 1); }
                    else if (extras && e.tw && !e.maps && !e.geom_kind && !e.dyn) { if (ac) GORDER_LAUNCH_UA(true, 3); else GORDER_LAUNCH_UA(false, 3); }
                    else if (extras) { if (ac) GORDER_LAUNCH_UA(true, 2); else GORDER_LAUNCH_UA(false, 2); }
                    else { if (ac) GORDER_LAUNCH_UA(true, 0); else GORDER_LAUNCH_UA(false, 0); }
#undef GORDER_LAUNCH_UA
                }
                {
                    if (staged) {   // second step: slot-major accumulation of the staged samples in LDS
                        const uint32_t planes = h->tables.leaflets.method != GORDER_LEAFLETS_NONE ? 2u : 1u;
                        const uint32_t ntm = h->map_nx * h->map_ny, n_words = planes * ntm;
                        // enough blocks for ~2 per CU; a block flushes <= n_words atomics, so keep its chunk long
                        // bonds: a lane per molecule of the slot (256 threads do for the slots of one molecule type of up to
                        // 256 molecules), so more, shorter chunks; united atoms: the whole block strides over long runs
                        const uint32_t mthreads = 1024u;
                        uint32_t mchunks = std::max(1u, (pass == 0 ? 512u : 512u) / std::max(1u, p.n_acc));
                        if (const char *ev = getenv("GORDER_HIP_MAP_CHUNKS")) mchunks = (uint32_t)std::max(1, atoi(ev));
                        mchunks = std::min(mchunks, std::max(1u, nf / 16u));
                        const uint32_t mfpc = ((nf + mchunks - 1) / mchunks + 15u) / 16u * 16u;   // whole frame blocks
                        mchunks = (nf + mfpc - 1) / mfpc;
                        name("k_map_accumulate");
                        hipLaunchKernelGGL(k_map_accumulate, dim3(p.n_acc * mchunks), dim3(mthreads),
                                           n_words * sizeof(unsigned long long), h->stream, h->d_map_rec,
                                           pass == 0 ? h->d_runs : h->d_ua_runs, pass == 0 ? h->d_run_begin : h->d_ua_run_begin,
                                           p.n_acc, nf, e.rec_stride, mfpc, pass == 0 ? 1u : 3u, n_words, ntm, h->d_map_packed,
                                           p.n_acc);
                    }
                }
                HIP_TRY(h, hipGetLastError());
            }
        }
    }
    if (!p.direct.empty()) {
        const uint32_t n_items = (uint32_t)p.direct.size();
        const uint32_t bpc = (n_items + kBlock - 1) / kBlock;
        const uint32_t target = h->wg_target ? h->wg_target : 256u * 8u;
        uint32_t n_chunks = std::max(1u, (target + bpc - 1) / bpc);
        n_chunks = std::min(n_chunks, a.n_frames);
        const uint32_t fpc = (a.n_frames + n_chunks - 1) / n_chunks;
        n_chunks = (a.n_frames + fpc - 1) / fpc;
        FrameArgs b = a;
        b.frames_per_chunk = fpc;
        name("k_bonds_direct");
        if (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS)
            hipLaunchKernelGGL(k_bonds_direct<true>, dim3(bpc * n_chunks), dim3(kBlock), 0, h->stream, b, h->d_direct,
                               n_items, bpc);
        else
            hipLaunchKernelGGL(k_bonds_direct<false>, dim3(bpc * n_chunks), dim3(kBlock), 0, h->stream, b, h->d_direct,
                               n_items, bpc);
        HIP_TRY(h, hipGetLastError());
    }
#undef name
    return GORDER_OK;       // (the frames are counted by k_batch_end, which also closes the batch's timing chain)
}

}  // namespace

// ---- RCCL: SystemTopology::reduce across the GPUs of a node (topology/mod.rs:256-272) -----------------------------
// The library does not link RCCL: the few entry points are bound at the first call (dlopen by SONAME, so a process
// that already holds RCCL — e.g. through PyTorch — shares that copy; otherwise /opt/rocm's).
namespace {
struct UniqueId { char internal[128]; };     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct RcclApi {
    void *lib = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*comm_init_rank)(void **, int, UniqueId /* ncclUniqueId, by value */, int) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    const char *(*get_error_string)(int) = nullptr;
};
constexpr int kNcclInt64 = 4, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h
RcclApi *rccl_api(std::string *why) {
    static RcclApi api;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [&]() {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.lib) {
            err = std::string("cannot load RCCL: ") + dlerror();
        } else {
            api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.lib, "ncclGetUniqueId");
            api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.lib, "ncclCommInitRank");
            api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.lib, "ncclCommDestroy");
            api.all_reduce = (decltype(api.all_reduce))dlsym(api.lib, "ncclAllReduce");
            api.group_start = (decltype(api.group_start))dlsym(api.lib, "ncclGroupStart");
            api.group_end = (decltype(api.group_end))dlsym(api.lib, "ncclGroupEnd");
            api.get_error_string = (decltype(api.get_error_string))dlsym(api.lib, "ncclGetErrorString");
            if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce || !api.group_start ||
                !api.group_end) {
                err = "RCCL library lacks an expected entry point";
                api.lib = nullptr;
            }
        }
    });
    if (!api.lib) { if (why) *why = err; return nullptr; }
    return &api;
}
std::string rccl_error(const RcclApi *api, const char *what, int rc) {
    return std::string(what) + ": " + (api->get_error_string ? api->get_error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}
}  // namespace


extern "C" {

const char *gorder_hip_strerror(int status) {
    switch (status) {
        case GORDER_OK: return "ok";
        case GORDER_ERR_UNDEFINED_BOX: return "system has undefined simulation box";
        case GORDER_ERR_NOT_ORTHOGONAL_BOX: return "the simulation box is not orthogonal";
        case GORDER_ERR_ZERO_BOX: return "all dimensions of the simulation box are zero";
        case GORDER_ERR_UNDEFINED_POSITION: return "atom has an undefined position";
        case GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER: return "could not calculate global membrane center";
        case GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER: return "could not calculate local membrane center";
        case GORDER_ERR_DYNAMIC_NORMAL: return "not enough points for dynamic local membrane normal calculation (need 3)";
        case GORDER_ERR_INVALID_ARGUMENT: return "invalid argument";
        case GORDER_ERR_DEVICE: return "HIP runtime error";
        case GORDER_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
        case GORDER_ERR_BOX_RANGE: return "box edge <= 0 or coordinate too far outside the box";
        case GORDER_ERR_LEAFLETS_NOT_PRIMED: return "leaflet assignment missing for the first frame";
        case GORDER_ERR_OVERFLOW: return "order accumulator overflowed";
        case GORDER_ERR_TRAJECTORY_FORMAT: return "corrupt or truncated trajectory frame";
        default: return "unknown status";
    }
}

namespace {
__global__ void k_selftest_arithmetic(uint64_t n, uint64_t seed, unsigned long long *mismatches) {
    uint64_t bad_div = 0, bad_sqrt = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;          // splitmix64
        auto next = [&]() {
            x += 0x9E3779B97F4A7C15ull;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        auto make = [&](int e_lo, int e_hi, bool sign) {                // random mantissa, exponent uniform in [e_lo, e_hi]
            const uint64_t r = next();
            const uint32_t e = (uint32_t)(127 + e_lo + (int)((r >> 32) % (uint64_t)(e_hi - e_lo + 1)));
            uint32_t bits = (e << 23) | (uint32_t)(r & 0x7fffffu);
            if (sign && (r >> 63)) bits |= 0x80000000u;
            return __uint_as_float(bits);
        };
        float d = make(-40, 39, false);
        if (i % 1024u == 0) d = 0x1p+40f;
        if (i % 1024u == 1) d = 0x1p-40f;
        float num = make(-100, 60, true);
        if (i % 64u == 0) num = 0.0f;
        if (i % 64u == 1) num = d;                                        // quotient exactly 1
        if (i % 64u == 2) num = d * 0.5f;                                 // exact ties of the rounding to a tile index
        const float q_core = gm_div_core(num, d), q_ieee = num / d;
        if (__float_as_uint(q_core) != __float_as_uint(q_ieee)) bad_div++;
        const float s_core = gm_sqrt_core(d), s_ieee = __builtin_sqrtf(d);
        if (__float_as_uint(s_core) != __float_as_uint(s_ieee)) bad_sqrt++;
        // acos with the cores inside against acos with the IEEE operations: arguments over the whole of [-1, 1]
        // (every exponent down to 2^-60, both ends, the ties of the range split)
        float xa = make(-60, -1, true);
        if (i % 16u == 0) xa = __builtin_copysignf(1.0f - __uint_as_float(0x33800000u) * (float)(next() & 0xffffu), xa);   // 1 - k 2^-24
        if (i % 4096u == 1) xa = 1.0f;
        if (i % 4096u == 2) xa = -1.0f;
        if (i % 4096u == 3) xa = 0.5f;
        if (i % 4096u == 4) xa = 0.0f;
        if (__float_as_uint(gm_acosf_t<true>(xa)) != __float_as_uint(gm_acosf_t<false>(xa))) bad_sqrt++;
    }
    if (bad_div) atomicAdd(&mismatches[0], (unsigned long long)bad_div);
    if (bad_sqrt) atomicAdd(&mismatches[1], (unsigned long long)bad_sqrt);
}
}  // namespace

int gorder_hip_selftest_arithmetic(int device, uint64_t n, uint64_t seed, uint64_t mismatches[2]) {
    if (!mismatches) return GORDER_ERR_INVALID_ARGUMENT;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GORDER_ERR_NO_DEVICE;
    if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess) return GORDER_ERR_INVALID_ARGUMENT;
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 2 * sizeof(unsigned long long)) != hipSuccess) return GORDER_ERR_DEVICE;
    int st = GORDER_OK;
    if (hipMemset(d, 0, 2 * sizeof(unsigned long long)) != hipSuccess) st = GORDER_ERR_DEVICE;
    if (st == GORDER_OK) {
        hipLaunchKernelGGL(k_selftest_arithmetic, dim3(4096), dim3(256), 0, 0, n, seed, d);
        unsigned long long out[2] = {0, 0};
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess)
            st = GORDER_ERR_DEVICE;
        mismatches[0] = out[0];
        mismatches[1] = out[1];
    }
    (void)hipFree(d);
    return st;
}

namespace {
__global__ void k_selftest_trig(int fn, uint32_t first_bits, uint32_t stride, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(first_bits + i * stride);
    float s, c;
    gm_sincosf_0pi(x, s, c);
    out[i] = fn == 0 ? gm_acosf_t<false>(x) : fn == 1 ? gm_acosf_t<true>(x) : fn == 2 ? c : s;
}
}  // namespace

int gorder_hip_selftest_trig(int device, int fn, uint32_t first_bits, uint32_t stride, uint32_t n, float *out) {
    if (!out || fn < 0 || fn > 3) return GORDER_ERR_INVALID_ARGUMENT;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GORDER_ERR_NO_DEVICE;
    if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess) return GORDER_ERR_INVALID_ARGUMENT;
    if (n == 0) return GORDER_OK;
    float *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)n * sizeof(float)) != hipSuccess) return GORDER_ERR_DEVICE;
    hipLaunchKernelGGL(k_selftest_trig, dim3((n + 255u) / 256u), dim3(256), 0, 0, fn, first_bits, stride, n, d);
    int st = GORDER_OK;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        st = GORDER_ERR_DEVICE;
    (void)hipFree(d);
    return st;
}

int gorder_hip_plan_tables(const gorder_tables_t *tables, gorder_hip_plan_t *out, int *selfcheck) {
    if (!tables || !out) return GORDER_ERR_INVALID_ARGUMENT;
    Plan p;
    const int st = gorder::build_plan(*tables, env_flag("GORDER_HIP_FORCE_DIRECT"), p);
    if (st != GORDER_OK) return st;
    out->n_tiles = (uint32_t)p.tiles.size();
    out->block_threads = kBlock;
    out->max_window_atoms = p.max_window;
    out->n_direct_items = (uint32_t)p.direct.size();
    out->frames_per_stage = kFramesPerStage;
    const uint32_t lw = ((3u * p.max_window + 3u + 3u) / 4u) * 4u;
    size_t lds = (size_t)kFramesPerStage * lw * sizeof(float);
    if (lds < (size_t)kBlock * 24) lds = (size_t)kBlock * 24;
    out->lds_bytes = (uint32_t)lds;
    out->map_staged = 0;         // decided at gorder_hip_create (device LDS size, GORDER_HIP_MAP_DIRECT)
    out->map_lds_bytes = 0;
    out->leaflets_one_read = p.spec_ok ? 1u : 0u;
    if (selfcheck) *selfcheck = gorder::selfcheck_plan(*tables, p);
    return GORDER_OK;
}

int gorder_hip_create(const gorder_tables_t *t, gorder_hip_handle **out) {
    if (!t || !out) return GORDER_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return GORDER_ERR_NO_DEVICE;
    if (t->device < 0 || t->device >= n_dev) return GORDER_ERR_INVALID_ARGUMENT;

    gorder_hip_handle *h = new (std::nothrow) gorder_hip_handle();
    if (!h) return GORDER_ERR_DEVICE;
    *out = h;   // returned even on failure so that the caller can read the message; destroy() is safe
    h->tables = *t;
    h->tables.molecule_types = nullptr;
    h->tables.leaflets.membrane = nullptr;
    h->device = t->device;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((t->flags & GORDER_FLAG_UA_FAST_NORMALISE) && (t->flags & GORDER_FLAG_TRIG_ACOS_COS))
        return fail(h, GORDER_ERR_INVALID_ARGUMENT, "GORDER_FLAG_UA_FAST_NORMALISE (tolerance-bounded) and GORDER_FLAG_TRIG_ACOS_COS (literal) exclude each other");
    // united atoms: a wave = 4 slots x 16 molecules (plan.h) — except for per-frame rows and nothing else, where a wave of
    // ONE slot sends a fourth of the atomics (0.433 against 0.449 ms per 3 000 frames of the 256-lipid membrane)
    const bool ua_rows_only = t->timewise && !t->ordermap.enabled && t->geometry.kind == GORDER_GEOM_NONE && !t->dynamic_normal.enabled;
    int st = gorder::build_plan(*t, env_flag("GORDER_HIP_FORCE_DIRECT"), h->plan, !env_flag("GORDER_HIP_UA_SLOT_WAVES") && !ua_rows_only);
    if (st != GORDER_OK) return fail(h, st, "invalid bond tables");
    const Plan &p = h->plan;
    for (uint32_t m = 0; m < t->n_molecule_types; m++)
        h->map_max_mol = std::max(h->map_max_mol, t->molecule_types[m].n_molecules);
    HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    if ((st = upload(h, &h->d_tiles, p.tiles)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_items, p.items)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_tile_slots, p.tile_slots)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_direct, p.direct)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_tiles, p.ua_tiles)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_items, p.ua_items)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_tile_slots, p.ua_tile_slots)) != GORDER_OK) return st;
    {
        ExtraArgs &e = h->extra;
        // construction angles of the united-atom hydrogens (uaorder.rs:34-41); sin/cos with the host
        // libm, as the reference's nalgebra Rotation3::from_axis_angle does
        e.sin_tet = sinf(1.910633f); e.cos_tet = cosf(1.910633f);
        e.sin_ch3 = sinf(2.0943952f); e.cos_ch3 = cosf(2.0943952f);
        e.sin_half = sinf(0.9553165f); e.cos_half = cosf(0.9553165f);
        const gorder_ordermap_t &om = t->ordermap;
        if (om.enabled) {
            if (!(om.bin[0] > 0.0f) || !(om.bin[1] > 0.0f) || om.plane > 2)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "ordermap bin/plane");
            // groan_rs GridMap::new: n = round(span / bin) + 1 tiles per axis
            const float fx = roundf((om.span_x[1] - om.span_x[0]) / om.bin[0]);
            const float fy = roundf((om.span_y[1] - om.span_y[0]) / om.bin[1]);
            if (!(fx >= 0.0f) || !(fy >= 0.0f) || fx > 65535.0f || fy > 65535.0f)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "ordermap span");
            h->map_nx = (uint32_t)fx + 1u;
            h->map_ny = (uint32_t)fy + 1u;
            const size_t nmap = 3 * (size_t)p.n_acc * h->map_nx * h->map_ny;
            HIP_TRY(h, hipMalloc((void **)&h->d_map_sums, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMalloc((void **)&h->d_map_cnts, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_sums, 0, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_cnts, 0, nmap * sizeof(unsigned long long)));
            const size_t npk = (t->leaflets.method != GORDER_LEAFLETS_NONE ? 2 : 1) * (nmap / 3);
            HIP_TRY(h, hipMalloc((void **)&h->d_map_packed, npk * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_packed, 0, npk * sizeof(unsigned long long)));
            // united atoms: stage + accumulate in LDS when one slot's packed map (x2 with leaflets) fits
            const size_t lds_bytes = npk / p.n_acc * sizeof(unsigned long long);
            h->map_lds_bytes = lds_bytes;
            if (lds_bytes <= 150u * 1024u && !env_flag("GORDER_HIP_MAP_DIRECT")) {
                if ((st = upload(h, &h->d_ua_runs, p.ua_runs)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_ua_run_begin, p.ua_run_begin)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_runs, p.runs)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_run_begin, p.run_begin)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_items_by_slot, p.items_by_slot)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_ua_item_run, p.ua_item_run)) != GORDER_OK) return st;
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_map_accumulate),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                h->map_staged = true;
            }
            e.maps = 1; e.plane = om.plane; e.x0 = om.span_x[0]; e.y0 = om.span_y[0];
            e.binx = om.bin[0]; e.biny = om.bin[1]; e.nx = h->map_nx; e.ny = h->map_ny;
            e.inv_binx = 1.0f / om.bin[0]; e.inv_biny = 1.0f / om.bin[1];
            e.bin_core = (om.bin[0] >= 0x1p-40f && om.bin[0] <= 0x1p+40f && om.bin[1] >= 0x1p-40f && om.bin[1] <= 0x1p+40f) ? 1 : 0;
            e.map_packed = h->d_map_packed;
        }
        e.tw = t->timewise ? 1 : 0;
        if (t->timewise && !p.tiles.empty()) {      // k_bonds_tiled_tw: the tile items in slot order and their runs
            if (!h->d_items_by_slot && (st = upload(h, &h->d_items_by_slot, p.items_by_slot)) != GORDER_OK) return st;
            if ((st = upload(h, &h->d_item_run, p.item_run)) != GORDER_OK) return st;
        }
        const gorder_geometry_t &ge = t->geometry;
        if (ge.kind != GORDER_GEOM_NONE) {
            if (ge.kind > GORDER_GEOM_SPHERE || ge.reference > GORDER_GEOMREF_GROUP || ge.orientation > 2)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry kind/reference/orientation");
            if (ge.reference == GORDER_GEOMREF_BOX_CENTER && !t->handle_pbc)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "box-centre reference needs handle_pbc (pbc.rs:243-245)");
            if (ge.reference == GORDER_GEOMREF_POINT && t->handle_pbc &&
                !(ge.structure_box[0] > 0.0f && ge.structure_box[1] > 0.0f && ge.structure_box[2] > 0.0f))
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry.structure_box");
            if (ge.reference == GORDER_GEOMREF_GROUP) {
                if (!ge.group || ge.n_group == 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry.group");
                std::vector<uint32_t> grp(ge.group, ge.group + ge.n_group);
                for (uint32_t a : grp)
                    if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry group index out of range");
                if ((st = upload(h, &h->d_geom_group, grp)) != GORDER_OK) return st;
            }
            e.geom_kind = (int)ge.kind; e.geom_invert = ge.invert ? 1 : 0; e.geom_orient = (int)ge.orientation;
            e.geom_thr = local_radius_threshold(ge.radius);
            h->tables.geometry.group = nullptr;
        }
        if ((e.maps || e.tw || e.geom_kind) && !p.direct.empty())
            return fail(h, GORDER_ERR_INVALID_ARGUMENT,
                        "ordermaps / timewise / geometry need every bond to fit an atom window");
    }
    HIP_TRY(h, hipMalloc((void **)&h->d_err, (kErrWords + 2u) * sizeof(uint32_t)));     // (+ the ticket of k_batch_end)
    HIP_TRY(h, hipMemset(h->d_err, 0xff, kErrWords * sizeof(uint32_t)));   // kErrNone
    HIP_TRY(h, hipMemset(h->d_err + kErrWords, 0, 2u * sizeof(uint32_t)));
    h->acc_words = 4 * (size_t)p.n_acc + 1;
    HIP_TRY(h, hipMalloc((void **)&h->d_acc, h->acc_words * sizeof(unsigned long long)));
    HIP_TRY(h, hipMemset(h->d_acc, 0, h->acc_words * sizeof(unsigned long long)));
    if (const char *e = getenv("GORDER_HIP_MAP_FOLD_LIMIT")) {
        const long v = atol(e);
        if (v >= 2 && (unsigned long long)v <= kMapFoldLimit) h->map_fold_limit = (uint64_t)v;
    }
    if (const char *e = getenv("GORDER_HIP_REPLICAS")) {
        const int r = atoi(e);
        if (r >= 1 && r <= 1024) h->n_rep = (uint32_t)r;
    }
    if (p.n_acc) {
        const size_t rep_bytes = (size_t)h->n_rep * 4u * p.n_acc * sizeof(unsigned long long);
        HIP_TRY(h, hipMalloc((void **)&h->d_rep, rep_bytes));
        HIP_TRY(h, hipMemset(h->d_rep, 0, rep_bytes));
    }
    {   // |normal| with the f32 sequence of nalgebra's norm (oracle: norm3)
        const float *n = t->normal;
        h->n2sq = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
        h->n2 = sqrtf(h->n2sq);
        for (int d = 0; d < 3; d++)
            if (n[d] == 1.0f && n[(d + 1) % 3] == 0.0f && n[(d + 2) % 3] == 0.0f) h->axis = d;
        h->extra.axis = h->axis;
    }
    if (const char *e = getenv("GORDER_HIP_KERNEL")) h->use_gather = strcmp(e, "gather") == 0;
    if (const char *e = getenv("GORDER_HIP_WG_TARGET")) {
        const int w = atoi(e);
        if (w > 0) h->wg_target = (uint32_t)w;
    }
    h->lw = ((3u * p.max_window + 3u + 3u) / 4u) * 4u;
    h->lds_bytes = (size_t)h->frames_per_stage * h->lw * sizeof(float);
    if (h->lds_bytes < (size_t)kBlock * 24) h->lds_bytes = (size_t)kBlock * 24;
    {   // how many workgroups of the tiled kernel are co-resident: the frame range of a batch is cut so
        // that the grid is a whole number of such rounds (no half-empty last round)
        int n_cu = 256, per_cu = 6;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
        const bool ac = (t->flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        const hipError_t e = ac ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<4, 5, true, true, false, -1>, kBlock, h->lds_bytes)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<4, 5, false, true, false, 2>, kBlock, h->lds_bytes);
        if (e != hipSuccess || per_cu < 1) per_cu = 4;
        h->wg_capacity = (uint32_t)n_cu * (uint32_t)per_cu;
    }

    const gorder_dynamic_normal_t &dn = t->dynamic_normal;
    if (dn.enabled) {
        if (!(dn.radius > 0.0f)) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need a positive radius");
        if (!dn.cloud || dn.n_cloud == 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need the NormalHeads group");
        if (!p.direct.empty()) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals: a bond spans more than the LDS window");
        std::vector<uint32_t> cloud(dn.cloud, dn.cloud + dn.n_cloud), nheads;
        for (uint32_t a : cloud)
            if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "NormalHeads index out of range");
        for (uint32_t m = 0; m < t->n_molecule_types; m++) {
            const gorder_moltype_t &mt = t->molecule_types[m];
            if (!mt.normal_heads) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need normal_heads[] per molecule type");
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                if (mt.normal_heads[k] >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "normal head index out of range");
                nheads.push_back(mt.normal_heads[k]);
            }
        }
        if ((st = upload(h, &h->d_dyn_cloud, cloud)) != GORDER_OK) return st;
        if ((st = upload(h, &h->d_dyn_heads, nheads)) != GORDER_OK) return st;
        h->host_dyn_heads = nheads;
        const size_t nm = dn.n_cloud, ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
        const size_t sl = h->dyn_slab = local_slab_frames(nm);
        for (const void *fn : {reinterpret_cast<const void *>(k_local_build<2>), reinterpret_cast<const void *>(k_local_build<5>),
                               reinterpret_cast<const void *>(k_local_build<kLocalBuildTrips>)})
            HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLocalBuildLds));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_cell_of, sl * nm * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_rec, sl * nm * 4 * sizeof(float)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_count, sl * (2 * ncell + 1) * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_cov, sl * (size_t)(p.n_mol_total ? p.n_mol_total : 1) * sizeof(DynCov)));
        h->dyn = true;
    }

    const gorder_leaflets_t &lf = t->leaflets;
    if (lf.method != GORDER_LEAFLETS_NONE) {
        if (lf.normal_dim > 2) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "leaflets.normal_dim");
        std::vector<uint32_t> heads, mb(1, 0), ma;
        for (uint32_t m = 0; m < t->n_molecule_types; m++) {
            const gorder_moltype_t &mt = t->molecule_types[m];
            if (lf.method != GORDER_LEAFLETS_MANUAL && !mt.heads)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "leaflets need heads[] per molecule type");
            if (lf.method == GORDER_LEAFLETS_INDIVIDUAL && (!mt.methyls || mt.n_methyls == 0))
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "individual leaflets need methyls[]");
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                const uint32_t hd = mt.heads ? mt.heads[k] : 0;
                if (hd >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "head index out of range");
                heads.push_back(hd);
                if (lf.method == GORDER_LEAFLETS_INDIVIDUAL)
                    for (uint32_t q = 0; q < mt.n_methyls; q++) {
                        const uint32_t a = mt.methyls[(size_t)k * mt.n_methyls + q];
                        if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "methyl index out of range");
                        ma.push_back(a);
                    }
                mb.push_back((uint32_t)ma.size());
            }
        }
        if ((st = upload(h, &h->d_heads, heads)) != GORDER_OK) return st;
        h->host_heads = heads;
        if (lf.method == GORDER_LEAFLETS_LOCAL) {   // error key of a failed local centre: first slot of the molecule's type
            std::vector<uint32_t> ms;
            for (uint32_t m = 0; m < t->n_molecule_types; m++) ms.insert(ms.end(), t->molecule_types[m].n_molecules, p.slot0[m]);
            if ((st = upload(h, &h->d_mol_slot0, ms)) != GORDER_OK) return st;
        }
        if ((st = upload(h, &h->d_methyl_begin, mb)) != GORDER_OK) return st;
        if ((st = upload(h, &h->d_methyl_atoms, ma)) != GORDER_OK) return st;
        if (lf.method == GORDER_LEAFLETS_LOCAL && !(lf.radius > 0.0f))
            return fail(h, GORDER_ERR_INVALID_ARGUMENT, "local leaflets need a positive radius");
        if (lf.method == GORDER_LEAFLETS_GLOBAL || lf.method == GORDER_LEAFLETS_LOCAL) {
            if (!lf.membrane || lf.n_membrane == 0)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "global/local leaflets need the membrane group");
            std::vector<uint32_t> mem(lf.membrane, lf.membrane + lf.n_membrane);
            for (uint32_t a : mem)
                if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "membrane index out of range");
            if ((st = upload(h, &h->d_membrane, mem)) != GORDER_OK) return st;
            h->membrane_is_frame = lf.n_membrane == t->n_atoms && !env_flag("GORDER_HIP_LEAFLETS_GENERIC");
            for (uint32_t i = 0; i < lf.n_membrane && h->membrane_is_frame; i++) h->membrane_is_frame = mem[i] == i;
        }
        // one read for global leaflets + order parameters: the tables of k_bonds_tiled<..., MOM> and k_spec_fixup
        if (lf.method == GORDER_LEAFLETS_GLOBAL && p.spec_ok && !env_flag("GORDER_HIP_NO_SPECULATE")) {
            std::vector<uint2> own(p.tiles.size());
            for (size_t i = 0; i < own.size(); i++) own[i] = make_uint2(p.own[2 * i], p.own[2 * i + 1]);
            if ((st = upload(h, &h->d_own, own)) != GORDER_OK) return st;
            std::vector<std::vector<SpecSample>> per_mol(p.n_mol_total);
            for (const Tile &tile : p.tiles)
                for (uint32_t q = 0; q < tile.n_items; q++) {
                    const Item &it = p.items[tile.item0 + q];
                    per_mol[it.mol].push_back({tile.atom0 + it.li, tile.atom0 + it.lj, p.tile_slots[tile.slot0 + it.lslot]});
                }
            std::vector<uint32_t> mbeg(1, 0);
            std::vector<SpecSample> all;
            for (const auto &v : per_mol) { all.insert(all.end(), v.begin(), v.end()); mbeg.push_back((uint32_t)all.size()); }
            if ((st = upload(h, &h->d_spec_mol_begin, mbeg)) != GORDER_OK) return st;
            if ((st = upload(h, &h->d_spec_samples, all)) != GORDER_OK) return st;
            std::vector<uint2> oh(p.own_heads.size() / 2);
            for (size_t i = 0; i < oh.size(); i++) oh[i] = make_uint2(p.own_heads[2 * i], p.own_heads[2 * i + 1]);
            if ((st = upload(h, &h->d_own_head_begin, p.own_head_begin)) != GORDER_OK) return st;
            if ((st = upload(h, &h->d_own_heads, oh)) != GORDER_OK) return st;
            HIP_TRY(h, hipMalloc((void **)&h->d_spec_counters, 4 * sizeof(uint32_t)));
            HIP_TRY(h, hipMemset(h->d_spec_counters, 0, 4 * sizeof(uint32_t)));
            HIP_TRY(h, hipHostMalloc((void **)&h->h_spec_counters, 2 * gorder_hip_handle::kSpecRing * sizeof(uint32_t)));
            for (hipEvent_t &ev : h->spec_counters_copied) HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            h->spec_enabled = true;
        }
        if (lf.method == GORDER_LEAFLETS_LOCAL) {
            const size_t nm = lf.n_membrane, ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
            const size_t sl = h->local_slab = local_slab_frames(nm);
            for (const void *fn : {reinterpret_cast<const void *>(k_local_build<2>), reinterpret_cast<const void *>(k_local_build<5>),
                                   reinterpret_cast<const void *>(k_local_build<kLocalBuildTrips>)})
                HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLocalBuildLds));
            // rows of cells with a halo (periodic boxes, k_local_flags_rows): the first cells of a row are there twice,
            // records included — room for twice the membrane
            h->local_halo = t->handle_pbc && !env_flag("GORDER_HIP_LOCAL_ATOMS_ONLY");
            h->local_rec_stride = (h->local_halo ? 2u : 1u) * nm;
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_of, sl * nm * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&h->d_ltrig, sl * h->local_rec_stride * sizeof(LocalRec)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_count, sl * (ncell + 1) * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_fill, sl * ncell * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lgrid, sl * sizeof(uint4)));
            if (h->local_halo) {    // (GORDER_HIP_LOCAL_ATOMS_ONLY, an A/B switch: every candidate atom by atom, the general passes)
                HIP_TRY(h, hipMalloc((void **)&h->d_lrowpre, sl * (size_t)kLocalMaxCells1D * (kLocalMaxCells1D + 1u) * sizeof(LocalRowPre)));
                HIP_TRY(h, hipMalloc((void **)&h->d_ledge, sl * (size_t)kLocalMaxCells1D * (kLocalMaxCells1D + 1u) * sizeof(LocalEdge)));
                HIP_TRY(h, hipMalloc((void **)&h->d_lfinfo, sl * sizeof(float4)));
                HIP_TRY(h, hipMalloc((void **)&h->d_lneed, (sl + 1) * sizeof(uint32_t)));
                HIP_TRY(h, hipMalloc((void **)&h->d_lsummary, 2 * sizeof(uint32_t)));
                HIP_TRY(h, hipMemset(h->d_lsummary, 0, 2 * sizeof(uint32_t)));
                HIP_TRY(h, hipHostMalloc((void **)&h->h_lsummary, 2 * sizeof(uint32_t)));
                h->h_lsummary[0] = h->h_lsummary[1] = 0u;
                HIP_TRY(h, hipEventCreateWithFlags(&h->lsummary_written, hipEventDisableTiming));
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_local_decide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)kDecideLds));
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_local_sums), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)kSumsLds));
                HIP_TRY(h, hipMalloc((void **)&h->d_ltodo, (1 + sl * (size_t)(p.n_mol_total ? p.n_mol_total : 1)) * sizeof(uint2)));
            }
        }
        HIP_TRY(h, hipMalloc((void **)&h->d_adist, sizeof(float) * (p.n_mol_total ? p.n_mol_total : 1)));
    }
    // the memsets above ran on the null stream, which the handle's non-blocking stream does not wait for
    HIP_TRY(h, hipDeviceSynchronize());
    return GORDER_OK;
}

void gorder_hip_destroy(gorder_hip_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->traj_cache_free) h->traj_cache_free(h);
    for (hipEvent_t ev : h->timing_ev)
        if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(h->d_mol_slot0);
    (void)hipFree(h->d_tiles); (void)hipFree(h->d_items); (void)hipFree(h->d_tile_slots);
    (void)hipFree(h->d_direct); (void)hipFree(h->d_err); (void)hipFree(h->d_xtc_cp);
    (void)hipFree(h->d_ua_tiles); (void)hipFree(h->d_ua_items); (void)hipFree(h->d_ua_tile_slots);
    (void)hipFree(h->d_map_sums); (void)hipFree(h->d_map_cnts); (void)hipFree(h->d_map_packed); (void)hipFree(h->d_tw_sums); (void)hipFree(h->d_tw_cnts);
    (void)hipFree(h->d_geom_group); (void)hipFree(h->d_shapes); (void)hipFree(h->d_inv_box);
    (void)hipFree(h->d_map_rec); (void)hipFree(h->d_ua_runs); (void)hipFree(h->d_ua_run_begin);
    (void)hipFree(h->d_runs); (void)hipFree(h->d_run_begin); (void)hipFree(h->d_items_by_slot);
    (void)hipFree(h->d_ua_item_run); (void)hipFree(h->d_item_run); (void)hipFree(h->d_lgrid); (void)hipFree(h->d_lrowpre); (void)hipFree(h->d_ledge);
    (void)hipFree(h->d_own); (void)hipFree(h->d_mom); (void)hipFree(h->d_head_z); (void)hipFree(h->d_own_head_begin); (void)hipFree(h->d_own_heads); (void)hipFree(h->d_spec_center); (void)hipFree(h->d_spec_ok);
    (void)hipFree(h->d_spec_counters); (void)hipFree(h->d_spec_mol_begin); (void)hipFree(h->d_spec_samples);
    if (h->h_spec_counters) (void)hipHostFree(h->h_spec_counters);
    for (hipEvent_t ev : h->spec_counters_copied) if (ev) (void)hipEventDestroy(ev); (void)hipFree(h->d_lfinfo); (void)hipFree(h->d_ltodo); (void)hipFree(h->d_lneed);
    (void)hipFree(h->d_lsummary); if (h->h_lsummary) (void)hipHostFree(h->h_lsummary); if (h->lsummary_written) (void)hipEventDestroy(h->lsummary_written);
    (void)hipFree(h->d_dyn_cloud); (void)hipFree(h->d_dyn_heads); (void)hipFree(h->d_dyn_cell_of); (void)hipFree(h->d_dyn_count);
    (void)hipFree(h->d_dyn_rec); (void)hipFree(h->d_dyn_normals); (void)hipFree(h->d_dyn_cov);
    if (!h->acc_external) (void)hipFree(h->d_acc);
    (void)hipFree(h->d_rep);
    (void)hipFree(h->d_heads); (void)hipFree(h->d_membrane); (void)hipFree(h->d_methyl_begin);
    (void)hipFree(h->d_methyl_atoms); (void)hipFree(h->d_aflags); (void)hipFree(h->d_adist);
    (void)hipFree(h->d_arow); (void)hipFree(h->d_aframes);
    (void)hipFree(h->d_lcell_of); (void)hipFree(h->d_lcell_count); (void)hipFree(h->d_lcell_fill);
    (void)hipFree(h->d_ltrig);
    for (int k = 0; k < 2; k++) {
        (void)hipFree(h->d_stage_xyz[k]); (void)hipFree(h->d_stage_box[k]);
        if (h->stage_copied[k]) (void)hipEventDestroy(h->stage_copied[k]);
        if (h->stage_computed[k]) (void)hipEventDestroy(h->stage_computed[k]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

uint32_t gorder_hip_n_accumulators(const gorder_hip_handle *h) { return h ? h->plan.n_acc : 0; }
uint32_t gorder_hip_ordermap_dims(const gorder_hip_handle *h, uint32_t *nx, uint32_t *ny) {
    if (nx) *nx = h ? h->map_nx : 0;
    if (ny) *ny = h ? h->map_ny : 0;
    return h ? h->map_nx * h->map_ny : 0;
}

int gorder_hip_set_stream(gorder_hip_handle *h, void *hip_stream) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return GORDER_OK;
}

static void spec_poll(gorder_hip_handle *h, bool wait);
int gorder_hip_speculation_stats(gorder_hip_handle *h, uint64_t out[4]) {
    if (!h || !out) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    spec_poll(h, true);
    out[0] = h->spec_batches; out[1] = h->spec_fixed; out[2] = h->spec_exact_frames; out[3] = h->spec_enabled ? 1 : 0;
    return GORDER_OK;
}

int gorder_hip_local_decide_stats(gorder_hip_handle *h, uint64_t out[4]) {
    if (!h || !out) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    out[0] = h->decide_submits; out[1] = h->decide_paused_submits;
    out[2] = h->h_lsummary ? h->h_lsummary[0] : 0; out[3] = h->h_lsummary ? h->h_lsummary[1] : 0;
    return GORDER_OK;
}

int gorder_hip_plan(const gorder_hip_handle *h, gorder_hip_plan_t *plan) {
    if (!h || !plan) return GORDER_ERR_INVALID_ARGUMENT;
    plan->n_tiles = (uint32_t)h->plan.tiles.size();
    plan->block_threads = kBlock;
    plan->max_window_atoms = h->plan.max_window;
    plan->n_direct_items = (uint32_t)h->plan.direct.size();
    plan->frames_per_stage = (uint32_t)h->frames_per_stage;
    plan->lds_bytes = (uint32_t)h->lds_bytes;
    plan->map_staged = h->map_staged ? 1u : 0u;
    plan->map_lds_bytes = (uint32_t)h->map_lds_bytes;
    plan->leaflets_one_read = h->plan.spec_ok ? 1u : 0u;
    return GORDER_OK;
}

// What the speculative batches that have finished cost (never waits unless `wait`): their counters are added to the
// handle's statistics, and a batch that left more than 1/8 of its frames to the exact kernel, or mispredicted more than
// 1/16 of its (frame, molecule) pairs, ends the speculation for this handle — a membrane across the periodic boundary,
// lipids that keep changing sides: the two-kernel path is the cheaper one for such a trajectory.
static void spec_poll(gorder_hip_handle *h, bool wait) {
    for (uint32_t i = 0; i < gorder_hip_handle::kSpecRing; i++) {
        if (!h->spec_ring_frames[i]) continue;
        if (wait) (void)hipEventSynchronize(h->spec_counters_copied[i]);
        else if (hipEventQuery(h->spec_counters_copied[i]) != hipSuccess) continue;
        const uint32_t moved = h->h_spec_counters[2 * i], exact = h->h_spec_counters[2 * i + 1];
        h->spec_fixed += moved;
        h->spec_exact_frames += exact;
        if ((uint64_t)exact * 8u > h->spec_ring_frames[i] ||
            (uint64_t)moved * 16u > h->spec_ring_frames[i] * h->plan.n_mol_total)
            h->spec_enabled = false;
        h->spec_ring_frames[i] = 0;
    }
}

// ---- leaflet assignment rows for a batch (host part of leaflets.rs:435-441, 1437-1472) --------
static int run_leaflets(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                        const std::vector<uint32_t> &aframes, uint32_t row0, const uint8_t *skip = nullptr) {
    if (aframes.empty()) return GORDER_OK;
    int st;
    const size_t cap_before = h->aframes_cap;
    if ((st = ensure(h, &h->d_aframes, &h->aframes_cap, aframes.size())) != GORDER_OK) return st;
    if (h->aframes_cap != cap_before) h->up_aframes_at = nullptr;      // a new allocation holds nothing yet
    if (h->up_aframes_at != h->d_aframes || h->up_aframes != aframes) {
        HIP_TRY(h, hipMemcpyAsync(h->d_aframes, aframes.data(), aframes.size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, h->stream));
        h->up_aframes = aframes;
        h->up_aframes_at = h->d_aframes;
    }
    const gorder_leaflets_t &lf = h->tables.leaflets;
    LeafletArgs la{};
    la.xyz = d_xyz; la.box9 = d_box; la.n_atoms = h->plan.n_atoms;
    la.aframes = h->d_aframes; la.row0 = row0; la.aflags = h->d_aflags; la.adist = h->d_adist;
    la.n_mol_total = h->plan.n_mol_total; la.heads = h->d_heads;
    la.membrane = h->d_membrane; la.n_membrane = lf.n_membrane;
    la.methyl_begin = h->d_methyl_begin; la.methyl_atoms = h->d_methyl_atoms;
    la.dim = lf.normal_dim; la.flip = lf.flip ? 1 : 0; la.pbc = h->tables.handle_pbc ? 1 : 0;
    la.err = h->d_err;
    la.skip = skip;
    la.n_assign = (uint32_t)aframes.size();
    if (lf.method == GORDER_LEAFLETS_GLOBAL) {
        // (behind a speculative batch nearly every frame is skipped: a few hundred workgroups take the frames in turn)
        const dim3 g(skip ? std::min<uint32_t>((uint32_t)aframes.size(), 512u) : (uint32_t)aframes.size()), b(1024);
        TIMING_MARK(h, h->membrane_is_frame ? "k_leaflets_global_contig" : "k_leaflets_global");
        if (h->membrane_is_frame) {
            if (skip) hipLaunchKernelGGL(k_leaflets_global_contig<true>, g, dim3(256), 0, h->stream, la);
            else hipLaunchKernelGGL(k_leaflets_global_contig<false>, g, dim3(256), 0, h->stream, la);
        } else {
            if (skip) hipLaunchKernelGGL(k_leaflets_global<true>, g, b, 0, h->stream, la);
            else hipLaunchKernelGGL(k_leaflets_global<false>, g, b, 0, h->stream, la);
        }
    } else if (lf.method == GORDER_LEAFLETS_INDIVIDUAL) {
        // gridDim.y <= 65535: launch in slabs
        TIMING_MARK(h, "k_leaflets_individual");
        size_t done = 0;
        while (done < aframes.size()) {
            const uint32_t ny = (uint32_t)std::min<size_t>(aframes.size() - done, 65535);
            LeafletArgs lb = la;
            lb.aframes = h->d_aframes + done;
            lb.row0 = row0 + (uint32_t)done;
            // adist is only written by the last slab's last row
            if (done + ny < aframes.size()) lb.adist = nullptr;
            hipLaunchKernelGGL(k_leaflets_individual, dim3((la.n_mol_total + 255) / 256, ny), dim3(256), 0,
                               h->stream, lb);
            done += ny;
        }
    }
    else if (lf.method == GORDER_LEAFLETS_LOCAL) {
        const size_t ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
        LocalArgs lo{};
        lo.xyz = d_xyz; lo.box9 = d_box; lo.n_atoms = h->plan.n_atoms;
        lo.aflags = h->d_aflags; lo.adist = h->d_adist; lo.n_mol_total = h->plan.n_mol_total;
        lo.heads = h->d_heads; lo.mol_slot0 = h->d_mol_slot0; lo.n_membrane = lf.n_membrane;
        lo.membrane = h->membrane_is_frame ? nullptr : h->d_membrane;      // (the whole frame in order: no index list)
        lo.dim = lf.normal_dim; lo.flip = lf.flip ? 1 : 0; lo.pbc = h->tables.handle_pbc ? 1 : 0;
        lo.radius = lf.radius;
        lo.radius_thr = local_radius_threshold(lf.radius);
        lo.cell_of = h->d_lcell_of; lo.trig = h->d_ltrig; lo.cell_count = h->d_lcell_count;
        lo.cell_fill = h->d_lcell_fill; lo.err = h->d_err;
        lo.grid = h->d_lgrid;
        lo.halo = h->local_halo ? 1 : 0;
        lo.prune = env_flag("GORDER_HIP_LOCAL_NO_PRUNE") ? 0 : (env_flag("GORDER_HIP_LOCAL_DECIDE_NOTHING") ? 2 : 1);
        lo.rec_stride = (uint32_t)h->local_rec_stride;
        lo.rowpre = h->d_lrowpre;
        lo.edge = h->d_ledge;
        lo.finfo = h->d_lfinfo;
        lo.todo = h->d_ltodo;
        // the bound first, a lane per head (k_local_decide); the rows kernel then only for the frames with a head left open
        bool decide = lo.halo && lo.prune && lo.n_mol_total && !env_flag("GORDER_HIP_LOCAL_NO_DECIDE");
        if (decide) {
            if (h->lsummary_pending) {
                if (hipEventQuery(h->lsummary_written) == hipSuccess) {
                    h->lsummary_pending = false;
                    if (2u * h->h_lsummary[0] > h->h_lsummary[1]) h->decide_pause = gorder_hip_handle::kDecidePause;
                } else {
                    (void)hipGetLastError();        // (hipErrorNotReady is an answer, not a failure of this submit)
                }
            }
            if (h->decide_pause) { h->decide_pause--; h->decide_paused_submits++; decide = false; }
            else h->decide_submits++;
        }
        lo.need = decide ? h->d_lneed : nullptr;
        // ... and the table that kernel reads made from per-cell sums first (k_local_sums): the cell list only for the frames left open
        lo.sums = decide && lf.n_membrane <= kLocalBuildMax && !env_flag("GORDER_HIP_LOCAL_THREE_KERNELS") &&
                  !env_flag("GORDER_HIP_LOCAL_NO_SUMS") ? 1 : 0;
        const bool report = decide && !h->lsummary_pending;      // (one report in flight)
        lo.summary = report ? h->d_lsummary : nullptr;
        for (size_t done = 0; done < aframes.size(); done += h->local_slab) {
            const uint32_t ns = (uint32_t)std::min<size_t>(aframes.size() - done, h->local_slab);
            lo.aframes = h->d_aframes + done;
            lo.n_slab = ns;
            lo.row0 = row0 + (uint32_t)done;
            lo.write_dist_frame = (done + ns == aframes.size()) ? (int)ns - 1 : -1;
            lo.summary_host = report && done + ns == aframes.size() ? h->h_lsummary : nullptr;
            if (lf.n_membrane > kLocalBuildMax || env_flag("GORDER_HIP_LOCAL_THREE_KERNELS"))
                HIP_TRY(h, hipMemsetAsync(h->d_lcell_fill, 0, ns * ncell * sizeof(uint32_t), h->stream));
            if (lo.sums) {
                HIP_TRY(h, hipMemsetAsync(h->d_lneed, 0, (ns + 1) * sizeof(uint32_t), h->stream));
                TIMING_MARK(h, "k_local_sums");
                hipLaunchKernelGGL(k_local_sums, dim3(ns), dim3(1024), kSumsLds, h->stream, lo);
                TIMING_MARK(h, "k_local_decide");
                hipLaunchKernelGGL(k_local_decide, dim3(ns), dim3(1024), kDecideLds, h->stream, lo);
            }
            const int cst = launch_cell_list(h, lo, ns, lf.n_membrane, h->d_lcell_count, ns * (ncell + 1) * sizeof(uint32_t));
            if (cst != GORDER_OK) return cst;
            if (lo.halo) {
                TIMING_MARK(h, "k_local_rowprefix");
                hipLaunchKernelGGL(k_local_rowprefix, dim3(kLocalMaxCells1D / 4u, ns), dim3(256), 0, h->stream, lo);
                lo.rows_groups = (lo.n_mol_total + 15u) / 16u;
                if (lo.need && !lo.sums) {
                    TIMING_MARK(h, "k_local_decide");
                    hipLaunchKernelGGL(k_local_decide, dim3(ns), dim3(1024), kDecideLds, h->stream, lo);
                }
                TIMING_MARK(h, "k_local_flags_rows");
                if (lo.need) hipLaunchKernelGGL(k_local_flags_rows_open, dim3(lo.rows_groups * 8u * kRowsSlots), dim3(256), 0, h->stream, lo);
                else hipLaunchKernelGGL(k_local_flags_rows, dim3(lo.rows_groups * ((ns + 7u) / 8u * 8u)), dim3(256), 0, h->stream, lo);
                // the heads the rows left over (normally none: the grid finds an empty list and leaves)
                TIMING_MARK(h, "k_local_flags_todo");
                hipLaunchKernelGGL(k_local_flags_todo, dim3(512), dim3(256), 0, h->stream, lo);
            } else {
                TIMING_MARK(h, "k_local_flags");
                hipLaunchKernelGGL(k_local_flags, dim3((lo.n_mol_total + 3) / 4, ns), dim3(256), 0, h->stream, lo);
            }
        }
        if (report && !aframes.empty()) {
            HIP_TRY(h, hipEventRecord(h->lsummary_written, h->stream));
            h->lsummary_pending = true;
        }
    }
    HIP_TRY(h, hipGetLastError());
    return GORDER_OK;
}

int gorder_hip_submit_device(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                             const uint64_t *frame_index, uint32_t n_frames) {
    if (!h || !d_xyz || !frame_index) return GORDER_ERR_INVALID_ARGUMENT;
    const bool pbc = h->tables.handle_pbc != 0;
    if (pbc && !d_box) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "box required when handle_pbc = 1");
    if (((uintptr_t)d_xyz & 15u) != 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "d_xyz must be 16-byte aligned");
    if (n_frames == 0) return GORDER_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const Plan &p = h->plan;
    const gorder_leaflets_t &lf = h->tables.leaflets;
    int st;
    // OrderValue is a checked i64 (order.rs:44-60 panics on overflow).  |tick| <= 1e6 and a slot receives at most one
    // sample per molecule of its type and frame, so no sum can overflow while frames * molecules stays below 2^63 / 1e6:
    // refuse the batch that would cross that bound instead of wrapping silently.
    if ((h->n_frames + n_frames) > (uint64_t)(9223372036854775807ll / 1000000ll) / h->map_max_mol)
        return fail(h, GORDER_ERR_OVERFLOW, "order accumulators could overflow i64 (frames x molecules >= 2^63 / 1e6)");

    // ---- leaflet assignment rows of this batch; row 0 = assignment carried over from earlier
    // batches (AssignedLeaflets::local, leaflets.rs:1371-1380), rows 1.. = assignment frames here
    const bool leaflets = lf.method != GORDER_LEAFLETS_NONE;
    size_t n_new_rows = 0;
    bool spec = false;
    std::vector<uint32_t> spec_aframes;
    // argument errors come before the first kernel of the batch is queued (a batch that fails later leaves through
    // abort_batch below, so that a key its kernels raised is not committed under the next batch's ordinal)
    if (h->manual_frames) {
        if (h->manual_frames != n_frames) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "gorder_hip_set_normals: frame count differs from the batch");
        if (!p.direct.empty()) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "manual normals: a bond spans more than the LDS window");
    }
    auto abort_batch = [&](int status) {
        hipLaunchKernelGGL(k_batch_abort, dim3(1), dim3(1), 0, h->stream, h->d_err);
        (void)hipGetLastError();
        (void)timing_mark(h, nullptr);
        return status;
    };
    if (leaflets) {
        std::vector<uint32_t> arow(n_frames), aframes;
        uint32_t cur = 0;
        bool have = h->have_assignment;
        uint64_t last_assign_frame = h->assignment_frame;
        for (uint32_t f = 0; f < n_frames; f++) {
            if (lf.method != GORDER_LEAFLETS_MANUAL && should_assign(lf.frequency, frame_index[f])) {
                aframes.push_back(f);
                cur = (uint32_t)aframes.size();
                have = true;
                last_assign_frame = frame_index[f];
            }
            if (!have) return fail(h, GORDER_ERR_LEAFLETS_NOT_PRIMED, "no leaflet assignment for the first frame");
            arow[f] = cur;
        }
        // One read for global leaflets + order parameters: every frame of the batch is an assignment frame, an earlier
        // assignment exists (row 0), nothing but the plain order kernel runs.  The order kernel then routes by row 0.
        // (per-frame rows too where they come out of the tiled kernel: k_bonds_tiled_tw, the default cosine)
        const bool tw_tiled = h->extra.tw && !(h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS) && h->d_item_run &&
                              h->frames_per_stage == (int)kRecFrames && !env_flag("GORDER_HIP_TW_GATHER");
        spec = h->spec_enabled && lf.method == GORDER_LEAFLETS_GLOBAL && h->have_assignment && aframes.size() == n_frames &&
               !h->extra.maps && (!h->extra.tw || tw_tiled) && !h->extra.geom_kind && !h->dyn && !h->manual_frames &&
               !h->use_gather && !p.tiles.empty();
        if (spec) {
            spec_poll(h, false);
            const uint32_t slot = (uint32_t)(h->spec_batches % gorder_hip_handle::kSpecRing);
            if (h->spec_ring_frames[slot]) (void)hipEventSynchronize(h->spec_counters_copied[slot]), spec_poll(h, false);
            spec = h->spec_enabled;
        }
        if (spec) std::fill(arow.begin(), arow.end(), 0u);
        const size_t rows = aframes.size() + 1;            // (a speculative batch: row 0 and every frame's exact sides)
        if (rows > h->aflags_rows) {
            uint8_t *nb = nullptr;
            const size_t nrows = rows + rows / 4;
            HIP_TRY(h, hipMalloc((void **)&nb, nrows * (size_t)p.n_mol_total));
            if (h->d_aflags) {
                // stream-ordered: the handle's stream is non-blocking, a null-stream copy would not be
                HIP_TRY(h, hipMemcpyAsync(nb, h->d_aflags, p.n_mol_total, hipMemcpyDeviceToDevice, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                HIP_TRY(h, hipFree(h->d_aflags));
            }
            h->d_aflags = nb;
            h->aflags_rows = nrows;
        }
        const size_t cap_before = h->arow_cap;
        if ((st = ensure(h, &h->d_arow, &h->arow_cap, n_frames)) != GORDER_OK) return st;
        if (h->arow_cap != cap_before) h->up_arow_at = nullptr;        // a new allocation holds nothing yet
        if (h->up_arow_at != h->d_arow || h->up_arow != arow) {
            HIP_TRY(h, hipMemcpyAsync(h->d_arow, arow.data(), n_frames * sizeof(uint32_t), hipMemcpyHostToDevice,
                                      h->stream));
            h->up_arow = arow;
            h->up_arow_at = h->d_arow;
        }
        if (spec) spec_aframes = aframes;
        else if ((st = run_leaflets(h, d_xyz, d_box, aframes, 1)) != GORDER_OK) return abort_batch(st);
        h->have_assignment = true;
        h->assignment_frame = last_assign_frame;
        n_new_rows = spec ? 0 : aframes.size();
    }
    // (check_box: k_batch_end, at the end of the batch)
    if (h->extra.tw && h->n_frames + n_frames > h->tw_cap) {   // grow the per-frame rows (timewise.rs:183-186)
        const size_t row = 3 * (size_t)p.n_acc;
        const uint64_t ncap = (h->n_frames + n_frames) * 2;
        unsigned long long *ns = nullptr, *nc = nullptr;
        HIP_TRY(h, hipMalloc((void **)&ns, ncap * row * sizeof(unsigned long long)));
        HIP_TRY(h, hipMalloc((void **)&nc, ncap * row * sizeof(unsigned long long)));
        // everything on the handle's own (non-blocking) stream: a null-stream memset / copy is NOT ordered
        // with the kernels that follow and left stale rows behind (seen as a wrong error estimate)
        HIP_TRY(h, hipMemsetAsync(ns, 0, ncap * row * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, hipMemsetAsync(nc, 0, ncap * row * sizeof(unsigned long long), h->stream));
        if (h->d_tw_sums) {
            HIP_TRY(h, hipMemcpyAsync(ns, h->d_tw_sums, h->n_frames * row * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(nc, h->d_tw_cnts, h->n_frames * row * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipFree(h->d_tw_sums));
            HIP_TRY(h, hipFree(h->d_tw_cnts));
        }
        h->d_tw_sums = ns; h->d_tw_cnts = nc; h->tw_cap = ncap;
    }
    FrameArgs a{};
    a.xyz = d_xyz; a.box9 = d_box; a.n_atoms = p.n_atoms; a.n_frames = n_frames;
    a.pbc = pbc ? 1 : 0;
    a.nx = h->tables.normal[0]; a.ny = h->tables.normal[1]; a.nz = h->tables.normal[2]; a.n2 = h->n2; a.n2sq = h->n2sq;
    a.leaflets = leaflets ? 1 : 0; a.aflags = h->d_aflags; a.arow = h->d_arow; a.n_mol_total = p.n_mol_total;
    a.acc = h->d_acc; a.rep = h->d_rep; a.n_rep = h->n_rep; a.n_acc = p.n_acc; a.err = h->d_err;
    h->manual_active = false;
    if (h->manual_frames) {   // normals the host supplied for exactly this batch (arguments checked above)
        const size_t n4 = (size_t)n_frames * p.n_mol_total;
        if ((st = ensure(h, &h->d_dyn_normals, &h->dyn_normals_cap, n4)) != GORDER_OK) return abort_batch(st);
        if (hipStreamSynchronize(h->stream) != hipSuccess ||   // earlier batches may still read the buffer
            hipMemcpy(h->d_dyn_normals, h->manual_normals.data(), n4 * 4 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return abort_batch(fail(h, GORDER_ERR_DEVICE, "manual normals: copy to the device failed"));
        h->manual_active = true;
        h->manual_frames = 0;
    }
    const uint32_t n_tiles_all = (uint32_t)p.tiles.size();
    if (spec) {
        if ((st = ensure(h, &h->d_mom, &h->mom_cap, (size_t)n_frames * n_tiles_all)) != GORDER_OK) return abort_batch(st);
        if ((st = ensure(h, &h->d_head_z, &h->head_z_cap, (size_t)n_frames * p.n_mol_total)) != GORDER_OK) return abort_batch(st);
        a.own = h->d_own; a.mom = h->d_mom; a.mom_dim = (int)lf.normal_dim;
        a.own_head_begin = h->d_own_head_begin; a.own_heads = h->d_own_heads; a.head_z = h->d_head_z;
        h->spec_now = true;
    }
    st = launch_orders(h, a);
    h->spec_now = false;
    h->manual_active = false;
    if (st != GORDER_OK) return abort_batch(st);
    h->rep_dirty = true;
    if (spec) {
        // the exact centres from the order kernel's sums, the exact kernel for the frames they cannot vouch for, the sides
        // of every (frame, molecule) against the prediction, the mispredicted ones moved
        size_t cap_c = h->spec_frames_cap, cap_o = h->spec_frames_cap;
        if ((st = ensure(h, &h->d_spec_center, &cap_c, n_frames)) != GORDER_OK) return abort_batch(st);
        if ((st = ensure(h, &h->d_spec_ok, &cap_o, n_frames)) != GORDER_OK) return abort_batch(st);
        h->spec_frames_cap = std::min(cap_c, cap_o);
        uint32_t *cnt = h->d_spec_counters + 2u * (uint32_t)(h->spec_batches & 1u);
        uint32_t *cnt_next = h->d_spec_counters + 2u * (uint32_t)((h->spec_batches + 1u) & 1u);
        SpecArgs sa{};
        sa.xyz = d_xyz; sa.box9 = d_box; sa.n_atoms = p.n_atoms; sa.n_frames = n_frames; sa.n_tiles = n_tiles_all;
        sa.n_mol_total = p.n_mol_total; sa.n_membrane = lf.n_membrane; sa.dim = lf.normal_dim; sa.flip = lf.flip ? 1 : 0;
        sa.pbc = pbc ? 1 : 0; sa.mom = h->d_mom; sa.head_z = h->d_head_z; sa.center = h->d_spec_center; sa.ok = h->d_spec_ok;
        sa.aflags = h->d_aflags; sa.adist = h->d_adist; sa.counters = cnt; sa.counters_next = cnt_next;
        const uint32_t ring_slot = (uint32_t)(h->spec_batches % gorder_hip_handle::kSpecRing);
        sa.host_counters = h->h_spec_counters + 2u * ring_slot; sa.err = h->d_err;
        (void)timing_mark(h, "k_spec_check + k_spec_fixup");
        hipLaunchKernelGGL(k_spec_check, dim3(std::min<uint32_t>(n_frames, 2048u)), dim3(256), 0, h->stream, sa);
        if ((st = run_leaflets(h, d_xyz, d_box, spec_aframes, 1, h->d_spec_ok)) != GORDER_OK) return abort_batch(st);
        (void)timing_mark(h, "k_spec_check + k_spec_fixup");
        const uint32_t fix_grid = (uint32_t)std::min<uint64_t>(((uint64_t)n_frames * p.n_mol_total + 63u) / 64u, 4096u);
        if (h->extra.tw)
            hipLaunchKernelGGL((k_spec_fixup<false, true>), dim3(fix_grid), dim3(64), 0, h->stream, a, sa, h->d_spec_mol_begin,
                               h->d_spec_samples, h->d_tw_sums, h->d_tw_cnts, (uint64_t)h->n_frames);
        else if (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS)
            hipLaunchKernelGGL((k_spec_fixup<true, false>), dim3(fix_grid), dim3(64), 0, h->stream, a, sa, h->d_spec_mol_begin,
                               h->d_spec_samples, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint64_t)0);
        else
            hipLaunchKernelGGL((k_spec_fixup<false, false>), dim3(fix_grid), dim3(64), 0, h->stream, a, sa, h->d_spec_mol_begin,
                               h->d_spec_samples, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint64_t)0);
        hipLaunchKernelGGL(k_spec_finish, dim3(1), dim3(256), 0, h->stream, sa);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(h->spec_counters_copied[ring_slot], h->stream));
        h->spec_ring_frames[ring_slot] = n_frames;
        h->spec_batches++;
    }
    if (n_new_rows) {   // newest assignment becomes the carry row of the next batch
        HIP_TRY(h, hipMemcpyAsync(h->d_aflags, h->d_aflags + n_new_rows * (size_t)p.n_mol_total, p.n_mol_total,
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    // the batch's error key (if any) becomes the run's unless an earlier batch had one
    {
        gorder_hip_handle::BatchLog rec{h->n_submits, frame_index[0], n_frames > 1 ? frame_index[1] - frame_index[0] : 0, {}};
        for (uint32_t f = 1; f < n_frames; f++)
            if (frame_index[f] != rec.first + (uint64_t)f * rec.stride) { rec.list.assign(frame_index, frame_index + n_frames); break; }
        if (h->batch_log.size() >= gorder_hip_handle::kBatchLog) h->batch_log.pop_front();
        h->batch_log.push_back(std::move(rec));
        // check_box, total_frames and the batch's error key in one launch behind the batch's kernels
        TIMING_MARK(h, "k_batch_end");
        hipLaunchKernelGGL(k_batch_end, dim3(pbc ? std::min<uint32_t>((n_frames + 255u) / 256u, 256u) : 1u), dim3(256), 0, h->stream,
                           pbc ? d_box : nullptr, n_frames, h->d_err,
                           h->d_acc + 4 * (size_t)p.n_acc, (unsigned long long)h->n_submits, h->decoder_key);
        HIP_TRY(h, hipGetLastError());
        TIMING_MARK(h, nullptr);          // the batch's chain of timed segments ends here
        if (h->timing_on) h->timing_launches++;
        h->n_submits++;
    }
    h->n_frames += n_frames;
    return GORDER_OK;
}

int gorder_hip_submit_host(gorder_hip_handle *h, const float *xyz, const float *box, const uint64_t *frame_index,
                           uint32_t n_frames) {
    if (!h || !xyz || !frame_index) return GORDER_ERR_INVALID_ARGUMENT;
    if (n_frames == 0) return GORDER_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int st;
    const size_t nx = (size_t)n_frames * h->plan.n_atoms * 3u, nb = (size_t)n_frames * 9u;
    if (!h->copy_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(h, hipEventCreateWithFlags(&h->stage_copied[k], hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->stage_computed[k], hipEventDisableTiming));
        }
    }
    const int b = h->stage_next;
    h->stage_next ^= 1;
    // this buffer was last read by the kernels of two calls ago (normally long finished); the kernels of the
    // previous call keep running on the other buffer while this copy is in flight
    if (h->stage_used[b]) HIP_TRY(h, hipEventSynchronize(h->stage_computed[b]));
    if ((st = ensure(h, &h->d_stage_xyz[b], &h->stage_xyz_cap[b], nx)) != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage_xyz[b], xyz, nx * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    if (box) {
        if ((st = ensure(h, &h->d_stage_box[b], &h->stage_box_cap[b], nb)) != GORDER_OK) return st;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_box[b], box, nb * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    }
    HIP_TRY(h, hipEventRecord(h->stage_copied[b], h->copy_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->stage_copied[b], 0));
    st = gorder_hip_submit_device(h, h->d_stage_xyz[b], box ? h->d_stage_box[b] : nullptr, frame_index, n_frames);
    if (st != GORDER_OK) return st;
    HIP_TRY(h, hipEventRecord(h->stage_computed[b], h->stream));
    h->stage_used[b] = true;
    // the caller owns its buffer again when this returns
    HIP_TRY(h, hipEventSynchronize(h->stage_copied[b]));
    return GORDER_OK;
}

int gorder_hip_prime_leaflets(gorder_hip_handle *h, const float *d_xyz, const float *d_box, uint64_t frame_index) {
    if (!h || !d_xyz) return GORDER_ERR_INVALID_ARGUMENT;
    const gorder_leaflets_t &lf = h->tables.leaflets;
    if (lf.method == GORDER_LEAFLETS_NONE || lf.method == GORDER_LEAFLETS_MANUAL) return GORDER_ERR_INVALID_ARGUMENT;
    if (h->tables.handle_pbc && !d_box) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "box required when handle_pbc = 1");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->tables.handle_pbc) {   // the priming frame goes through check_box like any analysed frame (common.rs:186-198)
        hipLaunchKernelGGL(k_check_box, dim3(1), dim3(256), 0, h->stream, d_box, 1u, h->d_err);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->aflags_rows < 2) {
        uint8_t *nb = nullptr;
        HIP_TRY(h, hipMalloc((void **)&nb, 2 * (size_t)h->plan.n_mol_total));
        if (h->d_aflags) HIP_TRY(h, hipFree(h->d_aflags));
        h->d_aflags = nb;
        h->aflags_rows = 2;
    }
    std::vector<uint32_t> aframes(1, 0);
    const int st = run_leaflets(h, d_xyz, d_box, aframes, 0);
    (void)timing_mark(h, nullptr);      // (the priming frame's kernels are a chain of their own: nothing stays open until the next submit)
    if (st != GORDER_OK) return st;
    h->have_assignment = true;
    h->assignment_frame = frame_index;
    return GORDER_OK;
}

int gorder_hip_set_manual_leaflets(gorder_hip_handle *h, const uint8_t *flags, uint64_t frame_index) {
    if (!h || !flags) return GORDER_ERR_INVALID_ARGUMENT;
    if (h->tables.leaflets.method != GORDER_LEAFLETS_MANUAL) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const uint32_t n = h->plan.n_mol_total;
    std::vector<uint8_t> tmp(n);
    for (uint32_t i = 0; i < n; i++) tmp[i] = (uint8_t)((flags[i] & 1) ^ (h->tables.leaflets.flip ? 1 : 0));
    if (h->aflags_rows < 1) {
        HIP_TRY(h, hipMalloc((void **)&h->d_aflags, 2 * (size_t)n));
        h->aflags_rows = 2;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->d_aflags, tmp.data(), n, hipMemcpyHostToDevice));
    h->have_assignment = true;
    h->assignment_frame = frame_index;
    return GORDER_OK;
}

int gorder_hip_synchronize(gorder_hip_handle *h) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_device_error(h);
}

int gorder_hip_finish(gorder_hip_handle *h, int64_t *sums, uint64_t *counts, int64_t *map_sums,
                      uint64_t *map_counts, uint64_t *n_frames_analyzed) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    int st = fold_replicas(h);
    if (st != GORDER_OK) return st;
    if ((st = fold_maps(h)) != GORDER_OK) return st;
    st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const uint32_t n = h->plan.n_acc;
    std::vector<unsigned long long> raw(4 * (size_t)n + 1);
    HIP_TRY(h, hipMemcpy(raw.data(), h->d_acc, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const bool lf = h->tables.leaflets.method != GORDER_LEAFLETS_NONE;
    for (uint32_t s = 0; s < n; s++) {
        const int64_t tot = (int64_t)raw[s], up = (int64_t)raw[n + s];
        const uint64_t ctot = raw[2 * (size_t)n + s], cup = raw[3 * (size_t)n + s];
        if (sums) {
            sums[s] = tot;
            sums[n + s] = lf ? up : 0;
            sums[2 * (size_t)n + s] = lf ? tot - up : 0;   // every sample is upper or lower (bond.rs:199-213)
        }
        if (counts) {
            counts[s] = ctot;
            counts[n + s] = lf ? cup : 0;
            counts[2 * (size_t)n + s] = lf ? ctot - cup : 0;
        }
    }
    if (n_frames_analyzed) *n_frames_analyzed = raw[4 * (size_t)n];
    if (h->extra.maps) {
        const size_t nmap = 3 * (size_t)n * h->map_nx * h->map_ny;
        if (map_sums) HIP_TRY(h, hipMemcpy(map_sums, h->d_map_sums, nmap * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (map_counts) HIP_TRY(h, hipMemcpy(map_counts, h->d_map_cnts, nmap * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return GORDER_OK;
}

int gorder_hip_set_normals(gorder_hip_handle *h, const float *normals, uint32_t n_frames) {
    if (!h || !normals || n_frames == 0) return GORDER_ERR_INVALID_ARGUMENT;
    const size_t n = (size_t)n_frames * h->plan.n_mol_total;
    h->manual_normals.resize(4 * n);
    for (size_t i = 0; i < n; i++) {
        h->manual_normals[4 * i + 0] = normals[3 * i + 0];
        h->manual_normals[4 * i + 1] = normals[3 * i + 1];
        h->manual_normals[4 * i + 2] = normals[3 * i + 2];
        h->manual_normals[4 * i + 3] = 3.0f;     // "enough points": the sample kernels treat it like a computed normal
    }
    h->manual_frames = n_frames;
    return GORDER_OK;
}

int gorder_hip_normals(gorder_hip_handle *h, float *normals, uint32_t *n_points) {
    if (!h || !h->dyn) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const uint32_t n_mol = h->plan.n_mol_total;
    if (h->last_normals.size() < 4 * (size_t)n_mol) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "no frame submitted yet");
    for (uint32_t m = 0; m < n_mol; m++) {
        if (normals) for (int d = 0; d < 3; d++) normals[3 * (size_t)m + d] = h->last_normals[4 * (size_t)m + d];
        if (n_points) n_points[m] = (uint32_t)h->last_normals[4 * (size_t)m + 3];
    }
    return GORDER_OK;
}

int gorder_hip_timewise(gorder_hip_handle *h, int64_t *tw_sums, uint64_t *tw_counts, uint64_t capacity_frames) {
    if (!h || !h->extra.tw || !tw_sums || !tw_counts) return GORDER_ERR_INVALID_ARGUMENT;
    if (capacity_frames < h->n_frames) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const size_t n = h->n_frames * 3 * (size_t)h->plan.n_acc;
    if (n) {
        HIP_TRY(h, hipMemcpy(tw_sums, h->d_tw_sums, n * sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(tw_counts, h->d_tw_cnts, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        // the kernels keep total and upper only (a third fewer atomics): the lower leaflet's rows are their difference
        if (h->tables.leaflets.method != GORDER_LEAFLETS_NONE) {
            const size_t na = h->plan.n_acc;
            for (uint64_t f = 0; f < h->n_frames; f++) {
                int64_t *s = tw_sums + f * 3 * na;
                uint64_t *c = tw_counts + f * 3 * na;
                for (size_t k = 0; k < na; k++) { s[2 * na + k] = s[k] - s[na + k]; c[2 * na + k] = c[k] - c[na + k]; }
            }
        }
    }
    return GORDER_OK;
}

int gorder_hip_leaflets(gorder_hip_handle *h, uint8_t *flags, uint64_t *assignment_frame) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    if (!h->have_assignment) return GORDER_ERR_LEAFLETS_NOT_PRIMED;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    if (flags) HIP_TRY(h, hipMemcpy(flags, h->d_aflags, h->plan.n_mol_total, hipMemcpyDeviceToHost));
    if (assignment_frame) *assignment_frame = h->assignment_frame;
    return GORDER_OK;
}

int gorder_hip_leaflet_distances(gorder_hip_handle *h, float *dist) {
    if (!h || !dist || !h->d_adist) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpy(dist, h->d_adist, sizeof(float) * h->plan.n_mol_total, hipMemcpyDeviceToHost));
    return GORDER_OK;
}

int gorder_hip_accumulators_device(gorder_hip_handle *h, void **d_ptr, uint64_t *n_u64) {
    if (!h || !d_ptr || !n_u64) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    {   // the packed block must be complete before a collective reads it (stream-ordered)
        const int st = fold_replicas(h);
        if (st != GORDER_OK) return st;
    }
    *d_ptr = h->d_acc;
    *n_u64 = h->acc_words;
    return GORDER_OK;
}

int gorder_hip_export_maps(gorder_hip_handle *h, void *d_sums, void *d_counts, uint64_t n_u64) {
    if (!h || !h->extra.maps || !d_sums || !d_counts) return GORDER_ERR_INVALID_ARGUMENT;
    const uint64_t n = 3ull * h->plan.n_acc * h->map_nx * h->map_ny;
    if (n_u64 < n) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const int st = fold_maps(h);
    if (st != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(d_sums, h->d_map_sums, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_counts, h->d_map_cnts, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_device_error(h);
}

int gorder_hip_bind_accumulators(gorder_hip_handle *h, void *d_ptr, uint64_t n_u64) {
    if (!h || !d_ptr || n_u64 < h->acc_words || ((uintptr_t)d_ptr & 7u)) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int st = fold_replicas(h);
        if (st != GORDER_OK) return st;
    }
    HIP_TRY(h, hipMemcpyAsync(d_ptr, h->d_acc, h->acc_words * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->acc_external) HIP_TRY(h, hipFree(h->d_acc));
    h->d_acc = (unsigned long long *)d_ptr;
    h->acc_external = true;
    return GORDER_OK;
}

// ---- a fresh SystemTopology on the same tables --------------------------------------------------------------------
int gorder_hip_reset(gorder_hip_handle *h) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const Plan &p = h->plan;
    HIP_TRY(h, hipMemsetAsync(h->d_acc, 0, h->acc_words * sizeof(unsigned long long), h->stream));
    if (h->d_rep) HIP_TRY(h, hipMemsetAsync(h->d_rep, 0, (size_t)h->n_rep * 4u * p.n_acc * sizeof(unsigned long long), h->stream));
    h->rep_dirty = false;
    if (h->extra.maps) {
        const size_t nmap = 3 * (size_t)p.n_acc * h->map_nx * h->map_ny;
        const size_t npk = (h->tables.leaflets.method != GORDER_LEAFLETS_NONE ? 2 : 1) * (nmap / 3);
        HIP_TRY(h, hipMemsetAsync(h->d_map_sums, 0, nmap * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_map_cnts, 0, nmap * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_map_packed, 0, npk * sizeof(unsigned long long), h->stream));
        h->map_pending = 0;
    }
    if (h->d_tw_sums) {
        const size_t n = h->tw_cap * 3 * (size_t)p.n_acc * sizeof(unsigned long long);
        HIP_TRY(h, hipMemsetAsync(h->d_tw_sums, 0, n, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_tw_cnts, 0, n, h->stream));
    }
    HIP_TRY(h, hipMemsetAsync(h->d_err, 0xff, kErrWords * sizeof(uint32_t), h->stream));
    h->batch_log.clear();
    h->n_frames = 0;
    h->have_assignment = false;
    h->assignment_frame = 0;
    h->manual_frames = 0;
    h->err_index = 0;
    h->err_msg.clear();
    // a new run speculates again (and starts its statistics anew); the batches still in flight are waited for first —
    // their counters would otherwise be booked on the new run
    if (h->d_own) {
        spec_poll(h, true);
        h->spec_enabled = true;
        h->spec_batches = h->spec_fixed = h->spec_exact_frames = 0;
        HIP_TRY(h, hipMemsetAsync(h->d_spec_counters, 0, 4 * sizeof(uint32_t), h->stream));
    }
    // likewise k_local_decide: a new run tries the bound again (a report still in flight is read, and dropped)
    if (h->d_lsummary) {
        if (h->lsummary_pending) { (void)hipEventSynchronize(h->lsummary_written); h->lsummary_pending = false; }
        h->decide_pause = 0;
        h->decide_submits = h->decide_paused_submits = 0;
        h->h_lsummary[0] = h->h_lsummary[1] = 0u;
        HIP_TRY(h, hipMemsetAsync(h->d_lsummary, 0, 2 * sizeof(uint32_t), h->stream));
    }
    return GORDER_OK;
}

int gorder_hip_comm_unique_id(uint8_t id[128]) {
    if (!id) return GORDER_ERR_INVALID_ARGUMENT;
    RcclApi *api = rccl_api(nullptr);
    if (!api) return GORDER_ERR_DEVICE;
    UniqueId u;
    if (api->get_unique_id(&u) != 0) return GORDER_ERR_DEVICE;
    memcpy(id, u.internal, 128);
    return GORDER_OK;
}

int gorder_hip_comm_create(gorder_hip_handle *h, const uint8_t id[128], int n_ranks, int rank, void **comm_out) {
    if (!h || !id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return GORDER_ERR_INVALID_ARGUMENT;
    *comm_out = nullptr;
    std::string why;
    RcclApi *api = rccl_api(&why);
    if (!api) return fail(h, GORDER_ERR_DEVICE, why);
    HIP_TRY(h, hipSetDevice(h->device));
    UniqueId u;
    memcpy(u.internal, id, 128);
    const int rc = api->comm_init_rank(comm_out, n_ranks, u, rank);
    if (rc != 0) return fail(h, GORDER_ERR_DEVICE, rccl_error(api, "ncclCommInitRank", rc));
    return GORDER_OK;
}

void gorder_hip_comm_destroy(void *comm) {
    RcclApi *api = rccl_api(nullptr);
    if (api && comm) (void)api->comm_destroy(comm);
}

int gorder_hip_allreduce(gorder_hip_handle *h, void *nccl_comm) {
    if (!h || !nccl_comm) return GORDER_ERR_INVALID_ARGUMENT;
    std::string why;
    RcclApi *api = rccl_api(&why);
    if (!api) return fail(h, GORDER_ERR_DEVICE, why);
    HIP_TRY(h, hipSetDevice(h->device));
    int st = fold_replicas(h);
    if (st != GORDER_OK) return st;
    if ((st = fold_maps(h)) != GORDER_OK) return st;
    // everything that SystemTopology::add sums (topology/mod.rs:236-254) in one group on the handle's stream: the packed
    // accumulator block (order sums, counts, total_frames) and, with ordermaps, the folded i64 / u64 maps (Map::add,
    // ordermap.rs:116-138).  Integer sums: the result is bit-identical whatever the ring order.
    int rc = api->group_start();
    if (rc == 0) rc = api->all_reduce(h->d_acc, h->d_acc, h->acc_words, kNcclInt64, kNcclSum, nccl_comm, h->stream);
    if (rc == 0 && h->extra.maps) {
        const size_t nmap = 3 * (size_t)h->plan.n_acc * h->map_nx * h->map_ny;
        rc = api->all_reduce(h->d_map_sums, h->d_map_sums, nmap, kNcclInt64, kNcclSum, nccl_comm, h->stream);
        if (rc == 0) rc = api->all_reduce(h->d_map_cnts, h->d_map_cnts, nmap, kNcclInt64, kNcclSum, nccl_comm, h->stream);
    }
    const int rc_end = api->group_end();
    if (rc == 0) rc = rc_end;
    if (rc != 0) return fail(h, GORDER_ERR_DEVICE, rccl_error(api, "ncclAllReduce", rc));
    return GORDER_OK;
}

uint64_t gorder_hip_last_error_index(const gorder_hip_handle *h) { return h ? h->err_index : 0; }
uint64_t gorder_hip_last_error_frame(const gorder_hip_handle *h) { return h ? h->err_frame : 0; }
const char *gorder_hip_kernel_time_names(const gorder_hip_handle *h) { return h ? h->timed_kernels.c_str() : ""; }
const char *gorder_hip_last_error_message(const gorder_hip_handle *h) { return h ? h->err_msg.c_str() : ""; }

int gorder_hip_kernel_time(gorder_hip_handle *h, double *ms, uint64_t *launches, int reset) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    h->timing_on = true;   // the first call switches the events on (submits before it are not timed)
    while (!h->timing_pending.empty()) {
        const int st = timing_drain_oldest(h);
        if (st != GORDER_OK) return st;
    }
    double total = 0.0;
    for (double v : h->timing_label_ms) total += v;
    if (ms) *ms = total;
    if (launches) *launches = h->timing_launches;
    if (reset) {
        h->timing_labels.clear(); h->timing_label_ms.clear(); h->timing_label_n.clear();
        h->timed_kernels.clear();
        h->timing_launches = 0;
        h->timing_open_label = -1;       // (a segment left open — there is none between submits — would name a label that is gone)
    }
    return GORDER_OK;
}

int gorder_hip_kernel_time_group(gorder_hip_handle *h, uint32_t index, const char **name, double *ms, uint64_t *segments) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    while (!h->timing_pending.empty()) {
        const int st = timing_drain_oldest(h);
        if (st != GORDER_OK) return st;
    }
    if (index >= h->timing_labels.size()) return GORDER_ERR_INVALID_ARGUMENT;
    if (name) *name = h->timing_labels[index].c_str();
    if (ms) *ms = h->timing_label_ms[index];
    if (segments) *segments = h->timing_label_n[index];
    return GORDER_OK;
}

// ---- XTC frames decompressed on the device (kernels_xtc.h) ---------------------------------------------------------
// checkpoints a batch needs between its two kernels
size_t xtc_checkpoints(uint32_t n_frames, uint32_t n_stop) {
    return (size_t)n_frames * ((std::max(n_stop, 1u) + kXtcChunk - 1u) / kXtcChunk + 1u);
}
int xtc_decode_on(gorder_hip_handle *h, hipStream_t stream, const uint8_t *d_blob, uint64_t blob_bytes,
                  const gorder_xtc_frame_t *d_frames, uint32_t n_frames, uint32_t n_atoms_file, const int32_t *d_slot_of,
                  uint32_t n_stop, float *d_xyz, uint32_t n_atoms_out, uint32_t *d_stat = nullptr, uint32_t *d_short = nullptr,
                  uint32_t *d_err_key = nullptr, XtcCheckpoint *d_cp = nullptr) {
    if (!h || !d_blob || !d_frames || !d_xyz || blob_bytes < 64 || (reinterpret_cast<uintptr_t>(d_blob) & 63u) != 0 || n_atoms_file == 0 || n_atoms_out == 0 ||
        n_stop > n_atoms_file || (!d_slot_of && n_atoms_out < n_stop))
        return fail(h, GORDER_ERR_INVALID_ARGUMENT, "gorder_hip_xtc_decode: bad arguments");
    if (n_frames == 0) return GORDER_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // two kernels (kernels_xtc.h): where the chunks of ~kXtcChunk atoms start in every frame's stream (one lane per
    // frame, no decoding), then the chunks (one lane per frame and chunk).  The checkpoints between them live in the
    // caller's buffer (a slot of gorder_hip_run_trajectory has its own: several batches decode at once) or the handle's.
    const uint32_t n_chunks = (std::max(n_stop, 1u) + kXtcChunk - 1u) / kXtcChunk;
    if (!d_cp) {
        const int st = ensure(h, &h->d_xtc_cp, &h->xtc_cp_cap, xtc_checkpoints(n_frames, n_stop));
        if (st != GORDER_OK) return st;
        d_cp = h->d_xtc_cp;
    }
    hipLaunchKernelGGL(k_xtc_scan, dim3((n_frames + 3u) / 4u), dim3(256), 0, stream, d_blob, (unsigned long long)blob_bytes,
                       d_frames, n_frames, n_atoms_file, d_slot_of, n_stop, d_xyz, n_atoms_out,
                       d_err_key ? d_err_key : h->d_err, d_stat, d_short, d_cp, n_chunks);
    const unsigned long long items = (unsigned long long)n_frames * n_chunks;
    if (items + 63ull > 64ull * 0x7fffffffull) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "gorder_hip_xtc_decode: batch too large");
    hipLaunchKernelGGL(k_xtc_chunks, dim3((uint32_t)((items + 63ull) / 64ull)), dim3(64), 0, stream, d_blob,
                       (unsigned long long)blob_bytes, d_frames, n_frames, n_atoms_file, d_slot_of, n_stop, d_xyz, n_atoms_out,
                       d_cp, n_chunks);
    HIP_TRY(h, hipGetLastError());
    return GORDER_OK;
}
int gorder_hip_xtc_decode(gorder_hip_handle *h, const uint8_t *d_blob, uint64_t blob_bytes,
                          const gorder_xtc_frame_t *d_frames, uint32_t n_frames, uint32_t n_atoms_file,
                          const int32_t *d_slot_of, uint32_t n_stop, float *d_xyz, uint32_t n_atoms_out) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    return xtc_decode_on(h, h->stream, d_blob, blob_bytes, d_frames, n_frames, n_atoms_file, d_slot_of, n_stop, d_xyz,
                         n_atoms_out);
}

}  // extern "C"

#include "trajectory_driver.h"
