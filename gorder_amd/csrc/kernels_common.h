// kernels_common.h — shared kernel arguments, error record, check_box, accumulator folds, ordermap words.
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

constexpr int kFramesPerStage = 4;   // G: frames staged in LDS per barrier pair (= waves per block)
constexpr uint32_t kErrWords = 8;    // device error record, four 64-bit words: [0] key of the batch in flight (raise_error),
                                     // [1] the FIRST batch key that was not kErrNone, [2] that batch's ordinal (k_err_commit)

struct FrameArgs {
    const float *xyz;        // [n_frames][n_atoms][3]
    const float *box9;       // [n_frames][9]
    uint32_t n_atoms;
    uint32_t n_frames;       // end of the frame range this launch covers
    uint32_t frame0;         // its begin (only the scatter kernels launch sub-ranges; 0 elsewhere)
    uint32_t frames_per_chunk;
    int pbc;
    float nx, ny, nz, n2, n2sq;   // static normal, its norm and squared norm
    int leaflets;            // 0/1
    const uint8_t *aflags;   // [rows][n_mol_total]
    const uint32_t *arow;    // [n_frames] assignment row of each frame
    uint32_t n_mol_total;
    unsigned long long *acc; // [4][n_acc]: sum_total, sum_upper, cnt_total, cnt_upper
    unsigned long long *rep; // [n_rep][4][n_acc] replicas the tiled kernels add into (folded into acc later)
    uint32_t n_rep;
    uint32_t n_acc;
    uint32_t *err;
    // k_bonds_tiled<..., MOM> only (one read for global leaflets + order parameters):
    const uint2 *own;        // [n_tiles] the atoms of the membrane group each tile sums: [x, y)
    float4 *mom;             // [n_frames][n_tiles] (sum z, sum z^2, min z, max z) of the owned atoms
    int mom_dim;             // which coordinate z is (the leaflets' normal)
    const uint32_t *own_head_begin;   // [n_tiles + 1] the molecules whose head atom the tile owns (Plan::own_heads)
    const uint2 *own_heads;           // (window-relative atom, molecule)
    float *head_z;                    // [n_frames][n_mol_total] their normal coordinate, handed on to k_spec_check
};

// ---- device error record --------------------------------------------------------------------------
// ONE 64-bit key, lowered with atomicMin: the smallest key of a batch wins, and keys ascend in the order in which
// the reference's single-threaded walk would meet the errors — frame by frame (common.rs:201-235: box check, then
// the system-level leaflet step), then molecule type by molecule type (molecule.rs:54-95: leaflet assignment of the
// type, then bond type -> molecule, bond.rs:406-417 / united atom -> molecule, uaorder.rs:400-437).  Which error a
// batch reports therefore does not depend on the launch geometry or on which wave gets there first.
//   [63]    0 = an error of the reference (errors.rs:121-168), 1 = GORDER_ERR_BOX_RANGE: the library's own range
//           check (where the reference would spin) never hides an error the reference would have returned
//   [62:40] frame in batch (clamped to 2^23 - 1)      [39:38] stage (ErrStage)
//   [37:24] accumulator slot (first slot of the type for its leaflet assignment; clamped to 2^14 - 1)
//   [23]    0 = leaflet assignment of the type, 1 = order sample
//   [22:6]  global molecule id (clamped to 2^17 - 1)   [5:4] detail (which atom of the sample / cloud size)
//   [3:0]   code: gorder_status_t 1..7, 8 = GORDER_ERR_BOX_RANGE, 9 = GORDER_ERR_TRAJECTORY_FORMAT (k_xtc_scan)
// The payload of gorder_hip_last_error_index (an atom index) is resolved on the host from (slot, molecule, detail).
constexpr unsigned long long kErrNone = ~0ull;
enum ErrStage : uint32_t { kStageBox = 0, kStageSystem = 1, kStageTypes = 2, kStageEnd = 3 };
__device__ __forceinline__ void raise_error(uint32_t *err, uint32_t code, uint32_t frame, uint32_t stage,
                                            uint32_t slot = 0, uint32_t sample = 0, uint32_t mol = 0,
                                            uint32_t detail = 0) {
    const unsigned long long key =
        ((unsigned long long)(code == GORDER_ERR_BOX_RANGE ? 1u : 0u) << 63) |
        ((unsigned long long)min(frame, 0x7fffffu) << 40) | ((unsigned long long)(stage & 3u) << 38) |
        ((unsigned long long)min(slot, 0x3fffu) << 24) | ((unsigned long long)(sample & 1u) << 23) |
        ((unsigned long long)min(mol, 0x1ffffu) << 6) | ((unsigned long long)(detail & 3u) << 4) |
        (unsigned long long)(code == GORDER_ERR_BOX_RANGE ? 8u : (code == GORDER_ERR_TRAJECTORY_FORMAT ? 9u : (code & 15u)));
    atomicMin(reinterpret_cast<unsigned long long *>(err), key);
}
// the library's own range error (a coordinate so far outside the box that the reference would spin): end of frame
__device__ __forceinline__ void raise_box_range(uint32_t *err, uint32_t frame) {
    raise_error(err, GORDER_ERR_BOX_RANGE, frame, kStageEnd);
}

// End of a batch (k_batch_end, stream-ordered behind the batch's kernels): the batch's key — and the key its frames' decoder left in a
// record of its own (the decoder of a slot of gorder_hip_run_trajectory runs on another stream, beside the kernels
// of the batch before) — becomes THE error of the run if no earlier batch had one.  Keys order errors inside a batch
// only (frame IN BATCH is their leading field); across batches the order of submission decides, which is the order
// of the trajectory: the first error as the reference's sequential walk meets it (common.rs:248).

// ---- check_box (common.rs:186-198) for one frame
__device__ __forceinline__ void check_box_frame(const float *__restrict__ box9, uint32_t f, uint32_t *err) {
    const float *b = box9 + 9 * (size_t)f;
    bool all_nan = true;
    for (int i = 0; i < 9; i++) all_nan = all_nan && (b[i] != b[i]);
    if (all_nan) { raise_error(err, GORDER_ERR_UNDEFINED_BOX, f, kStageBox); return; }
    if (b[1] != 0.0f || b[2] != 0.0f || b[3] != 0.0f || b[5] != 0.0f || b[6] != 0.0f || b[7] != 0.0f) {
        raise_error(err, GORDER_ERR_NOT_ORTHOGONAL_BOX, f, kStageBox);
        return;
    }
    if (b[0] == 0.0f && b[4] == 0.0f && b[8] == 0.0f) { raise_error(err, GORDER_ERR_ZERO_BOX, f, kStageBox); return; }
    if (!(b[0] > 0.0f) || !(b[4] > 0.0f) || !(b[8] > 0.0f)) raise_error(err, GORDER_ERR_BOX_RANGE, f, kStageBox);
}

// ---- the end of a batch in ONE launch (three 5-us launches were 3 % of a 10 000-frame step of the headline workload):
// check_box for every frame (box9 null: no periodic boundaries), the frames counted (total_frames, topology/mod.rs:141-144,
// lives in the last word of the accumulator block so that a multi-GPU all-reduce sums it with the order sums,
// topology/mod.rs:243), and the batch's error key committed by the workgroup that finishes last: it sees the box errors of
// all frames.
__global__ __launch_bounds__(256) void k_batch_end(const float *__restrict__ box9, uint32_t n_frames, uint32_t *err,
                                                   unsigned long long *frames_word, unsigned long long ordinal,
                                                   const unsigned long long *decoder_key) {
    __shared__ uint32_t l_last;
    if (box9)
        for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frames; f += gridDim.x * blockDim.x) check_box_frame(box9, f, err);
    // the workgroup that finishes last commits (a ticket behind the error record; it leaves it at zero for the next batch)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) l_last = atomicAdd(err + kErrWords, 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!l_last || threadIdx.x != 0) return;
    err[kErrWords] = 0u;
    atomicAdd(frames_word, (unsigned long long)n_frames);
    unsigned long long *e = reinterpret_cast<unsigned long long *>(err);
    unsigned long long cur = atomicAdd(&e[0], 0ull);           // (an atomic read: the raises are atomics in L2)
    if (decoder_key && *decoder_key < cur) cur = *decoder_key;
    if (e[1] == kErrNone && cur != kErrNone) {
        e[1] = cur;
        e[2] = ordinal;
    }
    e[0] = kErrNone;
}

// A batch that failed on the host after some of its kernels were queued: what they raised is dropped with the batch
// (the host reports its own status), so that the next batch's k_batch_end does not commit it under ITS ordinal.
__global__ void k_batch_abort(uint32_t *err) {
    reinterpret_cast<unsigned long long *>(err)[0] = kErrNone;
    err[kErrWords] = 0u;
}

// ---- check_box, one thread per frame (the frame that primes a leaflet assignment) --------------
__global__ void k_check_box(const float *__restrict__ box9, uint32_t n_frames, uint32_t *err) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames) check_box_frame(box9, f, err);
}

// The box edges and their reciprocals per frame (GORDER_FLAG_UA_FAST_NORMALISE), eight floats a frame so that one scalar
// load fetches them: inv[f] = (Lx, Ly, Lz, -, 1 / Lx, 1 / Ly, 1 / Lz, -), the reciprocals IEEE divisions
__global__ void k_inv_box(const float *__restrict__ box9, uint32_t n_frames, float *__restrict__ inv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4u * n_frames) return;
    const uint32_t f = i >> 2, d = i & 3u;
    const float L = d < 3u ? box9[9u * (size_t)f + 4u * d] : 1.0f;
    inv[8u * (size_t)f + d] = L;
    inv[8u * (size_t)f + 4u + d] = 1.0f / L;
}

// acc[i] += sum_r rep[r][i]; rep := 0   (i < 4 * n_acc)
__global__ void k_fold_replicas(unsigned long long *acc, unsigned long long *rep, uint32_t n_rep, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long s = 0;
    for (uint32_t r = 0; r < n_rep; r++) {
        s += rep[(size_t)r * n + i];
        rep[(size_t)r * n + i] = 0;
    }
    acc[i] += s;
}

// ---- ordermap words --------------------------------------------------------------------------
// The scatter kernels add (1 << 42) + tick into one 64-bit word per (plane, slot, tile): the low 42 bits
// hold the signed tick sum, the bits above the sample count.  |tick| <= 1e6, so the sum of c samples stays
// inside 42 signed bits while c < 2^21; the host folds the words into the i64 sum / u64 count maps before
// any tile can have received that many samples (gorder_hip_handle::map_pending).
constexpr unsigned long long kMapOne = 1ull << 42;
constexpr unsigned long long kMapFoldLimit = 1ull << 21;
__device__ __forceinline__ void map_unpack(unsigned long long w, long long &sum, unsigned long long &cnt) {
    sum = (long long)(w << 22) >> 22;                      // sign-extend the low 42 bits
    cnt = (w - (unsigned long long)sum) >> 42;
}
constexpr unsigned long long kMapNoSample = ~0ull;   // staged entry of a lane without a sample in the map
// packed [planes][n] -> sums/cnts [3][n] (total, upper, lower); planes = 2 with leaflets (upper, lower), else 1
__global__ void k_fold_maps(unsigned long long *__restrict__ packed, unsigned long long *__restrict__ sums,
                            unsigned long long *__restrict__ cnts, size_t n, int leaflets) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long w0 = packed[i], w1 = leaflets ? packed[n + i] : 0ull;
        if (!(w0 | w1)) continue;
        long long s0, s1 = 0;
        unsigned long long c0, c1 = 0;
        map_unpack(w0, s0, c0);
        if (leaflets) map_unpack(w1, s1, c1);
        sums[i] += (unsigned long long)(s0 + s1);
        cnts[i] += c0 + c1;
        if (leaflets) {
            if (w0) { sums[n + i] += (unsigned long long)s0; cnts[n + i] += c0; packed[i] = 0; }
            if (w1) { sums[2 * n + i] += (unsigned long long)s1; cnts[2 * n + i] += c1; packed[n + i] = 0; }
        } else {
            packed[i] = 0;
        }
    }
}

}  // namespace
