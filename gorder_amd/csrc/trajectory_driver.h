// trajectory_driver.h — reader -> GPU feed: gorder_hip_run_trajectory and the double-buffered gorder_hip_submit_host.
// Part of the single translation unit gorder_hip.hip (included there after the handle's definition).
//
// What it replaces: the reference's driver `read_trajectory` (common.rs:239-342), a threaded map/reduce in which
// every analysis thread decodes its own frames (groan_rs traj_iter_map_reduce).  Here decoding and analysis are two
// stages of one pipeline:
//   reader thread  : gorder_xtc_read_window_mt (N decoder threads) fills the next PINNED host slot
//   calling thread : hipMemcpyAsync of the slot on a copy stream, then the kernels of that batch on the handle's
//                    stream (which waits for the copy through an event); the copy of batch k+1 and the decoding of
//                    batch k+2 overlap the kernels of batch k.
// A slot's host buffer is refilled once its copy has completed, its device buffer is overwritten once the kernels
// that read it have completed (the copy stream waits for that event) — no stream-wide synchronisation anywhere.
// With gorder_trajectory_t::device_decode the host threads only COPY the compressed blocks (gorder_xtc_pack_window),
// the slot's own stream carries blob + frame table + boxes to the device and runs the decoder (k_xtc_scan + k_xtc_chunks)
// into the slot's coordinate buffer; the handle's stream waits for that kernel instead of for a copy.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "../../include/gorder_xtc.h"

namespace {

struct TrajSlot {
    float *h_xyz = nullptr, *h_box = nullptr, *h_time = nullptr;   // pinned
    float *d_xyz = nullptr, *d_box = nullptr;
    // device decode: the compressed blocks and their table, pinned and on the device, and the slot's own stream
    uint8_t *h_blob = nullptr, *d_blob = nullptr;
    gorder_xtc_frame_t *h_frames = nullptr, *d_frames = nullptr;
    float *d_box_in = nullptr;      // the boxes land here first: d_box may still be read by the kernels of the slot's last batch
    // what the decoder reports back (2 words: short frames, largest need in 2^-16 of the bytes given; then the list of
    // the short frames), on the device and pinned; where every frame of the batch lies in which file
    uint32_t *d_stat = nullptr, *d_short = nullptr, *h_stat = nullptr, *h_short = nullptr;
    XtcCheckpoint *d_cp = nullptr;   // where the chunks of the batch's frames start (between the decoder's two kernels)
    std::vector<int64_t> file_pos;
    std::vector<uint32_t> file_idx;
    uint32_t prefix_q16 = 65536;    // the part of every block this batch was packed with
    size_t moved = 0;               // bytes copied to the device for this batch
    uint64_t blob_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t staged = nullptr, copied = nullptr, computed = nullptr;   // host buffers read; device buffers ready; kernels done
    bool copy_issued = false, compute_issued = false;
    uint32_t n = 0;
    std::vector<uint64_t> fidx;
};

// The staging buffers of a handle's trajectory runs.  They are kept by the handle between calls (and freed with it):
// pinning gigabytes of host memory costs ~0.1 s, which a second trajectory analysed with the same handle need not pay.
struct TrajCache {
    static constexpr int kSlots = 4;     // the host-decode route uses three of them
    TrajSlot slot[kSlots];
    bool dev = false;
    uint32_t batch = 0, n_stop = 0;
    size_t blob_cap = 0, xyz_bytes = 0;
    hipStream_t copy_stream = nullptr;
    unsigned long long *h_err = nullptr;     // pinned mirror of the device error key
    float *h_fix = nullptr;                  // pinned: one frame + its box, for the frames the host decodes after all
};

struct TrajPipe {
    static constexpr int kSlots = TrajCache::kSlots;
    TrajSlot *slot = nullptr;            // the cache's
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_q, filled_q;
    bool reader_done = false, stop = false;
    int reader_status = GORDER_XTC_OK;
    std::string reader_msg;
    double decode_s = 0.0, reader_stalled_s = 0.0, setup_s = 0.0;
    std::atomic<uint32_t> prefix_q16{65536};   // the leading part of every block the reader copies (device route)
    uint32_t reports = 0;                      // batches whose decoder report the submitter has seen (under mu)
    bool alloc_failed = false;
};

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void traj_cache_free(gorder_hip_handle *h) {
    TrajCache *c = static_cast<TrajCache *>(h->traj_cache);
    if (!c) return;
    for (TrajSlot &s : c->slot) {
        if (s.h_xyz) (void)hipHostFree(s.h_xyz);
        if (s.h_box) (void)hipHostFree(s.h_box);
        if (s.h_time) (void)hipHostFree(s.h_time);
        if (s.h_blob) (void)hipHostFree(s.h_blob);
        if (s.h_frames) (void)hipHostFree(s.h_frames);
        (void)hipFree(s.d_xyz);
        (void)hipFree(s.d_box);
        (void)hipFree(s.d_blob);
        (void)hipFree(s.d_frames);
        (void)hipFree(s.d_box_in);
        (void)hipFree(s.d_stat);
        (void)hipFree(s.d_short);
        (void)hipFree(s.d_cp);
        if (s.h_stat) (void)hipHostFree(s.h_stat);
        if (s.h_short) (void)hipHostFree(s.h_short);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.staged) (void)hipEventDestroy(s.staged);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.computed) (void)hipEventDestroy(s.computed);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->h_fix) (void)hipHostFree(c->h_fix);
    delete c;
    h->traj_cache = nullptr;
}

}  // namespace

extern "C" void gorder_hip_release_staging(gorder_hip_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    traj_cache_free(h);
}

extern "C" int gorder_hip_run_trajectory(gorder_hip_handle *h, const gorder_trajectory_t *tr,
                                         gorder_trajectory_stats_t *stats) {
    if (!h || !tr || !tr->paths || tr->n_paths == 0 || tr->step == 0) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t n_atoms = h->plan.n_atoms;
    // (0: the machine's threads, at most 16 — the copies and the decoder saturate there, and a node runs one rank per GPU)
    uint32_t n_threads = tr->n_threads ? tr->n_threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    // device decode: every file must be XTC (TRR / GRO have nothing to decompress); the atoms per frame of the
    // first file size the blob
    bool dev = tr->device_decode != 0;
    uint32_t n_file_atoms = 0, n_stop = 0, first_frame_bytes = 0;
    uint64_t total_file_bytes = 0;
    std::vector<int32_t> slot_of;
    for (uint32_t f = 0; dev && f < tr->n_paths; f++) {
        uint32_t na = 0, first = 0;
        uint64_t fbytes = 0;
        if (gorder_xtc_probe(tr->paths[f], &na, &fbytes, &first) != 1) { dev = false; break; }     // (an unreadable file: the reader thread reports it)
        total_file_bytes += fbytes;
        if (f == 0) first_frame_bytes = first;
        if (f == 0) n_file_atoms = na;
        else if (na != n_file_atoms) dev = false;
    }
    if (dev) {
        n_stop = n_file_atoms;
        if (tr->group && tr->n_group) {
            n_stop = 0;
            for (uint32_t k = 0; k < tr->n_group; k++) {
                if (tr->group[k] >= n_file_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "group index beyond the atoms of the file");
                n_stop = std::max(n_stop, tr->group[k] + 1u);
            }
        }
    }
    if (dev && tr->group && tr->n_group) {
        slot_of.assign(n_file_atoms, -1);
        for (uint32_t k = 0; k < tr->n_group && dev; k++) {                                 // (gorder_xtc_open checked the range)
            // an atom listed twice has two output slots but one entry here: the device decoder would leave the first
            // slot unwritten.  The host decoder fills every slot (gorder_xtc_next walks the group), so such a group
            // takes the host route for the whole run.
            if (slot_of[tr->group[k]] >= 0) dev = false;
            slot_of[tr->group[k]] = (int32_t)k;
        }
        if (!dev) slot_of.clear();
    }
    // contiguous frame shards: count what the window selects, headers only, and take this rank's share
    uint64_t shard_lo = 0, shard_n = UINT64_MAX, shard_total = 0;
    if (tr->shard_count > 1) {
        if (tr->shard_index >= tr->shard_count) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "shard_index >= shard_count");
        uint64_t state = 0;
        double last_time = -INFINITY;
        for (uint32_t f = 0; f < tr->n_paths; f++) {
            gorder_xtc_reader *r = nullptr;
            int st = gorder_xtc_open(tr->paths[f], nullptr, 0, &r);
            int64_t n = st;
            if (st == GORDER_XTC_OK) n = gorder_xtc_skip_window(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time, UINT64_MAX);
            if (r) gorder_xtc_close(r);
            if (n < 0)
                return fail(h, n == GORDER_XTC_ERR_FORMAT ? GORDER_ERR_TRAJECTORY_FORMAT : GORDER_ERR_INVALID_ARGUMENT,
                            std::string("cannot read ") + tr->paths[f] + " (reader status " + std::to_string(n) + ")");
            shard_total += (uint64_t)n;
        }
        shard_lo = shard_total * tr->shard_index / tr->shard_count;
        shard_n = shard_total * (tr->shard_index + 1ull) / tr->shard_count - shard_lo;
    }
    // A shard that begins between two assignment frames (leaflet frequency Every(n) / Once) depends on ONE frame of an
    // earlier rank's share: fetch it (host decoder, one frame) and prime the handle with it, so that sharded ranks need
    // nothing from the host beyond (i, n) — the cross-thread wait of leaflets.rs:1529-1565 becomes one extra frame read.
    {
        const gorder_leaflets_t &lf = h->tables.leaflets;
        const uint64_t f_first = tr->first_frame_index + shard_lo * tr->step;
        const uint64_t f_assign = lf.frequency == 0 ? tr->first_frame_index
                                                    : f_first / lf.frequency * lf.frequency;
        if (tr->shard_count > 1 && shard_n > 0 && lf.method != GORDER_LEAFLETS_NONE && lf.method != GORDER_LEAFLETS_MANUAL &&
            f_assign != f_first && f_assign >= tr->first_frame_index && (f_assign - tr->first_frame_index) % tr->step == 0) {
            uint64_t to_skip = (f_assign - tr->first_frame_index) / tr->step, state = 0;
            double last_time = -INFINITY;
            std::vector<float> x((size_t)n_atoms * 3u), bx(9);
            float t_ps = 0.0f;
            int64_t got = 0;
            for (uint32_t f = 0; f < tr->n_paths && got == 0; f++) {
                gorder_xtc_reader *r = nullptr;
                if (gorder_xtc_open(tr->paths[f], tr->group, tr->n_group, &r) != GORDER_XTC_OK) break;
                const int64_t sk = gorder_xtc_skip_window(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time, to_skip);
                if (sk >= 0) to_skip -= (uint64_t)sk;
                if (sk >= 0 && to_skip == 0 && gorder_xtc_n_atoms_out(r) == n_atoms)
                    got = gorder_xtc_read_window(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time, x.data(), bx.data(), &t_ps, 1);
                gorder_xtc_close(r);
                if (sk < 0) break;
            }
            if (got != 1) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "cannot read the leaflet assignment frame that precedes this shard");
            float *d_x = nullptr, *d_b = nullptr;
            hipError_t e = hipMalloc((void **)&d_x, x.size() * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void **)&d_b, 9 * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(d_x, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_b, bx.data(), 9 * sizeof(float), hipMemcpyHostToDevice);
            int st = e == hipSuccess ? gorder_hip_prime_leaflets(h, d_x, h->tables.handle_pbc ? d_b : nullptr, f_assign)
                                     : fail(h, GORDER_ERR_DEVICE, std::string("priming frame: ") + hipGetErrorString(e));
            if (st == GORDER_OK) st = gorder_hip_synchronize(h);       // (the buffers go away below)
            (void)hipFree(d_x);
            (void)hipFree(d_b);
            if (st != GORDER_OK) return st;
        }
    }
    // frames per batch: ~128 MB of coordinates per slot unless the host asks otherwise (>= 16 so that the launches
    // amortise; a batch is also what one decoder pass spreads over its threads).  The device decoder works one frame
    // per lane: its batches are as large as 1 GiB of coordinates allows, up to 16384 frames, and there are four slots
    // (the decoding of two batches overlaps the packing and the copy of the next ones).
    uint32_t batch = tr->batch_frames;
    if (batch == 0 && !dev) batch = (uint32_t)std::min<size_t>(4096, std::max<size_t>(16, ((size_t)128 << 20) / ((size_t)n_atoms * 12u)));
    if (batch == 0 && dev) {
        // One lane decodes one frame, so a launch takes (atoms per frame) x 0.6 us whatever its size: below a few hundred
        // frames per launch the host's decoder threads are faster.  Large systems get a larger budget (4 GiB of
        // coordinates per slot; HBM has the room), and beyond that (about a million atoms per frame) the run uses
        // the host decoder.
        size_t b = std::min<size_t>(16384, ((size_t)1 << 30) / ((size_t)n_atoms * 12u));
        if (b < 512) b = std::min<size_t>(512, ((size_t)4 << 30) / ((size_t)n_atoms * 12u));
        if (b < 512) dev = false;
        else batch = (uint32_t)b;
        // A short trajectory does not need (and should not pay for pinning) four full-size slots: about a third of its
        // frames per batch, estimated from the files' sizes and the first frame's
        if (dev && first_frame_bytes) {
            const uint64_t est_frames = total_file_bytes / first_frame_bytes + 1u;
            batch = (uint32_t)std::min<uint64_t>(batch, std::max<uint64_t>(256u, (est_frames + 2u) / 3u + 16u));
        }
    }
    if (batch == 0) batch = (uint32_t)std::min<size_t>(4096, std::max<size_t>(16, ((size_t)128 << 20) / ((size_t)n_atoms * 12u)));
    const size_t xyz_bytes = (size_t)batch * n_atoms * 3u * sizeof(float), box_bytes = (size_t)batch * 9u * sizeof(float);
    // a compressed atom takes 3-5 bytes at the usual precision, never more than 10: 6 per atom and frame on average,
    // and room for one worst-case frame
    // (a file with much more in it than the analysed atoms — water — would ask for gigabytes: at most 1 GiB per slot,
    // a batch then simply ends when its blob is full)
    const size_t blob_cap = dev ? std::max<size_t>(std::min<size_t>((size_t)batch * n_file_atoms * 6u, (size_t)1 << 30),
                                                   (size_t)n_file_atoms * 12u + 4096u) + 4096u : 0;
    TrajCache *cache = static_cast<TrajCache *>(h->traj_cache);
    if (cache && !(cache->dev == dev && cache->batch == batch && cache->blob_cap == blob_cap && cache->xyz_bytes == xyz_bytes &&
                   cache->n_stop == n_stop)) {
        traj_cache_free(h);            // another shape of run: start over
        cache = nullptr;
    }
    if (!cache) {
        cache = new (std::nothrow) TrajCache();
        if (!cache) return fail(h, GORDER_ERR_DEVICE, "out of host memory");
        cache->dev = dev; cache->batch = batch; cache->blob_cap = blob_cap; cache->xyz_bytes = xyz_bytes; cache->n_stop = n_stop;
        h->traj_cache = cache;
        h->traj_cache_free = &traj_cache_free;
    }
    TrajPipe pipe;
    pipe.slot = cache->slot;
    for (int k = 0; k < TrajPipe::kSlots; k++) { pipe.slot[k].copy_issued = pipe.slot[k].compute_issued = false; }   // (the last run ended synchronised)
    int32_t *d_slot_of = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_slot_of); };
#define TRAJ_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return fail(h, GORDER_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
        }                                                                                           \
    } while (0)
    if (!cache->copy_stream) TRAJ_TRY(hipStreamCreateWithFlags(&cache->copy_stream, hipStreamNonBlocking));
    if (!cache->h_err) TRAJ_TRY(hipHostMalloc((void **)&cache->h_err, sizeof(unsigned long long), hipHostMallocDefault));
    if (dev && !cache->h_fix) TRAJ_TRY(hipHostMalloc((void **)&cache->h_fix, ((size_t)n_atoms * 3u + 9u) * sizeof(float), hipHostMallocDefault));
    const hipStream_t copy_stream = cache->copy_stream;
    unsigned long long *h_err = cache->h_err;   // lets the loop stop at the first error
    *h_err = kErrNone;
    if (dev && !slot_of.empty()) {
        TRAJ_TRY(hipMalloc((void **)&d_slot_of, slot_of.size() * sizeof(int32_t)));
        TRAJ_TRY(hipMemcpy(d_slot_of, slot_of.data(), slot_of.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // Pinning a gigabyte takes tens of milliseconds: only the first slot's share of that stands before the first frame
    // is read, the other slots are made ready by a thread of their own.
    auto alloc_slot = [&](TrajSlot &s) -> hipError_t {
        if (s.computed) return hipSuccess;           // kept from an earlier call
        hipError_t e = hipSuccess;
        auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
        if (dev) {
            ok(hipHostMalloc((void **)&s.h_blob, blob_cap, hipHostMallocDefault));   // (write-combined: no faster, measured)
            ok(hipHostMalloc((void **)&s.h_frames, (size_t)batch * sizeof(gorder_xtc_frame_t), hipHostMallocDefault));
            ok(hipMalloc((void **)&s.d_blob, blob_cap));
            ok(hipMalloc((void **)&s.d_frames, (size_t)batch * sizeof(gorder_xtc_frame_t)));
            ok(hipMalloc((void **)&s.d_box_in, box_bytes));
            ok(hipMalloc((void **)&s.d_stat, 4 * sizeof(uint32_t)));      // [0..1] the decoder's report, [2..3] its error key
            ok(hipMalloc((void **)&s.d_short, (size_t)batch * sizeof(uint32_t)));
            ok(hipMalloc((void **)&s.d_cp, xtc_checkpoints(batch, n_stop) * sizeof(XtcCheckpoint)));
            ok(hipHostMalloc((void **)&s.h_stat, 2 * sizeof(uint32_t), hipHostMallocDefault));
            ok(hipHostMalloc((void **)&s.h_short, (size_t)batch * sizeof(uint32_t), hipHostMallocDefault));
            s.file_pos.resize(batch);
            s.file_idx.resize(batch);
            ok(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        } else {
            ok(hipHostMalloc((void **)&s.h_xyz, xyz_bytes, hipHostMallocDefault));
        }
        ok(hipHostMalloc((void **)&s.h_box, box_bytes, hipHostMallocDefault));
        ok(hipHostMalloc((void **)&s.h_time, (size_t)batch * sizeof(float), hipHostMallocDefault));
        ok(hipMalloc((void **)&s.d_xyz, xyz_bytes));
        ok(hipMalloc((void **)&s.d_box, box_bytes));
        ok(hipEventCreateWithFlags(&s.staged, hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
        return e;
    };
    const int n_slots = dev ? TrajPipe::kSlots : 3;
    // slot 0 here, the others by a thread of their own while the reader already fills slot 0
    {
        const hipError_t e0 = alloc_slot(pipe.slot[0]);
        if (e0 != hipSuccess) {
            traj_cache_free(h);        // (a slot that is only partly there)
            TRAJ_TRY(e0);
        }
    }
    pipe.free_q.push_back(0);
    std::thread allocator([&, n_slots]() {
        (void)hipSetDevice(h->device);
        for (int k = 1; k < n_slots; k++) {
            {
                std::lock_guard<std::mutex> lk(pipe.mu);
                if (pipe.stop) return;
            }
            const auto t0 = std::chrono::steady_clock::now();
            const hipError_t e = alloc_slot(pipe.slot[k]);
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.setup_s += seconds_since(t0);
            if (e != hipSuccess) { pipe.alloc_failed = true; return; }   // the run goes on with the slots there are
            pipe.free_q.push_back(k);
            pipe.cv.notify_all();
        }
    });
    const double setup_s = seconds_since(t_start);

    // the device route's block copies go to a pool that outlives a window: the header scan of the next file or window
    // runs while the blocks of this one are still being copied
    gorder_xtc_pool *pool = nullptr;
    if (dev && gorder_xtc_pool_create(n_threads, &pool) != GORDER_XTC_OK) pool = nullptr;
    const char *forced_prefix = getenv("GORDER_HIP_PREFIX_Q16");          // test switch: a fixed part, whatever it leads to
    const bool adapt_prefix = dev && !forced_prefix && !env_flag("GORDER_HIP_NO_PREFIX") && n_stop < n_file_atoms;
    if (dev && forced_prefix) pipe.prefix_q16.store((uint32_t)std::max(1l, std::min(65536l, atol(forced_prefix))));
    // only a run that copies parts of blocks has anything to look at between the two stages; any other queues stage B
    // right behind stage A and never waits for the device
    const bool verify = dev && (adapt_prefix || forced_prefix);
    // ---- reader thread: the sequential part of read_trajectory (time window, step, concatenation) + decoding
    const int device = h->device;
    std::thread reader([&, device]() {
        (void)hipSetDevice(device);
        uint64_t state = 0, analysed = shard_lo;   // (frames are numbered as in the whole trajectory)
        double last_time = -INFINITY;
        uint32_t f = 0;                       // next file to open
        gorder_xtc_reader *r = nullptr;       // the open one
        bool done = shard_n == 0;
        uint32_t n_filled = 0;                // batches this thread has started to fill
        uint64_t to_skip = shard_lo, left = shard_n;      // frames before this rank's shard; frames of it still to read
        auto give_up = [&](int st, const std::string &msg) {
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.reader_status = st;
            pipe.reader_msg = msg;
            done = true;
        };
        while (!done) {
            int k = -1;
            {
                const auto t0 = std::chrono::steady_clock::now();
                std::unique_lock<std::mutex> lk(pipe.mu);
                pipe.cv.wait(lk, [&] { return pipe.stop || !pipe.free_q.empty(); });
                pipe.reader_stalled_s += seconds_since(t0);
                if (pipe.stop) break;
                k = pipe.free_q.front();
                pipe.free_q.pop_front();
            }
            TrajSlot &s = pipe.slot[k];
            const auto t_wait = std::chrono::steady_clock::now();
            if (s.copy_issued) (void)hipEventSynchronize(s.staged);   // the previous batch has left the host buffers
            const auto t1 = std::chrono::steady_clock::now();
            {
                std::lock_guard<std::mutex> lk(pipe.mu);
                pipe.reader_stalled_s += std::chrono::duration<double>(t1 - t_wait).count();
            }
            // a batch is filled across file boundaries: a trajectory split into many short files still makes full batches
            s.n = 0;
            s.blob_bytes = 0;
            // A run that will copy only parts of blocks learns the part from the decoder's first report: its first
            // two batches are short (256 and 1024 frames) and the third waits for that report, so that no more than
            // ~1 300 frames travel whole.
            uint32_t fill = batch;
            if (adapt_prefix && tr->batch_frames == 0) {
                if (n_filled == 0) fill = std::min(batch, 256u);
                else if (n_filled == 1) fill = std::min(batch, 1024u);
                else if (n_filled == 2) {
                    std::unique_lock<std::mutex> lk(pipe.mu);
                    pipe.cv.wait(lk, [&] { return pipe.stop || pipe.reports > 0; });
                }
            }
            n_filled++;
            s.prefix_q16 = pipe.prefix_q16.load();
            while (!done && s.n < fill) {
                if (left == 0) { done = true; break; }
                if (!r) {
                    if (f == tr->n_paths) { done = true; break; }
                    int st = gorder_xtc_open(tr->paths[f], tr->group, tr->n_group, &r);
                    if (st == GORDER_XTC_OK && gorder_xtc_n_atoms_out(r) != n_atoms) st = GORDER_XTC_ERR_ARGUMENT;
                    if (st != GORDER_XTC_OK) {
                        give_up(st, std::string("cannot read ") + tr->paths[f] +
                                        (st == GORDER_XTC_ERR_ARGUMENT ? ": atoms per frame differ from the tables" : ""));
                        break;
                    }
                }
                if (to_skip) {                           // pass over the frames of the ranks before this one
                    const int64_t sk = gorder_xtc_skip_window(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time, to_skip);
                    if (sk < 0) { give_up((int)sk, std::string("read error in ") + tr->paths[f]); break; }
                    to_skip -= (uint64_t)sk;
                    if (to_skip) {                       // this file ended inside the part to pass over
                        gorder_xtc_close(r);
                        r = nullptr;
                        f++;
                        continue;
                    }
                }
                const uint64_t want = std::min<uint64_t>(fill - s.n, left);
                uint64_t used = 0;
                const int64_t got =
                    dev ? gorder_xtc_pack_window_ex(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time,
                                                    s.h_blob + s.blob_bytes, blob_cap - s.blob_bytes, &used, s.h_frames + s.n,
                                                    s.h_box + 9u * (size_t)s.n, s.h_time + s.n, want, n_threads, pool,
                                                    s.prefix_q16, s.file_pos.data() + s.n)
                        : gorder_xtc_read_window_mt(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time,
                                                    s.h_xyz + (size_t)s.n * n_atoms * 3u, s.h_box + 9u * (size_t)s.n,
                                                    s.h_time + s.n, want, n_threads);
                if (dev && got == GORDER_XTC_ERR_NO_SPACE) {
                    if (s.n > 0) break;                                           // the blob is full: this batch is complete
                    give_up(GORDER_XTC_ERR_ARGUMENT, std::string("a single frame of ") + tr->paths[f] + " does not fit the staging blob");
                    break;
                }
                if (got < 0) {
                    give_up((int)got, std::string("read error in ") + tr->paths[f]);
                    break;
                }
                if (got == 0) {                                                   // end of this file
                    gorder_xtc_close(r);
                    r = nullptr;
                    f++;
                    continue;
                }
                if (dev) {
                    for (int64_t q = 0; q < got; q++) {
                        s.h_frames[s.n + (size_t)q].offset += s.blob_bytes;
                        s.file_idx[s.n + (size_t)q] = f;
                    }
                    s.blob_bytes += used;
                }
                s.n += (uint32_t)got;
                left -= (uint64_t)got;
            }
            if (pool) {                                  // the slot's blocks are complete only now
                const int cst = gorder_xtc_pool_wait(pool);
                if (cst != GORDER_XTC_OK && !done) give_up(cst, "read error while copying the compressed blocks");
                else if (cst != GORDER_XTC_OK) { std::lock_guard<std::mutex> lk(pipe.mu); if (pipe.reader_status == GORDER_XTC_OK) { pipe.reader_status = cst; pipe.reader_msg = "read error while copying the compressed blocks"; } }
            }
            const double dt = seconds_since(t1);
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.decode_s += dt;
            if (s.n == 0 || pipe.reader_status != GORDER_XTC_OK) {
                pipe.free_q.push_front(k);
                break;
            }
            s.fidx.resize((size_t)s.n);
            // SystemTopology::frame of the k-th analysed frame = k * step (topology/mod.rs:141-144)
            for (uint32_t q = 0; q < s.n; q++) s.fidx[q] = tr->first_frame_index + (analysed + q) * tr->step;
            analysed += s.n;
            pipe.filled_q.push_back(k);
            pipe.cv.notify_all();
        }
        if (r) gorder_xtc_close(r);
        std::lock_guard<std::mutex> lk(pipe.mu);
        pipe.reader_done = true;
        pipe.cv.notify_all();
    });

    // ---- submitter (this thread)
    // Two stages per batch.  A: as soon as a batch is filled, its copies (and, on the device route, its decode kernel
    // and the decoder's report) are queued on the slot's stream.  B: once that has finished — the device route looks
    // at the report first: frames of which too short a part was copied are decoded by the host after all, and the part
    // copied of later batches follows what the decoder needed — the analysis is queued on the handle's stream.
    // Batches are analysed in the order they were read.
    int status = GORDER_OK;
    uint64_t frames = 0, batches = 0, bytes = 0, frames_fixed = 0;
    double starved_s = 0.0;
    std::string hip_msg;
    uint32_t need_q16 = 0;                                                   // the largest part of a block a frame needed so far
    auto stage_a = [&](int k) {
        TrajSlot &s = pipe.slot[k];
        const size_t nx = (size_t)s.n * n_atoms * 3u * sizeof(float), nb = (size_t)s.n * 9u * sizeof(float);
        hipError_t e = hipSuccess;
        const hipStream_t feed = dev ? s.stream : copy_stream;
        // the slot's coordinate buffer is free once the kernels of its last batch are done; the compressed blocks go
        // to a buffer of their own and need not wait for that
        if (s.compute_issued && !dev) e = hipStreamWaitEvent(feed, s.computed, 0);
        s.moved = nb;
        if (dev) {
            const size_t nf = (size_t)s.n * sizeof(gorder_xtc_frame_t);
            s.moved += (size_t)s.blob_bytes + nf;
            if (e == hipSuccess) e = hipMemsetAsync(s.d_stat, 0, 2 * sizeof(uint32_t), feed);
            if (e == hipSuccess) e = hipMemsetAsync(s.d_stat + 2, 0xff, 2 * sizeof(uint32_t), feed);      // kErrNone
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_blob, s.h_blob, (size_t)s.blob_bytes, hipMemcpyHostToDevice, feed);
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_frames, s.h_frames, nf, hipMemcpyHostToDevice, feed);
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_box_in, s.h_box, nb, hipMemcpyHostToDevice, feed);
            if (e == hipSuccess) e = hipEventRecord(s.staged, feed);
            if (e == hipSuccess && s.compute_issued) e = hipStreamWaitEvent(feed, s.computed, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_box, s.d_box_in, nb, hipMemcpyDeviceToDevice, feed);
            if (e == hipSuccess && status == GORDER_OK)
                status = xtc_decode_on(h, feed, s.d_blob, s.blob_bytes, s.d_frames, s.n, n_file_atoms, d_slot_of, n_stop,
                                       s.d_xyz, n_atoms, s.d_stat, s.d_short, s.d_stat + 2, s.d_cp);
            if (e == hipSuccess) e = hipMemcpyAsync(s.h_stat, s.d_stat, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, feed);
            if (e == hipSuccess) e = hipMemcpyAsync(s.h_short, s.d_short, (size_t)s.n * sizeof(uint32_t), hipMemcpyDeviceToHost, feed);
        } else {
            s.moved += nx;
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_xyz, s.h_xyz, nx, hipMemcpyHostToDevice, feed);
            if (e == hipSuccess) e = hipMemcpyAsync(s.d_box, s.h_box, nb, hipMemcpyHostToDevice, feed);
            if (e == hipSuccess) e = hipEventRecord(s.staged, feed);
        }
        if (e == hipSuccess) e = hipEventRecord(s.copied, feed);
        if (e == hipSuccess) s.copy_issued = true;
        if (e != hipSuccess) { status = GORDER_ERR_DEVICE; hip_msg = std::string("trajectory copy: ") + hipGetErrorString(e); }
    };
    // the frames of a batch the decoder could not finish with the part it was given: decoded here, one by one
    auto fix_short_frames = [&](TrajSlot &s, uint32_t n_short) {
        gorder_xtc_reader *r = nullptr;
        uint32_t open_idx = UINT32_MAX;
        float *fx = cache->h_fix, *fb = cache->h_fix + (size_t)n_atoms * 3u;
        for (uint32_t q = 0; q < n_short && status == GORDER_OK; q++) {
            const uint32_t fr = s.h_short[q];
            if (fr >= s.n) { status = fail(h, GORDER_ERR_DEVICE, "decoder report out of range"); break; }
            if (s.file_idx[fr] != open_idx) {
                if (r) gorder_xtc_close(r);
                r = nullptr;
                open_idx = s.file_idx[fr];
                if (gorder_xtc_open(tr->paths[open_idx], tr->group, tr->n_group, &r) != GORDER_XTC_OK) r = nullptr;
            }
            const int st = r ? gorder_xtc_read_at(r, s.file_pos[fr], fx, fb) : GORDER_XTC_ERR_OPEN;
            if (st != GORDER_XTC_OK) {
                status = fail(h, st == GORDER_XTC_ERR_FORMAT ? GORDER_ERR_TRAJECTORY_FORMAT : GORDER_ERR_INVALID_ARGUMENT,
                              std::string("cannot decode a frame of ") + tr->paths[open_idx] + " (reader status " + std::to_string(st) + ")");
                break;
            }
            const hipError_t e = hipMemcpy(s.d_xyz + (size_t)fr * n_atoms * 3u, fx, (size_t)n_atoms * 3u * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) status = fail(h, GORDER_ERR_DEVICE, std::string("trajectory copy: ") + hipGetErrorString(e));
        }
        if (r) gorder_xtc_close(r);
        frames_fixed += n_short;
    };
    auto stage_b = [&](int k) {
        TrajSlot &s = pipe.slot[k];
        hipError_t e = hipSuccess;
        if (verify && status == GORDER_OK) {
            e = hipEventSynchronize(s.copied);
            if (e == hipSuccess) {
                const uint32_t n_short = s.h_stat[0], q = s.h_stat[1];
                if (n_short) fix_short_frames(s, std::min(n_short, s.n));
                if (adapt_prefix) {
                    // what this batch needed, as a part of the whole block; the next batches get an eighth and 1 % more
                    // (nothing less than nine tenths is worth the trouble), and more at once when frames came out short
                    need_q16 = std::max<uint32_t>(need_q16, (uint32_t)std::min<uint64_t>(65536u, ((uint64_t)q * s.prefix_q16) >> 16));
                    uint32_t next = need_q16 + need_q16 / 8u + 656u;
                    if (n_short) next = std::max<uint32_t>(next, s.prefix_q16 + s.prefix_q16 / 4u);
                    pipe.prefix_q16.store(next >= 58982u ? 65536u : next);
                }
                std::lock_guard<std::mutex> lk(pipe.mu);
                pipe.reports++;
                pipe.cv.notify_all();
            }
        }
        if (e == hipSuccess && status == GORDER_OK) e = hipStreamWaitEvent(h->stream, s.copied, 0);
        if (e != hipSuccess && status == GORDER_OK) { status = GORDER_ERR_DEVICE; hip_msg = std::string("trajectory copy: ") + hipGetErrorString(e); }
        if (status == GORDER_OK) {
            // a frame the decoder could not make sense of is an error of THIS batch, ordered with the errors its analysis
            // raises (the frame's number leads the key) and behind those of the batches before (k_batch_end)
            h->decoder_key = dev ? reinterpret_cast<const unsigned long long *>(s.d_stat + 2) : nullptr;
            status = gorder_hip_submit_device(h, s.d_xyz, h->tables.handle_pbc ? s.d_box : nullptr, s.fidx.data(), s.n);
            h->decoder_key = nullptr;
        }
        if (status == GORDER_OK) {
            e = hipEventRecord(s.computed, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h_err, reinterpret_cast<unsigned long long *>(h->d_err) + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
            if (e != hipSuccess) { status = GORDER_ERR_DEVICE; hip_msg = std::string("trajectory submit: ") + hipGetErrorString(e); }
            s.compute_issued = true;
            frames += s.n;
            batches++;
            bytes += s.moved;
        }
        const bool device_error = *reinterpret_cast<volatile unsigned long long *>(h_err) != kErrNone;
        std::lock_guard<std::mutex> lk(pipe.mu);
        pipe.free_q.push_back(k);
        if (status != GORDER_OK || device_error) pipe.stop = true;   // first error aborts the iteration (common.rs:248)
        pipe.cv.notify_all();
        return status == GORDER_OK && !device_error;
    };
    std::deque<int> decoding;            // batches past stage A whose decoder report is still to come, oldest first
    for (bool go = true; go;) {
        // stage A for every batch the reader has filled
        std::vector<int> fresh;
        bool finished = false;
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(pipe.mu);
            if (decoding.empty()) {      // nothing of ours in flight: wait for the reader
                pipe.cv.wait(lk, [&] { return !pipe.filled_q.empty() || pipe.reader_done; });
                starved_s += seconds_since(t0);
            }
            while (!pipe.filled_q.empty()) {
                fresh.push_back(pipe.filled_q.front());
                pipe.filled_q.pop_front();
            }
            finished = pipe.reader_done && fresh.empty() && decoding.empty();
        }
        if (finished) break;
        for (int k : fresh) {
            if (!go) break;
            stage_a(k);
            if (verify) decoding.push_back(k);
            else go = stage_b(k);
        }
        // stage B for the batches whose report has arrived; with nothing fresh and nothing ready, sleep a little
        // (the copies of the batches behind are queued already: nothing idles while this thread does)
        bool any = !fresh.empty();
        while (go && !decoding.empty() &&
               (status != GORDER_OK || hipEventQuery(pipe.slot[decoding.front()].copied) == hipSuccess)) {
            const int front = decoding.front();
            decoding.pop_front();
            go = stage_b(front);
            any = true;
        }
        if (go && !any && !decoding.empty()) {
            std::unique_lock<std::mutex> lk(pipe.mu);
            pipe.cv.wait_for(lk, std::chrono::microseconds(100), [&] { return !pipe.filled_q.empty(); });
        }
    }
    {
        std::lock_guard<std::mutex> lk(pipe.mu);
        pipe.stop = true;
        pipe.cv.notify_all();
    }
    reader.join();
    allocator.join();
    if (pool) gorder_xtc_pool_destroy(pool);
    (void)hipStreamSynchronize(copy_stream);
    for (int k = 0; k < TrajPipe::kSlots; k++)
        if (pipe.slot[k].stream) (void)hipStreamSynchronize(pipe.slot[k].stream);
    const int sync_status = gorder_hip_synchronize(h);      // surfaces device errors
    if (status == GORDER_OK) status = sync_status;
    else if (!hip_msg.empty()) h->err_msg = hip_msg;
    if (status == GORDER_OK && pipe.reader_status != GORDER_XTC_OK) {
        // a corrupt or truncated file is the same error whichever side decodes it
        status = pipe.reader_status == GORDER_XTC_ERR_FORMAT ? GORDER_ERR_TRAJECTORY_FORMAT : GORDER_ERR_INVALID_ARGUMENT;
        h->err_msg = pipe.reader_msg + " (reader status " + std::to_string(pipe.reader_status) + ")";
    }
    if (stats) {
        stats->n_frames = frames;
        stats->n_batches = batches;
        stats->bytes_h2d = bytes;
        stats->seconds_total = seconds_since(t_start);
        stats->seconds_decode = pipe.decode_s;
        stats->seconds_reader_stalled = pipe.reader_stalled_s;
        stats->seconds_gpu_starved = starved_s;
        stats->batch_frames = batch;
        stats->decoder_threads = n_threads;
        stats->device_decode = dev ? 1u : 0u;
        stats->frames_decoded_by_host = (uint32_t)std::min<uint64_t>(frames_fixed, UINT32_MAX);
        stats->seconds_setup = setup_s + pipe.setup_s;
        stats->shard_first = shard_lo;
        stats->shard_frames_total = tr->shard_count > 1 ? shard_total : frames;
    }
    cleanup();
    if (pipe.alloc_failed) traj_cache_free(h);      // (a slot that is only partly there)
#undef TRAJ_TRY
    return status;
}
