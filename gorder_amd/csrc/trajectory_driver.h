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
#pragma once

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
    hipEvent_t copied = nullptr, computed = nullptr;
    bool copy_issued = false, compute_issued = false;
    uint32_t n = 0;
    std::vector<uint64_t> fidx;
};

struct TrajPipe {
    static constexpr int kSlots = 3;
    TrajSlot slot[kSlots];
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> free_q, filled_q;
    bool reader_done = false, stop = false;
    int reader_status = GORDER_XTC_OK;
    std::string reader_msg;
    double decode_s = 0.0, reader_stalled_s = 0.0;
};

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void traj_free(gorder_hip_handle *h, TrajPipe &p) {
    for (TrajSlot &s : p.slot) {
        if (s.h_xyz) (void)hipHostFree(s.h_xyz);
        if (s.h_box) (void)hipHostFree(s.h_box);
        if (s.h_time) (void)hipHostFree(s.h_time);
        (void)hipFree(s.d_xyz);
        (void)hipFree(s.d_box);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.computed) (void)hipEventDestroy(s.computed);
    }
    (void)h;
}

}  // namespace

extern "C" int gorder_hip_run_trajectory(gorder_hip_handle *h, const gorder_trajectory_t *tr,
                                         gorder_trajectory_stats_t *stats) {
    if (!h || !tr || !tr->paths || tr->n_paths == 0 || tr->step == 0) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t n_atoms = h->plan.n_atoms;
    uint32_t n_threads = tr->n_threads ? tr->n_threads : std::max(1u, std::thread::hardware_concurrency());
    // frames per batch: ~128 MB of coordinates per slot unless the host asks otherwise (>= 16 so that the launches
    // amortise; a batch is also what one decoder pass spreads over its threads)
    uint32_t batch = tr->batch_frames;
    if (batch == 0) batch = (uint32_t)std::min<size_t>(4096, std::max<size_t>(16, ((size_t)128 << 20) / ((size_t)n_atoms * 12u)));
    const size_t xyz_bytes = (size_t)batch * n_atoms * 3u * sizeof(float), box_bytes = (size_t)batch * 9u * sizeof(float);

    TrajPipe pipe;
    hipStream_t copy_stream = nullptr;
    unsigned long long *h_err = nullptr;   // pinned mirror of the device error key: lets the loop stop at the first error
    auto cleanup = [&]() {
        traj_free(h, pipe);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (h_err) (void)hipHostFree(h_err);
    };
#define TRAJ_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cleanup();                                                                              \
            return fail(h, GORDER_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
        }                                                                                           \
    } while (0)
    TRAJ_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    TRAJ_TRY(hipHostMalloc((void **)&h_err, sizeof(unsigned long long), hipHostMallocDefault));
    *h_err = kErrNone;
    for (TrajSlot &s : pipe.slot) {
        TRAJ_TRY(hipHostMalloc((void **)&s.h_xyz, xyz_bytes, hipHostMallocDefault));
        TRAJ_TRY(hipHostMalloc((void **)&s.h_box, box_bytes, hipHostMallocDefault));
        TRAJ_TRY(hipHostMalloc((void **)&s.h_time, (size_t)batch * sizeof(float), hipHostMallocDefault));
        TRAJ_TRY(hipMalloc((void **)&s.d_xyz, xyz_bytes));
        TRAJ_TRY(hipMalloc((void **)&s.d_box, box_bytes));
        TRAJ_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        TRAJ_TRY(hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
    }
    for (int k = 0; k < TrajPipe::kSlots; k++) pipe.free_q.push_back(k);

    // ---- reader thread: the sequential part of read_trajectory (time window, step, concatenation) + decoding
    const int device = h->device;
    std::thread reader([&, device]() {
        (void)hipSetDevice(device);
        uint64_t state = 0, analysed = 0;
        double last_time = -INFINITY;
        for (uint32_t f = 0; f < tr->n_paths; f++) {
            gorder_xtc_reader *r = nullptr;
            int st = gorder_xtc_open(tr->paths[f], tr->group, tr->n_group, &r);
            if (st == GORDER_XTC_OK && gorder_xtc_n_atoms_out(r) != n_atoms) st = GORDER_XTC_ERR_ARGUMENT;
            if (st != GORDER_XTC_OK) {
                std::lock_guard<std::mutex> lk(pipe.mu);
                pipe.reader_status = st;
                pipe.reader_msg = std::string("cannot read ") + tr->paths[f] +
                                  (st == GORDER_XTC_ERR_ARGUMENT ? ": atoms per frame differ from the tables" : "");
                if (r) gorder_xtc_close(r);
                break;
            }
            for (;;) {
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
                if (s.copy_issued) (void)hipEventSynchronize(s.copied);   // the previous batch has left the host buffer
                const auto t0 = std::chrono::steady_clock::now();
                const int64_t got = gorder_xtc_read_window_mt(r, tr->begin_ps, tr->end_ps, tr->step, &state, &last_time,
                                                              s.h_xyz, s.h_box, s.h_time, batch, n_threads);
                const double dt = seconds_since(t0);
                std::lock_guard<std::mutex> lk(pipe.mu);
                pipe.decode_s += dt;
                if (got <= 0) {
                    pipe.free_q.push_front(k);
                    if (got < 0) {
                        pipe.reader_status = (int)got;
                        pipe.reader_msg = std::string("read error in ") + tr->paths[f];
                    }
                    break;
                }
                s.n = (uint32_t)got;
                s.fidx.resize((size_t)got);
                // SystemTopology::frame of the k-th analysed frame = k * step (topology/mod.rs:141-144)
                for (int64_t q = 0; q < got; q++) s.fidx[(size_t)q] = tr->first_frame_index + (analysed + (uint64_t)q) * tr->step;
                analysed += (uint64_t)got;
                pipe.filled_q.push_back(k);
                pipe.cv.notify_all();
            }
            gorder_xtc_close(r);
            std::lock_guard<std::mutex> lk(pipe.mu);
            if (pipe.stop || pipe.reader_status != GORDER_XTC_OK) break;
        }
        std::lock_guard<std::mutex> lk(pipe.mu);
        pipe.reader_done = true;
        pipe.cv.notify_all();
    });

    // ---- submitter (this thread)
    int status = GORDER_OK;
    uint64_t frames = 0, batches = 0, bytes = 0;
    double starved_s = 0.0;
    std::string hip_msg;
    for (;;) {
        int k = -1;
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(pipe.mu);
            pipe.cv.wait(lk, [&] { return !pipe.filled_q.empty() || pipe.reader_done; });
            starved_s += seconds_since(t0);
            if (pipe.filled_q.empty()) break;     // reader done and nothing left
            k = pipe.filled_q.front();
            pipe.filled_q.pop_front();
        }
        TrajSlot &s = pipe.slot[k];
        const size_t nx = (size_t)s.n * n_atoms * 3u * sizeof(float), nb = (size_t)s.n * 9u * sizeof(float);
        hipError_t e = hipSuccess;
        if (s.compute_issued) e = hipStreamWaitEvent(copy_stream, s.computed, 0);   // kernels of the slot's last batch
        if (e == hipSuccess) e = hipMemcpyAsync(s.d_xyz, s.h_xyz, nx, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.d_box, s.h_box, nb, hipMemcpyHostToDevice, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(s.copied, copy_stream);
        if (e == hipSuccess) { s.copy_issued = true; e = hipStreamWaitEvent(h->stream, s.copied, 0); }
        if (e != hipSuccess) { status = GORDER_ERR_DEVICE; hip_msg = std::string("trajectory copy: ") + hipGetErrorString(e); }
        if (status == GORDER_OK)
            status = gorder_hip_submit_device(h, s.d_xyz, h->tables.handle_pbc ? s.d_box : nullptr, s.fidx.data(), s.n);
        if (status == GORDER_OK) {
            e = hipEventRecord(s.computed, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h_err, h->d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
            if (e != hipSuccess) { status = GORDER_ERR_DEVICE; hip_msg = std::string("trajectory submit: ") + hipGetErrorString(e); }
            s.compute_issued = true;
            frames += s.n;
            batches++;
            bytes += nx + nb;
        }
        const bool device_error = *reinterpret_cast<volatile unsigned long long *>(h_err) != kErrNone;
        {
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.free_q.push_back(k);
            if (status != GORDER_OK || device_error) pipe.stop = true;   // first error aborts the iteration (common.rs:248)
            pipe.cv.notify_all();
        }
        if (status != GORDER_OK || device_error) break;
    }
    {
        std::lock_guard<std::mutex> lk(pipe.mu);
        pipe.stop = true;
        pipe.cv.notify_all();
    }
    reader.join();
    (void)hipStreamSynchronize(copy_stream);
    const int sync_status = gorder_hip_synchronize(h);      // surfaces device errors
    if (status == GORDER_OK) status = sync_status;
    else if (!hip_msg.empty()) h->err_msg = hip_msg;
    if (status == GORDER_OK && pipe.reader_status != GORDER_XTC_OK) {
        status = GORDER_ERR_INVALID_ARGUMENT;
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
    }
    cleanup();
#undef TRAJ_TRY
    return status;
}
