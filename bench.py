#!/usr/bin/env python3
"""bench.py — trajectory frames/s of the per-frame order-parameter path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--frames F] [--scaling strong|weak]

A "step" is one pass of the hot path (gorder_hip_submit_device: box check + P2 kernels + leaflet kernels if
enabled) over one batch of synthetic frames that are ALREADY resident in HBM.

N = 1 (default): BASELINE.json configs[1] — AAOrder, 256-lipid membrane (25 088 selected atoms, 16 384 C-H bonds per
  frame), 10 000 frames per step.  The JSON line also carries
    roofline          the dominant kernel against the HBM roofline (HIP events on the launch stream),
    cpu_baseline      the oracle (C restatement of the reference algorithm, libm trig like the Rust code) on this box's
                      host cores, at 1 thread and at all cores,
    end_to_end        the same workload from an XTC FILE (gorder_hip_run_trajectory): frames/s from file with the frames
                      decompressed on the device (value) and by host threads (host_decode), PCIe GB/s; `solvated`:
                      the same with three times as many solvent atoms behind the analysed ones in every frame,
    scaling_reference the north_star scaling job (CG-1M, 10 000 frames) on this one GPU: the N = 1 point of the curve
                      that `--gpus N` continues.
N > 1 (launched by torch.distributed.run, one rank per GPU): the north_star scaling experiment — STRONG scaling of
  the 10 000-frame, 1 000 008-bead trajectory: rank r owns the contiguous frame shard [r F/N, (r+1) F/N), a step is the
  whole job (reset, the rank's shard, ONE RCCL all-reduce of the packed i64 accumulators = SystemTopology::reduce,
  /root/reference/src/analysis/topology/mod.rs:256-272), value = F / (max-over-ranks time per step).
  `--scaling weak` keeps per-GPU work fixed instead (F frames per GPU and step, one all-reduce at the end).
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MIN_WARMUP_S = 0.03     # the chip needs ~20-30 ms of launches to settle its clocks, whatever --warmup says


def make_system(name):
    from gorder_amd import synthetic
    from gorder_amd.abi import LEAFLETS_GLOBAL
    if name == "aa256":
        return synthetic.aa_membrane(256), "AAOrder 256-lipid all-atom membrane (25088 atoms, 16384 C-H bonds/frame)"
    if name == "aa256-leaflets":
        return synthetic.aa_membrane(256, leaflets=LEAFLETS_GLOBAL), "AAOrder 256 lipids + global leaflets"
    if name == "aa256-leaflets-maps":
        from gorder_amd.abi import OrderMap
        om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
        return (synthetic.aa_membrane(256, leaflets=LEAFLETS_GLOBAL, ordermap=om),
                "AAOrder 256 lipids + global leaflets + 91x91 ordermaps per leaflet")
    if name == "aa256-leaflets-timewise":
        return (synthetic.aa_membrane(256, leaflets=LEAFLETS_GLOBAL, timewise=True),
                "AAOrder 256 lipids + global leaflets + per-frame rows (error estimation)")
    if name == "cg3k":
        return synthetic.cg_membrane(3072), "CGOrder Martini bilayer 3072 lipids (36864 beads, 33792 bonds/frame)"
    if name == "cg3k-leaflets":
        return synthetic.cg_membrane(3072, leaflets=LEAFLETS_GLOBAL), "CGOrder Martini bilayer 3072 lipids + global leaflets"
    if name == "cg3k-local":     # BASELINE configs[2]
        from gorder_amd.abi import LEAFLETS_LOCAL
        return (synthetic.cg_membrane(3072, leaflets=LEAFLETS_LOCAL, radius=2.5),
                "CGOrder Martini bilayer 3072 lipids + local leaflets (r = 2.5 nm, every frame)")
    if name == "aa256-timewise":
        return synthetic.aa_membrane(256, timewise=True), "AAOrder 256 lipids + per-frame rows (error estimation)"
    if name == "aa256-maps":
        from gorder_amd.abi import OrderMap
        om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
        return synthetic.aa_membrane(256, ordermap=om), "AAOrder 256 lipids + 91x91 ordermaps"
    if name == "aa256-maps-timewise":
        from gorder_amd.abi import OrderMap
        om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
        return synthetic.aa_membrane(256, ordermap=om, timewise=True), "AAOrder 256 lipids + 91x91 ordermaps + per-frame rows"
    if name == "ua256-maps":     # BASELINE configs[3]
        from gorder_amd.abi import OrderMap
        om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
        return (synthetic.ua_membrane(256, ordermap=om),
                "UAOrder 256 united-atom lipids (62 virtual C-H per lipid) + 91x91 ordermaps")
    if name == "ua256-timewise":
        return synthetic.ua_membrane(256, timewise=True), "UAOrder 256 united-atom lipids + per-frame rows (error estimation)"
    if name == "ua256":
        return synthetic.ua_membrane(256), "UAOrder 256 united-atom lipids (62 virtual C-H per lipid)"
    if name in ("ua256-fast", "ua256-maps-fast"):      # the same two with GORDER_FLAG_UA_FAST_NORMALISE (opt-in, tolerance-bounded)
        from gorder_amd.abi import FLAG_UA_FAST_NORMALISE
        system, text = make_system(name[:-5])
        system.tables.flags |= FLAG_UA_FAST_NORMALISE
        return system, text + " [GORDER_FLAG_UA_FAST_NORMALISE]"
    if name == "cg3k-dynamic":   # membrane normal: dynamic (normal.rs:160-199): per lipid and frame, PCA of the heads within 2 nm
        from gorder_amd.abi import DynamicNormal, LEAFLETS_GLOBAL as LG
        system = synthetic.cg_membrane(3072, leaflets=LG)
        cloud = []
        for m in system.tables.molecule_types:
            m.normal_heads = np.asarray(m.heads, dtype=np.uint32)
            cloud.append(m.normal_heads)
        system.tables.dynamic_normal = DynamicNormal(enabled=True, radius=2.0, cloud=np.concatenate(cloud))
        return system, "CGOrder Martini bilayer 3072 lipids + dynamic membrane normals (r = 2.0 nm)"
    if name == "aa256-cylinder":  # geometry selection (geometry.rs): bonds inside a cylinder around the box centre
        from gorder_amd.abi import GEOM_CYLINDER, GEOMREF_BOX_CENTER, Geometry
        system = synthetic.aa_membrane(256)
        system.tables.geometry = Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_BOX_CENTER, invert=False, radius=3.0, orientation=2,
                                          point=(0.0, 0.0, 0.0), xdim=(0.0, 0.0), ydim=(0.0, 0.0), zdim=(0.0, 0.0),
                                          span=(-float("inf"), float("inf")), structure_box=tuple(float(x) for x in system.box))
        return system, "AAOrder 256 lipids + cylinder selection (r = 3 nm around the box centre)"
    if name == "cg1m-local-r2.5":      # a membrane beyond k_local_build's 65 536 atoms: k_local_bin / _scan / _scatter make the cell lists
        from gorder_amd.abi import LEAFLETS_LOCAL
        return (synthetic.cg_membrane(83334, leaflets=LEAFLETS_LOCAL, radius=2.5),
                "CGOrder synthetic 1M-bead bilayer + local leaflets (r = 2.5 nm, every frame): the cell-list fallback for membranes > 65536 atoms")
    if name == "aa256-leaflets-subset":   # the membrane group is an index LIST (every second atom): k_leaflets_global, not the contiguous kernel
        system = synthetic.aa_membrane(256, leaflets=LEAFLETS_GLOBAL)
        lf = system.tables.leaflets
        lf.membrane = np.ascontiguousarray(np.asarray(lf.membrane)[::2], dtype=np.uint32)
        return system, "AAOrder 256 lipids + global leaflets, membrane group = every second atom (index list: the generic leaflet kernel)"
    if name == "cg1m":
        return synthetic.cg_membrane(83334), "CGOrder synthetic 1M-bead bilayer (1000008 beads, 916674 bonds/frame)"
    raise SystemExit(f"unknown workload {name}")


def cpu_quota():
    """CPUs the container may use at once according to its cgroup (v2 cpu.max, v1 cfs quota), or None: a box that shows
    256 hardware threads through the affinity mask may still be held to 16 by a CPU quota — 64 threads then run at a quarter
    of their speed each, which is what round 3's "25.2 k frames/s on 64 threads" was."""
    def read(path):
        try:
            with open(path) as fh:
                return fh.read().split()
        except OSError:
            return None
    candidates = ["/sys/fs/cgroup/cpu.max"]
    try:
        with open("/proc/self/cgroup") as fh:
            for line in fh:
                parts = line.strip().split(":", 2)
                if len(parts) == 3 and parts[1] == "":
                    candidates.insert(0, "/sys/fs/cgroup" + parts[2].rstrip("/") + "/cpu.max")
    except OSError:
        pass
    for path in candidates:
        v = read(path)
        if v and len(v) == 2 and v[0] != "max":
            return max(1.0, float(v[0]) / float(v[1]))
    q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    if q and per and float(q[0]) > 0:
        return max(1.0, float(q[0]) / float(per[0]))
    return None


def cores_available():
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = max(1, os.cpu_count() or 1)
    return n


def host_cores():
    """Threads the CPU-side measurements use: every core this process may run on, up to GORDER_BENCH_MAX_THREADS
    (default 64: the oracle's frame-interleaved threads and the reader's copy threads stop scaling long before that;
    both numbers are reported — `cores` and `cores_available` — so a cap is never silent)."""
    cap = int(os.environ.get("GORDER_BENCH_MAX_THREADS", "64"))
    return max(1, min(cores_available(), cap))


def reader_threads():
    """Host threads for the trajectory reader (block copies / host decoder): the library's own default, at most 16.  More
    is slower on a two-socket box: tools/microbench/copy_bench.cpp on the 256-thread EPYC host of an MI355X copies
    90-147 GB/s out of the page cache with 16 threads, 40-55 with 64, 22-40 with 128."""
    return max(1, min(cores_available(), 16))


def cpu_baseline(system, seconds_target=2.0, threads=None):
    """Oracle (kind 'port') on the host cores: reference-faithful libm trig, one accumulator clone per thread +
    ordered reduce like groan_rs' traj_iter_map_reduce.  Every thread gets at least 64 frames per pass
    and the worker threads stay alive across the passes of a timing (gorder_oracle_submit_passes) — the reference keeps
    its threads for the whole trajectory (common.rs:283-339).  How many threads this box really runs at once is
    MEASURED: the scaling table 1 / 2 / 4 / ... / `host_cores()` threads is part of the block, `value` is its best row
    and `cores` that row's thread count (the affinity mask of a GPU box shows every hardware thread of the host, its CPU
    quota may be a fraction of that)."""
    from oracle import oracle
    quota = cpu_quota()
    cap = threads or host_cores()
    if quota and not threads:       # more threads than the CPU quota allows only take turns (measured: 16 -> 13.4 x, 32 -> 14.1 x, 64 -> 13.3 x on a quota of 16)
        cap = max(1, min(cap, int(quota + 0.5)))
    counts = [1]
    while counts[-1] * 2 <= cap:
        counts.append(counts[-1] * 2)
    if counts[-1] != cap:
        counts.append(cap)
    n_sample = max(cap * 64, 512)
    n_sample = min(n_sample, max(cap, int(4e9 // max(1, system.n_atoms * 12))))       # at most 4 GB of frames
    xyz = system.frames(n_sample, seed=99)
    box = system.box9(n_sample)

    def timed(n_threads):
        n_frames = min(n_sample, max(512, 64 * n_threads))
        eng = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM, n_threads=n_threads)
        t0 = time.perf_counter()
        eng.submit(xyz[:n_frames], box[:n_frames])
        t1 = time.perf_counter() - t0
        passes = int(max(1, min(5000, seconds_target / max(t1, 1e-6))))
        t0 = time.perf_counter()
        eng.submit(xyz[:n_frames], box[:n_frames], passes=passes)          # ONE call: the threads live through all passes
        return passes * n_frames / (time.perf_counter() - t0), n_frames, passes

    table = []
    for n in counts:
        v, nf, passes = timed(n)
        table.append({"threads": n, "frames_per_s": v, "frames": nf, "passes": passes})
    one = table[0]["frames_per_s"]
    best = max(table, key=lambda r: r["frames_per_s"])
    for r in table:
        r["speedup"] = r["frames_per_s"] / one
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "cores_available": cores_available(),
            "cpu_quota": quota, "kind": "port", "value_1_thread": one,
            "parallel_efficiency": best["frames_per_s"] / (best["threads"] * one),
            "scaling": table,
            "sample": f"{best['frames']} synthetic frames of the same workload ({best['frames'] // best['threads']} per thread) x "
                      f"{best['passes']} passes on {best['threads']} threads that live through all passes — the best row of the "
                      f"scaling table over {counts} threads, ~{seconds_target:.0f} s each (1 thread: {table[0]['frames']} frames x "
                      f"{table[0]['passes']} passes); libm trig, frame-interleaved threads + ordered reduce"}


def warm_up(step, sync, n_steps, agree=None):
    """At least n_steps passes AND at least MIN_WARMUP_S of launches.  With several ranks `agree(flag)` returns the OR
    of the ranks' flags, so that every rank runs the same number of passes (the passes may contain collectives)."""
    t0 = time.perf_counter()
    done = 0
    for _ in range(n_steps):
        step()
        done += 1
    sync()
    while True:
        more = time.perf_counter() - t0 < MIN_WARMUP_S
        if agree is not None:
            more = agree(more)
        if not more:
            break
        for _ in range(8):
            step()
            done += 1
        sync()
    return done


def copy_ceilings():
    """What bounds the from-file rate when the decoder is not it: the PCIe link (one pinned GiB copied to the device,
    three times) — to be read next to `pcie_GBps` (what the device route moved) and `seconds.host_reader` (the share of
    the call the reader thread was busy copying compressed blocks out of the page cache)."""
    import torch
    pinned = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    return {"pcie_link_GBps": 3 * (1 << 30) / (time.perf_counter() - t0) / 1e9}


def end_to_end(system, device_index, n_unique=500, repeats=200):
    """The workload from an XTC FILE through gorder_hip_run_trajectory: the repo's encoder writes n_unique synthetic
    frames (precision 1000 like GROMACS), the file is read `repeats` times as one concatenated trajectory."""
    from gorder_amd import HipEngine, xtc
    cores = reader_threads()
    xyz = system.frames(n_unique, seed=4242)
    box = system.box9(n_unique)
    with tempfile.TemporaryDirectory(prefix="gorder_bench_") as tmp:
        path = os.path.join(tmp, "traj.xtc")
        t0 = time.perf_counter()
        xtc.write_trajectory(path, xyz, box, times=np.arange(n_unique, dtype=np.float32) * 10.0, precision=1000.0)
        t_write = time.perf_counter() - t0
        size = os.path.getsize(path)
        system.tables.device = device_index
        eng = HipEngine(system.tables)
        runs = {}
        for route, dev in (("host_decode", False), ("device_decode", True)):
            eng.reset()
            # warm: page cache, kernels, and the handle's staging buffers (sized to the trajectory: the same file list)
            first = eng.run_trajectory([path] * repeats, threads=cores, device_decode=dev)
            eng.reset()
            stats = eng.run_trajectory([path] * repeats, threads=cores, device_decode=dev)
            res = eng.finish()
            assert stats["n_frames"] == n_unique * repeats == res.n_frames
            assert stats["device_decode"] <= int(dev)      # (frames too large for one-lane-per-frame decoding: host decoder)
            stats["first_call_setup"] = first["seconds_setup"]
            stats["first_call_value"] = first["n_frames"] / first["seconds_total"]
            runs[route] = (stats, res)
        eng.close()
        ceilings = copy_ceilings()
    np.testing.assert_array_equal(runs["host_decode"][1].sums, runs["device_decode"][1].sums)   # same coordinates, same sums

    def block(stats):
        n, sec = stats["n_frames"], stats["seconds_total"]
        return {"value": n / sec, "unit": "frames/s", "first_call_value": stats["first_call_value"], "frames": n, "host_threads": stats["decoder_threads"],
                "batch_frames": stats["batch_frames"], "batches": stats["n_batches"], "decoded_on": "device" if stats["device_decode"] else "host",
                "pcie_GBps": stats["bytes_h2d"] / sec / 1e9, "file_read_MBps": size * repeats / sec / 1e6,
                "seconds": {"total": sec, "setup": stats["seconds_setup"],
                            "setup_of_the_handles_first_call": stats["first_call_setup"], "host_reader": stats["seconds_decode"],
                            "reader_waiting_for_gpu": stats["seconds_reader_stalled"],
                            "gpu_waiting_for_reader": stats["seconds_gpu_starved"]},
                "bottleneck": "host reader" if stats["seconds_gpu_starved"] > stats["seconds_reader_stalled"] else "copy/kernels"}

    out = block(runs["device_decode"][0])
    out["host_decode"] = block(runs["host_decode"][0])
    # which of the two bounds the device route: the compressed bytes it moves per second against the link and the reader
    out["ceilings"] = ceilings
    out["ceilings"]["device_route_moves_GBps"] = out["pcie_GBps"]
    out["ceilings"]["reader_busy_fraction"] = out["seconds"]["host_reader"] / out["seconds"]["total"]
    out["ceilings"]["bound_by"] = ("host reader (copies out of the page cache)" if out["ceilings"]["reader_busy_fraction"] > 0.8
                                   and out["pcie_GBps"] < 0.9 * ceilings["pcie_link_GBps"] else "PCIe link / kernels")
    out["file_MB"] = size * repeats / 1e6
    out["path"] = ("XTC file (repo encoder, precision 1000, %d frames read %d x as one concatenated trajectory, encoded in "
                   "%.1f s) -> gorder_hip_run_trajectory.  value: device_decode = host threads copy the compressed blocks "
                   "(gorder_xtc_pack_window) -> pinned staging x3 -> hipMemcpyAsync -> k_xtc_scan + k_xtc_chunks (a lane per 256-atom chunk) "
                   "-> kernels.  host_decode: gorder_xtc_read_window_mt on the same threads -> pinned -> hipMemcpyAsync "
                   "-> kernels.  Both routes give identical sums (checked)." % (n_unique, repeats, t_write))
    return out


def end_to_end_large(system, name, device_index, hot_value):
    """From-file throughput on a trajectory that no cache level between the disk and the DRAM holds: GORDER_BENCH_LARGE_GB
    (default 8) gigabytes of DISTINCT encoder-written frames (tools/make_large_xtc.py, one process per part file, all
    parts one concatenated trajectory), analysed by gorder_hip_run_trajectory (device decode)
      cold   after fsync + posix_fadvise(POSIX_FADV_DONTNEED) on every part: the bytes come from the disk,
      warm   again, right after: the bytes come from the page cache (DRAM) — 8 GB do not fit the host's L3, which is
             what the hot-file figure (one 62-MB file read 200 times) profits from.
    Returns a dict; {"skipped": reason} when the box has no room for the files."""
    import shutil
    import subprocess
    from gorder_amd import HipEngine
    gb = float(os.environ.get("GORDER_BENCH_LARGE_GB", "8"))
    if gb <= 0:
        return {"skipped": "GORDER_BENCH_LARGE_GB=0"}
    bytes_per_frame = 125e3 * system.n_atoms / 25088.0           # XTC at precision 1000: ~5 bytes per atom
    n_frames = int(gb * 1e9 / bytes_per_frame)
    workers = max(1, min(16, int(cpu_quota() or cores_available()), cores_available()))
    per = (n_frames + workers - 1) // workers
    n_frames = per * workers
    tmp_root = os.environ.get("GORDER_BENCH_TMPDIR") or tempfile.gettempdir()
    try:
        free = shutil.disk_usage(tmp_root).free
    except OSError as ex:
        return {"skipped": f"{tmp_root}: {ex}"}
    if free < 1.5 * gb * 1e9:
        return {"skipped": f"{tmp_root} has {free / 1e9:.1f} GB free, {1.5 * gb:.0f} GB wanted"}
    tmp = tempfile.mkdtemp(prefix="gorder_bench_large_", dir=tmp_root)
    out = {"frames": n_frames, "parts": workers}
    try:
        paths = [os.path.join(tmp, f"part{w:02d}.xtc") for w in range(workers)]
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "make_large_xtc.py"), name, paths[w],
                                   str(w * per), str(per)]) for w in range(workers)]
        if any(p.wait() != 0 for p in procs):
            return {"skipped": "tools/make_large_xtc.py failed"}
        out["seconds_to_write"] = time.perf_counter() - t0
        size = sum(os.path.getsize(p) for p in paths)
        out["file_GB"] = size / 1e9
        dropped = True
        for p in paths:          # the page cache lets go of clean pages only: write them out first
            fd = os.open(p, os.O_RDONLY)
            try:
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            except OSError:
                dropped = False
            finally:
                os.close(fd)
        system.tables.device = device_index
        eng = HipEngine(system.tables)
        threads = reader_threads()

        def block(st):
            sec = st["seconds_total"]
            return {"value": st["n_frames"] / sec, "unit": "frames/s", "seconds": sec, "file_read_MBps": size / sec / 1e6,
                    "pcie_GBps": st["bytes_h2d"] / sec / 1e9, "reader_busy_fraction": st["seconds_decode"] / sec,
                    "gpu_waiting_for_reader_s": st["seconds_gpu_starved"], "batches": st["n_batches"]}
        cold = eng.run_trajectory(paths, threads=threads, device_decode=True)
        res_cold = eng.finish()
        eng.reset()
        warm = eng.run_trajectory(paths, threads=threads, device_decode=True)
        res_warm = eng.finish()
        eng.reset()
        warm2 = eng.run_trajectory(paths, threads=threads, device_decode=True)
        eng.close()
        assert cold["n_frames"] == warm["n_frames"] == n_frames == res_cold.n_frames
        np.testing.assert_array_equal(res_cold.sums, res_warm.sums)
        out["cold"] = block(cold)
        out["cold"]["page_cache_dropped"] = dropped
        out["cold"]["note"] = ("first read after fsync + posix_fadvise(DONTNEED): disk -> page cache -> pinned -> device; also "
                               "this handle's first call (staging buffers are pinned inside it)")
        out["warm"] = block(warm if warm["seconds_total"] <= warm2["seconds_total"] else warm2)
        out["warm"]["note"] = "read again: every byte comes out of the page cache in DRAM (the better of two runs)"
        out["hot_file_value"] = hot_value
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def end_to_end_solvated(system, device_index, n_unique=100, repeats=200, water_per_atom=3):
    """The same workload as a real membrane simulation stores it: the analysed atoms in front of `water_per_atom` times
    as many solvent atoms.  The host decoder stops at the last analysed atom; the device route learns from the decoder's
    first report which leading part of every compressed block it needs and copies only that."""
    from gorder_amd import HipEngine, xtc
    cores = reader_threads()
    rng = np.random.default_rng(7)
    n_sel = system.n_atoms
    n_w = water_per_atom * n_sel
    span = float(system.box[0])
    centres = rng.uniform(0.0, span, size=(n_w // 3, 1, 3))
    w0 = (centres + rng.normal(0.0, 0.05, size=(n_w // 3, 3, 3))).reshape(-1, 3)
    water = (w0[None] + rng.normal(0.0, 0.03, size=(n_unique, n_w, 3))).astype(np.float32)
    xyz = np.concatenate([system.frames(n_unique, seed=11), water], axis=1)
    group = np.arange(n_sel, dtype=np.uint32)
    out = {"atoms_in_file": int(xyz.shape[1]), "atoms_analysed": int(n_sel), "frames": n_unique * repeats}
    with tempfile.TemporaryDirectory(prefix="gorder_bench_") as tmp:
        path = os.path.join(tmp, "solvated.xtc")
        xtc.write_trajectory(path, xyz, system.box9(n_unique), precision=1000.0)
        out["compressed_bytes_per_frame"] = os.path.getsize(path) / n_unique
        system.tables.device = device_index
        eng = HipEngine(system.tables)
        sums = {}
        for route, dev in (("host_decode", False), ("device_decode", True)):
            eng.reset()
            first = eng.run_trajectory([path] * repeats, group=group, threads=cores, device_decode=dev)     # warm, same shape
            eng.reset()
            st = eng.run_trajectory([path] * repeats, group=group, threads=cores, device_decode=dev)
            sums[route] = eng.finish().sums
            out[route] = {"first_call_value": first["n_frames"] / first["seconds_total"], "value": st["n_frames"] / st["seconds_total"], "unit": "frames/s",
                          "pcie_GBps": st["bytes_h2d"] / st["seconds_total"] / 1e9,
                          "decoded_on": "device" if st["device_decode"] else "host",
                          "frames_decoded_by_host_after_all": st["frames_decoded_by_host"]}
        eng.close()
    np.testing.assert_array_equal(sums["host_decode"], sums["device_decode"])
    return out


class Watchdog:
    """Extra blocks of the N > 1 line run collectives after the headline is final.  If a rank fails alone, the others would
    wait for it forever and the driver would lose the line: after `seconds` without disarm() rank 0 prints the line as it
    stands (the block marked as timed out) and every rank leaves with exit code 0."""

    def __init__(self, seconds, rank, line):
        import threading
        self.rank, self.line, self.lock, self.done = rank, line, threading.Lock(), False
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.timer.start()

    def fire(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                out = self.line()
                out.setdefault("extras_timed_out", True)
                print(json.dumps(out), flush=True)
            sys.stderr.write(f"[rank {self.rank}] bench: an extra block did not finish; leaving with the headline\n")
            sys.stderr.flush()
            os._exit(0)

    def disarm(self):
        with self.lock:
            self.done = True
        self.timer.cancel()


def all_agree(dist, ok, tdev):
    """MIN over ranks of a success flag — called OUTSIDE any try block, so that every rank reaches it."""
    import torch
    t = torch.tensor([1 if ok else 0], device=tdev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def lib_allreduce_check(dist, rank, world, device, rehearsal, tables, d_xyz, d_box, fidx, want):
    """The library's own RCCL route on N ranks (gorder_hip_comm_unique_id / _create / gorder_hip_allreduce — what a Rust
    host without a collective library calls; SystemTopology::reduce, topology/mod.rs:256-272): a second handle per rank
    analyses the rank's shard once, the library reduces, and the result must equal what torch.distributed reduced
    (`want`: the strong-scaling job's last step, or None).  Errors are reported, never raised; every collective sits
    outside the try blocks and is entered by all ranks or none."""
    import torch
    from gorder_amd import HipEngine
    tdev = "cpu" if rehearsal else device
    out = {"ok": False, "equal_to_torch": None, "ms": None, "ranks": world}
    ids = [None]
    if rank == 0:
        try:
            ids[0] = HipEngine.comm_unique_id()
        except Exception as ex:   # noqa: BLE001
            out["error"] = "gorder_hip_comm_unique_id: " + repr(ex)
    dist.broadcast_object_list(ids, src=0)
    if ids[0] is None:
        out.setdefault("error", "no unique id from rank 0")
        return out
    eng2, comm, err = None, None, None
    try:
        eng2 = HipEngine(tables)
        comm = eng2.comm_create(ids[0], world, rank)
        eng2.submit_device(d_xyz, d_box, fidx)
        eng2.synchronize()
    except Exception as ex:   # noqa: BLE001
        err = "setup: " + repr(ex)
    if not all_agree(dist, err is None, tdev):
        out["error"] = err or "another rank failed to create its communicator"
    else:
        t0 = time.perf_counter()
        try:
            eng2.allreduce(comm)
            eng2.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            got = eng2.finish()
            out["ms"] = ms
            out["ok"] = True
            if want is not None:
                out["equal_to_torch"] = bool(np.array_equal(got.sums, want.sums) and np.array_equal(got.counts, want.counts)
                                             and got.n_frames == want.n_frames)
        except Exception as ex:   # noqa: BLE001
            out["error"] = "gorder_hip_allreduce: " + repr(ex)
        ok_all = all_agree(dist, out["ok"], tdev)
        out["ok_on_all_ranks"] = ok_all
    try:
        if comm is not None:
            eng2.comm_destroy(comm)
        if eng2 is not None:
            eng2.close()
    except Exception as ex:   # noqa: BLE001
        out.setdefault("error", "teardown: " + repr(ex))
    return out


def end_to_end_sharded(dist, rank, world, local_rank, device, rehearsal, n_unique=500, repeats_per_rank=40):
    """N > 1: the path a multi-GPU host really takes from a FILE.  Every rank opens the SAME trajectory and calls
    gorder_hip_run_trajectory with shard_index = rank, shard_count = world (contiguous shares of the selected frames,
    SURVEY §8e; frames decompressed on the device), then ONE all-reduce of the packed accumulators
    (SystemTopology::reduce, topology/mod.rs:256-278).  Timed between barriers, max over ranks; afterwards rank 0
    analyses the whole trajectory on one handle and the reduced sums must equal that bit for bit."""
    import shutil
    import torch
    from gorder_amd import HipEngine, xtc
    system, workload = make_system("aa256")
    system.tables.device = local_rank
    cpu = torch.device("cpu")
    tdev = cpu if rehearsal else device
    box = [None]
    if rank == 0:
        tmp = tempfile.mkdtemp(prefix="gorder_bench_shards_")
        path = os.path.join(tmp, "traj.xtc")
        xtc.write_trajectory(path, system.frames(n_unique, seed=4242), system.box9(n_unique),
                             times=np.arange(n_unique, dtype=np.float32) * 10.0, precision=1000.0)
        box[0] = path
    dist.broadcast_object_list(box, src=0)          # one node: every rank sees rank 0's file
    path = box[0]
    repeats = repeats_per_rank * world
    paths = [path] * repeats
    threads = max(1, min(16, cores_available() // world))
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    acc = torch.zeros(eng.accumulator_words(), dtype=torch.int64, device=device)
    eng.bind_accumulators(acc)
    eng.run_trajectory(paths, threads=threads, device_decode=True, shard=(rank, world))      # warm: staging, page cache
    eng.reset()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = eng.run_trajectory(paths, threads=threads, device_decode=True, shard=(rank, world))
    t_run = time.perf_counter() - t0
    eng.flush()
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, t_run, float(stats["n_frames"]), float(stats["shard_first"])], dtype=torch.float64, device=tdev)
    rows = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(rows, t)
    rows = [[float(x) for x in r.tolist()] for r in rows]
    out = None
    if rank == 0:
        total = n_unique * repeats
        reduced = acc.cpu().numpy().copy()
        one = HipEngine(system.tables)
        one.run_trajectory(paths, threads=threads, device_decode=True)
        want = one.finish()
        n_acc = system.tables.n_acc
        same = bool(np.array_equal(reduced[:n_acc], want.sums[0]) and np.array_equal(reduced[2 * n_acc:3 * n_acc].astype(np.uint64), want.counts[0])
                    and int(reduced[4 * n_acc]) == want.n_frames == total)
        one.close()
        out = {"value": total / max(r[0] for r in rows), "unit": "frames/s", "frames": total, "workload": workload,
               "path": "one XTC trajectory (%d frames: %d unique x %d), every rank: gorder_hip_run_trajectory(shard_index = rank, "
                       "shard_count = %d, device_decode) -> one all-reduce of the packed accumulators" % (total, n_unique, repeats, world),
               "host_threads_per_rank": threads,
               "per_rank": [{"seconds_run_trajectory": r[1], "frames": int(r[2]), "first_frame_of_shard": int(r[3])} for r in rows],
               "frames_of_all_shards": int(sum(r[2] for r in rows)),
               "equal_to_one_handle": same}
        # (reported, not asserted: a failed check must not cost the other ranks a hang and the driver its curve)
        if not same or out["frames_of_all_shards"] != total:
            print("[bench] sharded end_to_end: the reduced sums differ from one handle over the whole trajectory", file=sys.stderr)
    eng.close()
    dist.barrier()
    if rank == 0:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    return out


def scaling_reference(device, steps=5):
    """The north_star scaling job on ONE GPU: CG-1M, 10 000 resident frames (120 GB), one pass per step."""
    import torch
    from gorder_amd import HipEngine
    system, workload = make_system("cg1m")
    frames = 10000
    system.tables.device = device.index
    d_xyz, d_box = system.frames_device(frames, seed=1000, device=str(device))
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    fidx = np.arange(frames, dtype=np.uint64)

    def step():
        eng.reset()
        eng.submit_device(d_xyz, d_box, fidx)
        eng.flush()

    warm_up(step, eng.synchronize, 2)
    eng.kernel_time(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_time()
    res = eng.finish()
    ok = bool((res.counts[0] == frames * 83334).all()) and res.n_frames == frames
    avg = kernel_ms / 1e3 / max(1, launches)
    del d_xyz, d_box
    eng.close()
    torch.cuda.empty_cache()
    return {"workload": workload + ", 10000 frames resident (120 GB), one job per step", "value": frames * steps / dt,
            "unit": "frames/s", "n_gpus": 1, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "roofline_frac": system.bytes_per_frame * frames / avg / 1e9 / HBM_PEAK_GBS if avg > 0 else None,
            "result_ok": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default=None, help="default: aa256 on one GPU, cg1m on several")
    ap.add_argument("--frames", type=int, default=0,
                    help="strong scaling: frames of the whole trajectory (default 10000); weak / one GPU: frames per "
                         "step and GPU (default 10000, cg1m: 1000)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None, help="N > 1 only; default strong")
    ap.add_argument("--collective", choices=["lib", "torch"], default="torch",
                    help="N > 1: torch.distributed.all_reduce on the handle's stream (default), or gorder_hip_allreduce — "
                         "RCCL called by the library itself, the route of a host without a collective library")
    ap.add_argument("--trig", choices=["squared", "acos"], default="squared",
                    help="squared: P2 from the squared cosine (library default); acos: the reference's literal acos -> cos "
                         "round trip (GORDER_FLAG_TRIG_ACOS_COS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-scaling-reference", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # rehearsal on a one-GPU box: GORDER_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (RCCL cannot run
    # two ranks on one device); the driver's multi-GPU runs never set it
    rehearsal = os.environ.get("GORDER_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        args.collective = "torch"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # ONE stream for everything that touches the results: the handle's launches, torch's collectives and the timing
    # events.  (torch's default stream is the NULL stream, which gorder_hip_set_stream reads as "use your own": the
    # handle's launches would then not be ordered with torch.distributed's collectives.)
    torch.cuda.set_stream(torch.cuda.Stream(device))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    if rank == 0:
        entry.build()          # no-op when the in-tree libraries are current
    if world > 1:
        dist.barrier()         # nobody loads the library while rank 0 may still be linking it

    from gorder_amd import HipEngine
    strong = world > 1 and (args.scaling or "strong") == "strong"
    name = args.workload or ("cg1m" if world > 1 else "aa256")
    system, workload = make_system(name)
    system.tables.device = local_rank
    if args.trig == "acos":
        from gorder_amd.abi import FLAG_TRIG_ACOS_COS
        system.tables.flags |= FLAG_TRIG_ACOS_COS
        workload += " [acos -> cos round trip like the reference]"
    if strong:
        total = args.frames or 10000
        edges = np.linspace(0, total, world + 1).astype(np.int64)      # contiguous frame shards
        first, frames = int(edges[rank]), int(edges[rank + 1] - edges[rank])
    else:
        frames = args.frames or (1000 if name == "cg1m" else 10000)
        total = frames * world
        first = rank * frames   # global frame indices of this rank's shard (topology/mod.rs:141-144)

    # synthetic frames, resident in HBM before the timed region; each rank owns its own shard
    d_xyz, d_box = system.frames_device(frames, seed=1000 + rank, device=str(device))
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    acc = torch.zeros(eng.accumulator_words(), dtype=torch.int64, device=device)
    eng.bind_accumulators(acc)
    fidx = np.arange(first, first + frames, dtype=np.uint64)

    # ---- the collective: one all-reduce of the packed accumulators (+ ordermap grids when they are on)
    comm, collective = None, None
    if world > 1:
        if args.collective == "lib":
            try:      # the C-ABI route a non-Python host takes: RCCL called from inside the library
                ids = [HipEngine.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                comm = eng.comm_create(ids[0], world, rank)
                collective = "gorder_hip_allreduce: ncclAllReduce(int64, sum) issued by the library on the handle's stream"
            except Exception as e:   # noqa: BLE001 — plumbing only: fall back to torch's RCCL binding
                comm = None
                print(f"[rank {rank}] library collective unavailable ({e}); using torch.distributed", file=sys.stderr)
        ok = torch.tensor([1 if comm is not None else 0], device=device if not rehearsal else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                eng.comm_destroy(comm)
            comm = None
            collective = "torch.distributed.all_reduce (%s)" % ("gloo, rehearsal" if rehearsal else "RCCL")
    maps_on = bool(system.tables.ordermap.enabled)

    def reduce_results():
        if comm is not None:
            eng.allreduce(comm)              # accumulators and maps, in place, stream-ordered
            return
        eng.flush()   # fold the kernel's accumulator replicas into the packed block (stream-ordered, tiny)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        if maps_on:   # ... plus the ordermap grids (Map::add)
            n_map = 3 * eng.tables.n_acc * int(np.prod(eng.ordermap_dims()))
            ms = torch.zeros(n_map, dtype=torch.int64, device=device)
            mc = torch.zeros_like(ms)
            eng.export_maps(ms, mc)
            dist.all_reduce(ms, op=dist.ReduceOp.SUM)
            dist.all_reduce(mc, op=dist.ReduceOp.SUM)

    def step():
        if strong:               # one step = the whole job: fresh accumulators, the rank's shard, the reduce
            eng.reset()
            eng.submit_device(d_xyz, d_box, fidx)
            reduce_results()
        else:
            eng.submit_device(d_xyz, d_box, fidx)

    def agree(flag):
        f = torch.tensor([1 if flag else 0], device=device if not rehearsal else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        return bool(f.item())

    n_warm = warm_up(step, eng.synchronize, args.warmup, agree if world > 1 else None)
    if world > 1 and not strong:   # untimed: RCCL sets up its buffers on the first collective of a size
        dist.all_reduce(torch.zeros_like(acc), op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    eng.kernel_time(reset=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1 and not strong:
        reduce_results()           # weak scaling: the only collective of the path, once at the end
    elif world == 1:
        eng.flush()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if not rehearsal else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernel_groups = eng.kernel_groups()    # [(name, ms, segments)] per kernel group of the timed submits: adds up to kernel_ms
    kernel_ms, launches = eng.kernel_time()
    kernel_names = eng.kernel_names()      # what the timed region really launched (gorder_hip_kernel_time_names)
    res = eng.finish()
    allreduce_ms = None
    if world > 1:                  # the collective alone, after the timed region (what a step pays for it)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            reduce_results()
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) / 10 * 1e3
    if strong:
        expect_frames = total                                   # every step starts from fresh accumulators
        # every slot holds total x (molecules of its type) samples after the last step's reduce
        ok_counts = int(res.counts[0].min()) > 0 and bool((res.counts[0] % np.uint64(total) == 0).all())
    else:
        expect_frames = (args.steps + n_warm) * frames * world   # warm-up passes accumulate too
        ok_counts = int(res.counts[0].min()) > 0
    per_rank = per_rank_ms = per_rank_dev = None
    if world > 1:                  # per-rank fraction of the HBM roofline on the rank's own shard
        avg = kernel_ms / 1e3 / max(1, launches)
        mine = torch.tensor([system.bytes_per_frame * frames / avg / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0,
                             avg * 1e3, float(torch.cuda.current_device())],
                            dtype=torch.float64, device=device if not rehearsal else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [float(g[0].item()) for g in gathered]
        per_rank_ms = [float(g[1].item()) for g in gathered]
        per_rank_dev = [int(g[2].item()) for g in gathered]

    if rank == 0:
        # HBM traffic of the dominant kernel comes from PMC counters, which need their own rocprofv3 passes
        # (tools/pmc.sh); the committed summary for this workload and launch size is reported, else null
        traffic, traffic_src = None, None
        prof_dir = os.path.join(ROOT, "profiles")
        # the step as the device saw it: every kernel group a submit queues, timed between events on the launch stream
        # (leaflet kernels, normals, order kernels, map accumulation, k_batch_end); the DOMINANT group is the longest one
        per_launch_bytes = system.bytes_per_frame * frames
        groups = [{"name": n, "ms": ms / max(1, launches),
                   "share_of_step": ms / kernel_ms if kernel_ms > 0 else 0.0,
                   # (the step's algorithmic bytes over this group's time: what the group would reach alone; meaningless for
                   # the bookkeeping launches, hence only for groups that are at least 2 % of the step)
                   "frac": per_launch_bytes / (ms / max(1, launches) / 1e3) / 1e9 / HBM_PEAK_GBS if ms > 0.02 * kernel_ms else None,
                   "launches_per_step": seg / max(1, launches)} for n, ms, seg in kernel_groups]
        dominant = max(groups, key=lambda g: g["ms"]) if groups else {"name": "k_bonds_tiled", "ms": 0.0}
        first_kernel = dominant["name"].split(" + ")[0]
        for cand in sorted(glob.glob(os.path.join(prof_dir, f"r*_pmc_{name}_{first_kernel}*.json")), reverse=True):
            with open(cand) as fh:
                pmc = json.load(fh)
            if not pmc.get("algorithmic_bytes_per_launch") or "hbm_traffic_bytes_per_launch" not in pmc:
                continue
            if pmc["algorithmic_bytes_per_launch"] == system.bytes_per_frame * frames:
                traffic, traffic_src = pmc["hbm_traffic_bytes_per_launch"], os.path.relpath(cand, ROOT)
            else:   # another launch size: the kernel streams, its traffic per algorithmic byte does not depend on the size
                ratio = pmc["hbm_traffic_bytes_per_launch"] / pmc["algorithmic_bytes_per_launch"]
                traffic = ratio * system.bytes_per_frame * frames
                traffic_src = "%s, measured at %d bytes per launch (x %.4f of the algorithmic bytes), scaled to this launch" % (
                    os.path.relpath(cand, ROOT), pmc["algorithmic_bytes_per_launch"], ratio)
            break
        total_frames = args.steps * total
        value = total_frames / dt
        step_s = (kernel_ms / 1e3) / max(1, launches)             # device time of a whole step (all groups)
        avg_launch_s = dominant["ms"] / 1e3                        # ... of its dominant kernel group
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        whole_step = per_launch_bytes / step_s / 1e9 if step_s > 0 else 0.0
        if world == 1:
            parallelism = "single GPU"
        elif strong:
            parallelism = (f"strong scaling: {total} frames cut into {world} contiguous shards, every step = reset + shard "
                           f"+ one all-reduce of {eng.accumulator_words()} int64 words")
        else:
            parallelism = f"weak scaling: frame-sharded x{world}, one int64 all-reduce at the end"
        out = {
            "metric": "trajectory frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "frames_per_step_per_gpu": frames, "frames_per_step": total,
                       "atoms": system.n_atoms, "bonds_per_frame": system.tables.n_samples_per_frame,
                       "parallelism": parallelism, "collective": collective, "warmup_steps_run": n_warm,
                       "plan": eng.plan()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": dominant["name"], "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": per_launch_bytes, "frac_per_rank": per_rank,
                         # every kernel group of the step; `ms` add up to whole_step_ms, `share_of_step` to 1
                         "kernels": groups, "timed": kernel_names, "whole_step_ms": step_s * 1e3,
                         "whole_step_achieved": whole_step, "whole_step_frac": whole_step / HBM_PEAK_GBS},
            "sanity": {"frames_accumulated": res.n_frames, "expected": expect_frames, "counts_ok": ok_counts},
        }
        if allreduce_ms is not None:
            out["allreduce_ms"] = allreduce_ms
        if world > 1:      # what the collective backend itself says about the job, so that the line explains itself
            try:
                nccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:   # noqa: BLE001
                nccl_version = None
            out["ranks"] = {"launched": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                            "rccl_version": nccl_version, "collective_issued_by": "library (gorder_hip_allreduce)" if comm is not None
                            else "torch.distributed", "kernel_ms_per_step_per_rank": per_rank_ms, "device_per_rank": per_rank_dev,
                            "rehearsal_on_one_gpu": rehearsal}
    if world == 1:
        # release the headline workload's frames before the secondary measurements
        del d_xyz, d_box
        eng.close()
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system)
        if not args.no_end_to_end:
            out["end_to_end"] = e2e = end_to_end(system, local_rank)
            e2e["value_is"] = "hot_file: one 62-MB file read 200 x as one trajectory (host L3 / page cache resident)"
            if name == "aa256":
                e2e["solvated"] = end_to_end_solvated(system, local_rank)
                try:
                    large = end_to_end_large(system, name, local_rank, e2e["value"])
                except Exception as ex:   # noqa: BLE001 — a full disk must not cost the line
                    large = {"skipped": repr(ex)}
                e2e["large"] = large
                if "warm" in large:
                    # the figure to quote: distinct frames, larger than any cache but DRAM
                    e2e["hot_file"] = {"value": e2e["value"], "first_call_value": e2e["first_call_value"], "pcie_GBps": e2e["pcie_GBps"],
                                       "file_read_MBps": e2e["file_read_MBps"]}
                    e2e["value"] = large["warm"]["value"]
                    e2e["pcie_GBps"] = large["warm"]["pcie_GBps"]
                    e2e["file_read_MBps"] = large["warm"]["file_read_MBps"]
                    e2e["value_is"] = ("large.warm: %.1f GB of distinct frames in %d files, read from the page cache (DRAM); "
                                       "large.cold = the same from the disk, hot_file = one 62-MB file read 200 x" % (large["file_GB"], large["parts"]))
                    e2e["path"] = "value = large.warm.  " + e2e["path"]
        if not args.no_scaling_reference and name == "aa256" and not args.frames and args.trig == "squared":
            free_b, _ = torch.cuda.mem_get_info(device)
            if free_b > 150 * (1 << 30):
                out["scaling_reference"] = scaling_reference(device)
    if world > 1:
        # The headline above is final.  The extra blocks below run collectives; a rank that fails alone would leave the
        # others waiting, so a watchdog prints the line as it stands and ends every rank if they do not come back.
        wd = Watchdog(float(os.environ.get("GORDER_BENCH_EXTRAS_TIMEOUT", "600")), rank, lambda: dict(out) if rank == 0 else {})
        # (1) the library's own RCCL route on all ranks, on a second handle over the same shard
        try:
            lib = lib_allreduce_check(dist, rank, world, device, rehearsal, system.tables, d_xyz, d_box, fidx,
                                      res if strong else None)
        except Exception as ex:   # noqa: BLE001
            lib = {"ok": False, "error": repr(ex)}
        if rank == 0:
            out["lib_allreduce"] = lib
        # (2) the sharded trajectory from a FILE
        if not args.no_end_to_end:
            del d_xyz, d_box
            torch.cuda.empty_cache()
            try:
                e2e = end_to_end_sharded(dist, rank, world, local_rank, device, rehearsal)
            except Exception as ex:   # noqa: BLE001
                e2e = {"error": repr(ex)}
            if rank == 0:
                out["end_to_end"] = e2e
        wd.disarm()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import threading
        t_end = threading.Timer(60.0, lambda: os._exit(0))     # the line is out: a stuck teardown must not turn into a failed run
        t_end.daemon = True
        t_end.start()
        dist.barrier()
        if comm is not None:
            eng.comm_destroy(comm)
        dist.destroy_process_group()
        t_end.cancel()


if __name__ == "__main__":
    main()
