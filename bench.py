#!/usr/bin/env python3
"""bench.py — trajectory frames/s of the per-frame order-parameter path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload aa256|cg3k|cg1m] [--frames F]

A "step" is one pass of the hot path (gorder_hip_submit_device: box check + P2 kernels + leaflet
kernels if enabled) over one batch of F synthetic frames that are ALREADY resident in HBM.
Default workload = BASELINE.json configs[1]: AAOrder, 256-lipid membrane (25 088 selected atoms,
16 384 C-H bonds per frame), 10 000 frames per step, one GPU.
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank analyses its own shard of
F frames per step (frames are independent units: weak scaling, no data-path collective) and the ranks'
i64 accumulators are summed by ONE RCCL all-reduce at the end of the timed region
(= SystemTopology::reduce, /root/reference/src/analysis/topology/mod.rs:256-272).

Rank 0 prints ONE JSON line.  `roofline` is computed live from HIP events recorded on the launch
stream around the per-frame kernels; `cpu_baseline` is the oracle (C restatement of the reference
algorithm, libm trig exactly like the Rust code) timed on this box's host cores on a bounded sample.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def make_system(name):
    from gorder_amd import synthetic
    from gorder_amd.abi import LEAFLETS_GLOBAL
    if name == "aa256":
        return synthetic.aa_membrane(256), "AAOrder 256-lipid all-atom membrane (25088 atoms, 16384 C-H bonds/frame)"
    if name == "aa256-leaflets":
        return synthetic.aa_membrane(256, leaflets=LEAFLETS_GLOBAL), "AAOrder 256 lipids + global leaflets"
    if name == "cg3k":
        return synthetic.cg_membrane(3072), "CGOrder Martini bilayer 3072 lipids (36864 beads, 33792 bonds/frame)"
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
    if name == "ua256-maps":     # BASELINE configs[3]
        from gorder_amd.abi import OrderMap
        om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
        return (synthetic.ua_membrane(256, ordermap=om),
                "UAOrder 256 united-atom lipids (62 virtual C-H per lipid) + 91x91 ordermaps")
    if name == "ua256":
        return synthetic.ua_membrane(256), "UAOrder 256 united-atom lipids (62 virtual C-H per lipid)"
    if name == "cg1m":
        return synthetic.cg_membrane(83334), "CGOrder synthetic 1M-bead bilayer (1000008 beads, 916674 bonds/frame)"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(system, seconds_target=12.0):
    """Oracle (kind 'port') on the host cores: reference-faithful libm trig, one accumulator clone per
    thread + ordered reduce like groan_rs' traj_iter_map_reduce."""
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))   # a one-GPU box grants a 16-core CPU share
    n_sample = max(cores * 4, min(512, int(2e8 // max(1, system.n_atoms * 12))))
    xyz = system.frames(n_sample, seed=99)
    box = system.box9(n_sample)
    eng = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM, n_threads=cores)
    t0 = time.perf_counter()
    eng.submit(xyz, box)
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(200, seconds_target / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.submit(xyz, box)
    dt = time.perf_counter() - t0
    frames = reps * n_sample
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} synthetic frames of the same workload x {reps} passes, "
                      f"{cores} threads (libm trig, frame-interleaved threads + ordered reduce)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)   # ~20 ms: the chip needs that long to settle its clocks
    ap.add_argument("--workload", default="aa256")
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (default: 10000, cg1m: 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        entry.build()          # no-op when the in-tree libraries are current
    if world > 1:
        dist.barrier()         # nobody loads the library while rank 0 may still be linking it

    from gorder_amd import HipEngine
    system, workload = make_system(args.workload)
    frames = args.frames or (1000 if args.workload == "cg1m" else 10000)
    system.tables.device = local_rank

    # synthetic frames, resident in HBM before the timed region; each rank owns its own shard
    d_xyz, d_box = system.frames_device(frames, seed=1000 + rank, device=f"cuda:{local_rank}")
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    acc = torch.zeros(eng.accumulator_words(), dtype=torch.int64, device=f"cuda:{local_rank}")
    eng.bind_accumulators(acc)

    first = rank * frames   # global frame indices of this rank's shard (topology/mod.rs:141-144)
    fidx = np.arange(first, first + frames, dtype=np.uint64)

    def step():
        eng.submit_device(d_xyz, d_box, fidx)

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    if world > 1:   # untimed: RCCL sets up its communicator and buffers on the first collective of a size
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
    eng.flush()   # fold the kernel's accumulator replicas into the packed block (stream-ordered, tiny)
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)   # RCCL over xGMI: the only collective of the path
        if system.tables.ordermap.enabled:           # ... plus the ordermap grids when they are on (Map::add)
            n_map = 3 * eng.tables.n_acc * int(np.prod(eng.ordermap_dims()))
            ms = torch.zeros(n_map, dtype=torch.int64, device=f"cuda:{local_rank}")
            mc = torch.zeros_like(ms)
            eng.export_maps(ms, mc)
            dist.all_reduce(ms, op=dist.ReduceOp.SUM)
            dist.all_reduce(mc, op=dist.ReduceOp.SUM)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernel_ms, launches = eng.kernel_time()
    res = eng.finish()
    expect_frames = (args.steps + args.warmup) * frames * world   # warmup passes accumulate too
    ok_counts = int(res.counts[0].min()) > 0

    if rank == 0:
        # HBM traffic of the dominant kernel comes from PMC counters, which need their own rocprofv3 passes
        # (tools/pmc.sh); the committed summary for this workload and launch size is reported, else null
        traffic, traffic_src = None, None
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        for cand in sorted(glob.glob(os.path.join(prof_dir, f"r*_pmc_{args.workload}_k_bonds_tiled.json")), reverse=True):
            with open(cand) as fh:
                pmc = json.load(fh)
            if pmc.get("algorithmic_bytes_per_launch") == system.bytes_per_frame * frames and \
                    "hbm_traffic_bytes_per_launch" in pmc:
                traffic, traffic_src = pmc["hbm_traffic_bytes_per_launch"], os.path.relpath(cand, os.path.dirname(prof_dir))
                break
        total_frames = args.steps * frames * world
        value = total_frames / dt
        per_launch_bytes = system.bytes_per_frame * frames
        avg_launch_s = (kernel_ms / 1e3) / max(1, launches)
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        out = {
            "metric": "trajectory frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "frames_per_step_per_gpu": frames, "atoms": system.n_atoms,
                       "bonds_per_frame": system.tables.n_samples_per_frame,
                       "parallelism": f"frame-sharded x{world}, one RCCL int64 all-reduce at the end"
                       if world > 1 else "single GPU", "plan": eng.plan()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": "k_bonds_tiled", "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": per_launch_bytes},
            "sanity": {"frames_accumulated": res.n_frames, "expected": expect_frames, "counts_ok": ok_counts},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
