#!/usr/bin/env python3
"""Quick kernel timing for development: python tools/quick_bench.py [workload] [frames] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gorder_amd import HipEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "aa256"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
system, name = bench.make_system(wl)
if os.environ.get("QB_ACOS"): system.tables.flags |= 1
d_xyz, d_box = system.frames_device(frames, seed=1)
eng = HipEngine(system.tables); eng.use_torch_stream()
for _ in range(int(os.environ.get('QB_WARM', '40'))): eng.submit_device(d_xyz, d_box)   # ~20 ms: clocks settle
eng.synchronize(); eng.kernel_time(reset=True)
ts, wall = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    eng.submit_device(d_xyz, d_box); eng.synchronize()
    wall.append((time.perf_counter() - t0) * 1e3)
    groups = eng.kernel_groups()
    ms, n = eng.kernel_time(reset=True); ts.append(ms)
ts = np.array(ts); b = system.bytes_per_frame * frames
env = {k: v for k, v in os.environ.items() if k.startswith('GORDER') or k.startswith('QB')}
print(f"{wl} frames={frames} plan={eng.plan()} env={env}")
f = lambda ms: f"{ms:.4f} ms {b/(ms*1e-3)/1e9:.0f} GB/s {b/(ms*1e-3)/8e12*100:.1f}%"
print("  per-launch ms:", " ".join(f"{t:.3f}" for t in ts))
print(f"  min {f(ts.min())} | median {f(np.median(ts))} | max {f(ts.max())}  ({frames/(np.median(ts)*1e-3)/1e6:.2f} Mframes/s)")
print("  groups (last submit):", "  ".join(f"{g}: {m:.4f} ms" for g, m, _ in groups))
print(f"  wall per submit (all kernels + host): median {np.median(wall):.3f} ms -> {frames/(np.median(wall)*1e-3)/1e6:.3f} Mframes/s")
