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
system.tables.flags = 1 if os.environ.get("QB_ACOS") else 0
d_xyz, d_box = system.frames_device(frames, seed=1)
eng = HipEngine(system.tables); eng.use_torch_stream()
for _ in range(2): eng.submit_device(d_xyz, d_box)
eng.synchronize(); eng.kernel_time(reset=True)
for _ in range(reps): eng.submit_device(d_xyz, d_box)
eng.synchronize()
ms, n = eng.kernel_time()
b = system.bytes_per_frame * frames
print(f"{wl} frames={frames} plan={eng.plan()} env={ {k:v for k,v in os.environ.items() if k.startswith('GORDER')} }")
print(f"  avg launch {ms/n:.4f} ms  -> {b/(ms/n*1e-3)/1e9:.1f} GB/s  ({b/(ms/n*1e-3)/8e12*100:.1f}% of 8 TB/s)  {frames/(ms/n*1e-3)/1e6:.2f} Mframes/s")
