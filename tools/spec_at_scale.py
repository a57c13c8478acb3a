#!/usr/bin/env python3
"""Global leaflets in one read against the two-kernel path at bench scale (three batches of thousands of frames):
sums, counts, exported sides — and the per-frame rows where the workload has them — must be EQUAL.
  python tools/spec_at_scale.py        (on the GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from gorder_amd import HipEngine
for wl, frames in (("aa256-leaflets", 4000), ("cg3k-leaflets", 1500), ("aa256-leaflets-timewise", 3000)):
    system, name = bench.make_system(wl)
    res = {}
    for mode in ("spec", "plain"):
        if mode == "plain": os.environ["GORDER_HIP_NO_SPECULATE"] = "1"
        else: os.environ.pop("GORDER_HIP_NO_SPECULATE", None)
        eng = HipEngine(system.tables); eng.use_torch_stream()
        keep = []
        for b in range(3):
            d_xyz, d_box = system.frames_device(frames, seed=5 + b)
            keep.append((d_xyz, d_box))
            torch.cuda.synchronize()      # (the frames are made on torch's stream, the engine queues on its own)
            eng.submit_device(d_xyz, d_box, np.arange(b * frames, (b + 1) * frames))
        try:
            r = eng.finish()
        except Exception as e:
            print(wl, mode, "FAILED:", e); raise
        res[mode] = (r.sums.copy(), r.counts.copy(), eng.speculation_stats(), eng.leaflets()[0].copy())
        if system.tables.timewise:
            res[mode] += tuple(np.asarray(x).copy() for x in eng.timewise(3 * frames))
    a, b = res["spec"], res["plain"]
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    if system.tables.timewise: ok = ok and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    print(wl, "EQUAL" if ok else "DIFFERENT", a[2], "upper samples", int(a[1][1].sum()), "of", int(a[1][0].sum()))
    assert ok and a[2]["batches"] == 2
