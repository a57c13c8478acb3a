#!/usr/bin/env python3
"""Local leaflets at bench scale (3 072 lipids, three submits of 1 500 frames, undulation growing from submit to submit):
k_local_decide + the rows kernel for the frames it leaves open, the rows kernel alone with the bound, and the exact path for
every head (GORDER_HIP_LOCAL_NO_PRUNE) — sums, counts and exported sides must be EQUAL between the three.
  python tools/local_at_scale.py        (on the GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from gorder_amd import HipEngine
frames = 1500
system, name = bench.make_system("cg3k-local")
L = [float(x) for x in system.box]
res = {}
for mode, env in (("decide", None), ("rows", "GORDER_HIP_LOCAL_NO_DECIDE"), ("exact", "GORDER_HIP_LOCAL_NO_PRUNE")):
    for e in ("GORDER_HIP_LOCAL_NO_DECIDE", "GORDER_HIP_LOCAL_NO_PRUNE"): os.environ.pop(e, None)
    if env: os.environ[env] = "1"
    eng = HipEngine(system.tables); eng.use_torch_stream()
    keep = []
    for b, amp in enumerate((0.0, 0.4, 1.2)):            # flat, gently undulating (decided), strongly (left open)
        d_xyz, d_box = system.frames_device(frames, seed=11 + b)
        if amp:
            d_xyz[:, :, 2] += amp * torch.sin(2 * np.pi * d_xyz[:, :, 0] / L[0]) * torch.cos(2 * np.pi * d_xyz[:, :, 1] / L[1])
        keep.append((d_xyz, d_box))
        torch.cuda.synchronize()          # (the frames are made on torch's stream, the engine queues on its own)
        eng.submit_device(d_xyz, d_box, np.arange(b * frames, (b + 1) * frames))
        eng.synchronize()
    r = eng.finish()
    res[mode] = (r.sums.copy(), r.counts.copy(), eng.leaflets()[0].copy(), eng.local_decide_stats())
    print(mode, res[mode][3], "upper samples", int(r.counts[1].sum()), "of", int(r.counts[0].sum()), flush=True)
ok = all(np.array_equal(res["decide"][i], res[m][i]) for m in ("rows", "exact") for i in range(3))
print("cg3k-local", "EQUAL" if ok else "DIFFERENT")
assert ok and res["decide"][3]["submits"] >= 2 and res["rows"][3]["submits"] == 0
