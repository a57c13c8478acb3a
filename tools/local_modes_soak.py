#!/usr/bin/env python3
"""Local leaflets, random membranes (300-3000 lipids, radius 1.2-3.5 nm, boxes 8-40 nm wide and 7-30 nm tall, flat to
strongly undulating, lipids shifted by whole boxes, a few atoms far up): k_local_sums + k_local_decide, the cell-list
order of kernels (GORDER_HIP_LOCAL_NO_SUMS), and the exact path for every head (GORDER_HIP_LOCAL_NO_PRUNE) must give
EQUAL sums, counts and exported sides.   python tools/local_modes_soak.py [cases]      (on the GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gorder_amd import HipEngine, synthetic
from gorder_amd.abi import LEAFLETS_LOCAL
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ENVS = ("GORDER_HIP_LOCAL_NO_SUMS", "GORDER_HIP_LOCAL_NO_PRUNE")
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(4000 + case)
    n_lip = int(rng.integers(300, 3000))
    side = float(rng.uniform(8, 40))
    box = (side, side * float(rng.uniform(0.7, 1.3)), float(rng.uniform(7, 30)))
    radius = float(rng.uniform(1.2, 3.5))
    system = synthetic.cg_membrane(n_lip, leaflets=LEAFLETS_LOCAL, radius=radius, n_types=int(rng.integers(1, 4)), box=box,
                                   frequency=int(rng.choice([1, 1, 2])))
    n = int(rng.integers(4, 24))
    xyz = system.frames(n, seed=case)
    amp = float(rng.choice([0.0, 0.3, 0.8, 1.6]))
    if amp:
        xyz[:, :, 2] += (amp * np.sin(2 * np.pi * xyz[:, :, 0] / box[0]) * np.cos(2 * np.pi * xyz[:, :, 1] / box[1])).astype(np.float32)
    if rng.random() < 0.3:
        k = int(rng.integers(0, n_lip - 3))
        xyz[n // 2:, 12 * k:12 * (k + 2), 2] += float(rng.integers(-2, 3)) * box[2]
        xyz[:, 12 * (k + 2):12 * (k + 3), 0] += float(rng.integers(-3, 4)) * box[0]
    if rng.random() < 0.2:
        xyz[1:, 12 * 5:12 * 6, 2] += 0.45 * box[2]
    bx = system.box9(n)
    batches = int(rng.integers(1, 4))
    res = {}
    for mode in ("sums",) + ENVS:
        for e in ENVS: os.environ.pop(e, None)
        if mode != "sums": os.environ[mode] = "1"
        eng = HipEngine(system.tables); eng.use_torch_stream()
        keep = []
        edges = np.linspace(0, n, batches + 1).astype(int)
        for a, b in zip(edges[:-1], edges[1:]):
            if b > a:
                dx, db = torch.from_numpy(xyz[a:b]).cuda(), torch.from_numpy(bx[a:b]).cuda()
                keep += [dx, db]
                eng.submit_device(dx, db, np.arange(a, b))
        try:
            r = eng.finish()
            res[mode] = (r.sums.copy(), r.counts.copy(), eng.leaflets()[0].copy())
        except Exception as err:            # the same error from every mode, please
            res[mode] = ("error", getattr(err, "status", None), getattr(err, "index", None))
    ref = res["sums"]
    ok = all((ref[0] == "error" and res[m][:2] == ref[:2]) if isinstance(ref[0], str) else
             (not isinstance(res[m][0], str) and all(np.array_equal(ref[i], res[m][i]) for i in range(3))) for m in ENVS)
    bad += not ok
    print(f"case {case}: {n_lip} lipids, box {box[0]:.1f} x {box[1]:.1f} x {box[2]:.1f}, r {radius:.2f}, undulation {amp}, {n} frames in {batches}:",
          "EQUAL" if ok else "DIFFERENT", flush=True)
print("cases", n_cases, "different", bad)
sys.exit(1 if bad else 0)
