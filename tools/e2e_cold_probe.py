#!/usr/bin/env python3
"""A handle's FIRST trajectory, short ones included: wall time of one gorder_hip_run_trajectory call on a fresh handle
(buffer setup and pipeline ramp inside), both decode routes.   python tools/e2e_cold_probe.py [workload] [frames ...]"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gorder_amd import HipEngine, xtc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "aa256"
sizes = [int(x) for x in sys.argv[2:]] or [1000, 5000, 20000]
system, desc = bench.make_system(name)
cores = bench.host_cores()
unique = 500
with tempfile.TemporaryDirectory(prefix="gorder_cold_") as tmp:
    path = os.path.join(tmp, "t.xtc")
    xtc.write_trajectory(path, system.frames(unique, seed=3), system.box9(unique), precision=1000.0)
    part = os.path.join(tmp, "p.xtc")
    HipEngine(system.tables).run_trajectory([path], threads=cores)          # page cache, kernels, the runtime itself
    for n in sizes:
        files = [path] * (n // unique)
        if n % unique:
            xtc.write_trajectory(part, system.frames(n % unique, seed=4), system.box9(n % unique), precision=1000.0)
            files.append(part)
        row = {"frames": n}
        for route, dev in (("host_decode", False), ("device_decode", True)):
            best = None
            for _ in range(3):
                eng = HipEngine(system.tables)                                # a fresh handle: nothing cached
                st = eng.run_trajectory(files, threads=cores, device_decode=dev)
                eng.finish()
                eng.close()
                assert st["n_frames"] == n
                best = st if best is None or st["seconds_total"] < best["seconds_total"] else best
            row[route] = {"seconds": round(best["seconds_total"], 4), "setup": round(best["seconds_setup"], 4),
                          "frames_per_s": round(n / best["seconds_total"]), "batch_frames": best["batch_frames"]}
        print(json.dumps(row))
