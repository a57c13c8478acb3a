#!/usr/bin/env python3
"""End to end from an XTC file in which the analysed atoms are a quarter of the system (a membrane in front of its
water): both decode routes of the trajectory driver, identical sums, frames/s."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gorder_amd import HipEngine, xtc  # noqa: E402


def main():
    n_unique = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    system, desc = bench.make_system("aa256")
    n_sel = system.n_atoms
    rng = np.random.default_rng(7)
    lip = system.frames(n_unique, seed=11)
    n_w = 3 * n_sel
    centres = rng.uniform(0.0, 9.0, size=(n_w // 3, 1, 3))
    w0 = (centres + rng.normal(0.0, 0.05, size=(n_w // 3, 3, 3))).reshape(-1, 3)
    water = (w0[None] + rng.normal(0.0, 0.03, size=(n_unique, n_w, 3))).astype(np.float32)
    xyz = np.concatenate([lip, water], axis=1)
    box = system.box9(n_unique)
    group = np.arange(n_sel, dtype=np.uint32)
    cores = bench.host_cores()
    with tempfile.TemporaryDirectory(prefix="gorder_probe_") as tmp:
        path = os.path.join(tmp, "t.xtc")
        xtc.write_trajectory(path, xyz, box, precision=1000.0)
        size = os.path.getsize(path)
        out = {"atoms_in_file": int(xyz.shape[1]), "atoms_analysed": int(n_sel), "compressed_bytes_per_frame": size / n_unique}
        eng = HipEngine(system.tables)
        sums = {}
        for route, dev in (("host_decode", False), ("device_decode", True)):
            eng.reset()
            eng.run_trajectory([path] * 2, group=group, threads=cores, device_decode=dev)
            eng.reset()
            st = eng.run_trajectory([path] * repeats, group=group, threads=cores, device_decode=dev)
            sums[route] = eng.finish().sums
            out[route] = {"frames_per_s": st["n_frames"] / st["seconds_total"], "pcie_GBps": st["bytes_h2d"] / st["seconds_total"] / 1e9,
                          "decoded_on": "device" if st["device_decode"] else "host", "batch_frames": st["batch_frames"]}
        assert np.array_equal(sums["host_decode"], sums["device_decode"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
