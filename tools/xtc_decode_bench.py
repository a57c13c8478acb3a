#!/usr/bin/env python3
"""Throughput of the device XTC decoder (k_xtc_scan + k_xtc_chunks: frames decompressed on the device) against the window size,
next to the host decoder on all cores.

    python tools/xtc_decode_bench.py [workload] [unique_frames] [window,window,...]

The repo's encoder writes `unique_frames` synthetic frames of the workload (precision 1000); the packed window is
repeated on the device to the sizes measured (the frame table is tiled, the blob is shared)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gorder_amd import HipEngine, xtc  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "aa256"
    n_unique = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    system, desc = bench.make_system(name)
    xyz, box = system.frames(n_unique, seed=99), system.box9(n_unique)
    with tempfile.TemporaryDirectory(prefix="gorder_xtc_") as tmp:
        path = os.path.join(tmp, "t.xtc")
        xtc.write_trajectory(path, xyz, box, precision=1000.0)
        size = os.path.getsize(path)
        cores = bench.host_cores()
        xtc.read_trajectory([path], chunk=n_unique, threads=cores)
        t0 = time.perf_counter()
        for _ in range(3):
            host = xtc.read_trajectory([path], chunk=n_unique, threads=cores)[0]
        host_fps = 3 * n_unique / (time.perf_counter() - t0)
        w = xtc.pack_trajectory([path], chunk=n_unique, threads=cores)[0]
    n_atoms = w["n_atoms_file"]
    torch.cuda.set_stream(torch.cuda.Stream())      # the handle shares torch's CURRENT stream (not the null stream)
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    blob = torch.from_numpy(w["blob"]).cuda()
    out = {"workload": f"{name}: {desc}", "atoms_per_frame": n_atoms, "compressed_bytes_per_frame": size / n_unique,
           "host_decoder": {"frames_per_s": host_fps, "threads": cores}, "device": []}
    windows = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 1024, 4096, 16384]
    for window in windows:
        reps = (window + n_unique - 1) // n_unique
        table = np.tile(w["frames"], reps)[:window]
        frames = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
        dst = torch.empty((window, n_atoms, 3), dtype=torch.float32, device="cuda")

        def run():
            eng.xtc_decode(blob.data_ptr(), blob.numel(), frames.data_ptr(), window, n_atoms, 0, n_atoms, dst.data_ptr(), n_atoms)

        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        got = dst[:n_unique].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), host.view(np.uint32)), "device decode differs from the host decoder"
        out["device"].append({"window_frames": window, "ms": ms, "frames_per_s": window / ms * 1e3,
                              "atoms_per_s": window * n_atoms / ms * 1e3, "ns_per_atom_and_wave": ms * 1e6 / n_atoms,
                              "out_GBps": window * n_atoms * 12 / ms / 1e6})
        del dst, frames
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
