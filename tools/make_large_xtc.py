#!/usr/bin/env python3
"""Write one part of a large synthetic XTC trajectory of DISTINCT frames (bench.py's end_to_end.large; CPU only).

    python tools/make_large_xtc.py <workload> <out.xtc> <first_frame> <n_frames>

Frame k of the trajectory is synthetic frame (seed 777, k) of the workload — every frame its own random displacement of
every atom, so no two compressed frames share bytes — with time 10 k ps; parts written by several of these processes
concatenate into one trajectory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workload, out, first, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import bench
    from gorder_amd import xtc
    system, _ = bench.make_system(workload)
    lib = xtc._lib()
    import ctypes as C
    w = C.c_void_p()
    if lib.gorder_xtc_writer_open(out.encode(), system.n_atoms, 1000.0, C.byref(w)) != 0:
        raise SystemExit(f"cannot create {out}")
    box = np.ascontiguousarray(system.box9(1).reshape(9), dtype=np.float32)
    try:
        chunk = 128
        for a in range(first, first + n, chunk):
            m = min(chunk, first + n - a)
            xyz = system.frames(m, seed=777, first=a)
            for k in range(m):
                if lib.gorder_xtc_writer_add(w, xyz[k].ctypes.data, box.ctypes.data, a + k, float(10.0 * (a + k))) != 0:
                    raise SystemExit(f"{out}: write error in frame {a + k}")
    finally:
        lib.gorder_xtc_writer_close(w)


if __name__ == "__main__":
    main()
