#!/usr/bin/env python3
"""End-to-end rate of the trajectory driver alone (bench.py's end_to_end block, several repeats, both routes)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

system, desc = bench.make_system(sys.argv[1] if len(sys.argv) > 1 else "aa256")
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    e = bench.end_to_end(system, 0, repeats=repeats, **({"n_unique": int(sys.argv[4])} if len(sys.argv) > 4 else {}))
    print(json.dumps({"device_decode": {k: e[k] for k in ("value", "pcie_GBps", "seconds", "batch_frames")},
                      "host_decode": {k: e["host_decode"][k] for k in ("value", "pcie_GBps")}}))
