#!/usr/bin/env python3
"""bench.py's end_to_end.large block alone:  python tools/e2e_large_probe.py [GB]   (env GORDER_XTC_NO_POPULATE=1 for the A/B)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
import __graft_entry__ as g   # noqa: E402

g.build()
if len(sys.argv) > 1:
    os.environ["GORDER_BENCH_LARGE_GB"] = sys.argv[1]
system, _ = bench.make_system("aa256")
out = bench.end_to_end_large(system, "aa256", 0, 0.0)
print(json.dumps({k: out.get(k) for k in ("frames", "file_GB", "seconds_to_write", "cold", "warm", "skipped")}, indent=1))
