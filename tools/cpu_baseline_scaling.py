#!/usr/bin/env python3
"""The CPU baseline's scaling table on this box (bench.py's cpu_baseline block, printed alone):
    python tools/cpu_baseline_scaling.py [workload] > profiles/rNN_cpu_baseline_scaling.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402

import __graft_entry__ as g   # noqa: E402
g.build()
system, name = bench.make_system(sys.argv[1] if len(sys.argv) > 1 else "aa256")
out = bench.cpu_baseline(system)
out["workload"] = name
out["nproc"] = os.cpu_count()
try:
    out["loadavg"] = os.getloadavg()
except OSError:
    pass
print(json.dumps(out, indent=1))
