#!/bin/bash
# usage (on the GPU box): tools/bench_all.sh <outfile>  — one bench.py line per workload: BASELINE.json's configs, then the other options
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/bench_all.jsonl}
: > $OUT
run() { python3 $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-scaling-reference --workload $1 --frames $2 --steps $3 --warmup $4 $5 $6 | grep '^{' >> $OUT; }
# warm-up launches cover >= 20 ms: the chip needs that long to settle its clocks
run aa256 10000 100 40
run aa256 10000 100 40 --trig acos
run aa256-leaflets 4000 50 40
GORDER_HIP_NO_SPECULATE=1 run aa256-leaflets 4000 50 40          # the two-kernel path (the frame read twice)
run cg3k-leaflets 4000 50 40
run aa256-leaflets-timewise 3000 50 40
GORDER_HIP_NO_SPECULATE=1 run aa256-leaflets-timewise 3000 50 40
run aa256-maps 3000 50 40
run cg3k 4000 100 60
run cg3k-local 512 10 4
run cg3k-local 10000 40 5             # 20 slabs a submit: the frame whose distances are exported is one in 10 000
run ua256 3000 50 40
run ua256-maps 3000 50 30
run ua256-fast 3000 50 40
run ua256-maps-fast 3000 50 30
run cg1m 500 40 30
# the options no BASELINE config uses (timewise rows)
run aa256-timewise 3000 50 40
run aa256-maps-timewise 3000 50 40
run ua256-timewise 3000 50 40
run aa256-cylinder 3000 50 40
run cg3k-dynamic 512 10 4
# fallback kernels real inputs can reach (round-3 review): membranes beyond 65 536 atoms, index-list membrane groups
run cg1m-local-r2.5 64 3 2
run aa256-leaflets-subset 4000 50 40
python3 - <<PY
import json
for line in open("$OUT"):
    d = json.loads(line)
    r = d["roofline"]
    print(f'{d["config"]["workload"][:70]:70s} {d["value"]/1e6:9.3f} Mframes/s  step {r["whole_step_ms"]:8.3f} ms = {r["whole_step_frac"]*100:5.1f}% HBM; longest: {r["kernel"]} {r["avg_launch_ms"]:8.3f} ms {r["frac"]*100:5.1f}%')
PY
