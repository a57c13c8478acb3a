#!/bin/bash
# usage (on the GPU box): tools/kernel_split.sh <outdir-name> <workload> <frames> [reps]
# rocprofv3 kernel trace of quick_bench.py -> per-kernel totals (all launches incl. warm-up), gpurun_out/<name>/split_<workload>.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
QB_WARM=${QB_WARM:-20} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$2 -- python3 $ROOT/tools/quick_bench.py $2 $3 ${4:-5} > $OUT/kt_$2.log 2>&1
python3 - <<PY
import csv, glob
p = glob.glob("$OUT/kt_$2/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/split_$2.txt", "w") as f:
    for r in rows:
        if float(r["TotalDurationNs"]) / tot > 0.002:
            line = f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:10.1f} us  {float(r["TotalDurationNs"])/tot*100:5.1f} %'
            print(line); f.write(line + "\n")
PY
