#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh <outdir-name> <workload> <frames> <kernel-substring> "<counters of pass 1>" ["<pass 2>" ...]
# Any set of counters, one rocprofv3 --pmc pass per argument (kernel trace only), summed per kernel name.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
W=$2; F=$3; K=$4; shift 4
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/quick_bench.py $W $F 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(p)):
        if '$K' in r['Kernel_Name']:
            k=r['Counter_Name']; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items(): print(f"{k:40s} dispatches={n:3d} mean={v/n:.6g}")
PY
