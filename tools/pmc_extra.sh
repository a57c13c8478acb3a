#!/bin/bash
# usage (on the GPU box): tools/pmc_extra.sh <outdir-name> <workload> <frames> [kernel-substring]
# wait and instruction-fetch counters of one kernel (separate --pmc passes, kernel trace only): where a kernel that is
# not VALU-bound waits.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"; do
  # (a pass with TA_* / TCP_*_sum counters made rocprofv3 abort and the run hang on this pool in round 4: not collected)
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/quick_bench.py $2 $3 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(p)):
        if '${4:-k_ua}' in r['Kernel_Name']:
            k=r['Counter_Name']; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items(): print(f"{k:40s} dispatches={n:3d} mean={v/n:.6g}")
PY
