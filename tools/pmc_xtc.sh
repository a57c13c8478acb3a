#!/bin/bash
# usage (on the GPU box): tools/pmc_xtc.sh <outdir-name> <kernel-substring> [windows]  -> counters of the XTC decode kernels (tools/xtc_decode_bench.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/x$i -- python3 $ROOT/tools/xtc_decode_bench.py aa256 256 ${3:-3566} > $OUT/x$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("$OUT/x*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(p)):
        if '$2' in r['Kernel_Name']:
            k=r['Counter_Name']; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items(): print(f"{k:28s} dispatches={n:3d} mean={v/n:.6g}")
PY
