#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outdir-name> <workload> <frames> [kernel-substring]  -> gpurun_out/<name>/pmc_*.csv
# PMC passes are separate runs, each with --kernel-trace only (never combined with other trace domains).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/quick_bench.py $2 $3 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(p)):
        if '${4:-k_bonds}' in r['Kernel_Name']:
            k=r['Counter_Name']; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items(): print(f"{k:28s} dispatches={n:3d} mean={v/n:.6g}")
PY
