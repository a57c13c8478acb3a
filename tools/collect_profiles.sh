#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/...  (copy what is to be judged into profiles/)
# headline under rocprofv3 (kernel trace + stats), the default bench line, every workload, PMC passes, kernel splits
TAG=${1:-r4z}
set -x
mkdir -p gpurun_out/$TAG
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/kt_head -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-scaling-reference > $GRAFT_REPO_ROOT/gpurun_out/$TAG/bench_headline_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG/kt_head.err
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_aa256.json 2> gpurun_out/$TAG/bench_aa256.err
tools/bench_all.sh gpurun_out/$TAG/bench_all.jsonl > gpurun_out/$TAG/bench_all.txt 2>&1
tools/pmc.sh ${TAG}_pmc_aa aa256 10000 k_bonds_tiled > gpurun_out/$TAG/pmc_aa.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_aa k_bonds_tiled aa256 10000 > gpurun_out/$TAG/pmc_aa256_k_bonds_tiled.json
tools/pmc.sh ${TAG}_pmc_ua ua256 3000 k_ua_extras > gpurun_out/$TAG/pmc_ua.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_ua k_ua_extras ua256 3000 > gpurun_out/$TAG/pmc_ua256_k_ua_extras.json
tools/pmc.sh ${TAG}_pmc_uaf ua256-fast 3000 k_ua_extras > gpurun_out/$TAG/pmc_uaf.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_uaf k_ua_extras ua256-fast 3000 > gpurun_out/$TAG/pmc_ua256-fast_k_ua_extras_fast.json
for w in cg3k-local ua256 ua256-fast ua256-maps ua256-maps-fast aa256-maps aa256-leaflets; do tools/kernel_split.sh $TAG $w $( [ $w = cg3k-local ] && echo 512 || echo 3000 ) > gpurun_out/$TAG/split_$w.log 2>&1; done
cat gpurun_out/$TAG/bench_all.txt | tail -20
