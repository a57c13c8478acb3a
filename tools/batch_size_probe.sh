#!/bin/bash
# usage (GPU box): tools/batch_size_probe.sh <workload> <frames...>  — throughput of one workload against the batch size
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
for f in "$@"; do
  python3 $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-scaling-reference --workload $W --frames $f --steps 100 --warmup 40 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames %6d  %8.3f Mframes/s  %8.3f ms/step' % ($f, d['value']/1e6, d['ms_per_step']))"
done
