#!/bin/bash
# usage (on the GPU box): tools/kernel_split.sh <outdir-name> <workload> <frames> [reps]
# rocprofv3 kernel trace of quick_bench.py -> per-kernel totals (all launches incl. warm-up), gpurun_out/<name>/split_<workload>.txt
# The file starts with what is needed to recompute every number from it alone: workload, frames per launch (= per
# submit), algorithmic bytes per frame; each kernel line carries calls, average duration and — for kernels launched
# once per submit over the whole batch — algorithmic bytes / average duration as a fraction of the 8 TB/s HBM peak.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
QB_WARM=${QB_WARM:-20} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$2 -- python3 $ROOT/tools/quick_bench.py $2 $3 ${4:-5} > $OUT/kt_$2.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, sys
sys.path.insert(0, "$ROOT")
import bench
system, desc = bench.make_system("$2")
frames, reps, warm = int("$3"), int("${4:-5}"), int("${QB_WARM:-20}")
submits = warm + reps
p = glob.glob("$OUT/kt_$2/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
bpf = system.bytes_per_frame
with open("$OUT/split_$2.txt", "w") as f:
    head = [f"# workload $2: {desc}",
            f"# frames per launch (one submit): {frames}; submits traced: {submits} ({warm} warm-up + {reps} timed)",
            f"# algorithmic bytes per frame: {bpf} (12 N + 12, SURVEY 8d); per submit: {bpf * frames}",
            f"# all kernels of one submit: {tot / submits / 1e3:.1f} us -> {frames / (tot / submits) * 1e3:.3f} M frames/s, "
            f"{bpf * frames / (tot / submits) / 8000.0 * 100:.1f} % of the single-read HBM roofline (8 TB/s)",
            "# frac = algorithmic bytes of one submit / (calls per submit x average duration) / 8 TB/s"]
    for line in head:
        print(line); f.write(line + "\n")
    for r in rows:
        share = float(r["TotalDurationNs"]) / tot
        if share > 0.002:
            calls, avg = int(r["Calls"]), float(r["AverageNs"])
            per_submit = calls / submits
            frac = bpf * frames / (per_submit * avg) / 8000.0 if per_submit > 0 else 0.0
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:84]
            line = f'{name:84s} calls {calls:5d} ({per_submit:6.2f}/submit)  avg {avg/1e3:9.1f} us  {share*100:5.1f} %  frac {frac*100:5.1f} %'
            print(line); f.write(line + "\n")
PY
