set -x
mkdir -p gpurun_out/r3z
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z/kt_head -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-scaling-reference > $GRAFT_REPO_ROOT/gpurun_out/r3z/bench_headline_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/kt_head.err
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3z/bench_aa256.json 2> gpurun_out/r3z/bench_aa256.err
tools/bench_all.sh gpurun_out/r3z/bench_all.jsonl > gpurun_out/r3z/bench_all.txt 2>&1
tools/pmc.sh r3z_pmc_aa aa256 10000 k_bonds_tiled > gpurun_out/r3z/pmc_aa.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/r3z_pmc_aa k_bonds_tiled aa256 10000 > gpurun_out/r3z/pmc_aa256_k_bonds_tiled.json
tools/pmc.sh r3z_pmc_loc cg3k-local 512 k_local_flags_rows > gpurun_out/r3z/pmc_loc.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/r3z_pmc_loc k_local_flags_rows cg3k-local 512 > gpurun_out/r3z/pmc_cg3k-local_k_local_flags_rows.json
for w in cg3k-local ua256 ua256-maps aa256-maps aa256-leaflets; do tools/kernel_split.sh r3z $w $( [ $w = cg3k-local ] && echo 512 || echo 3000 ) > gpurun_out/r3z/split_$w.log 2>&1; done
python3 tools/xtc_decode_bench.py aa256 256 256,1024,3566,16384 > gpurun_out/r3z/xtc_decode_bench.json 2> gpurun_out/r3z/xtc_decode_bench.err
cat gpurun_out/r3z/bench_all.txt | tail -10
