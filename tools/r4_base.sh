cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1 ATTN_BENCH_B=128
timeout 300 python tools/attn_bench.py > gpurun_out/r4base_attn_bench.txt 2>&1
timeout 600 bash tools/prof_attn.sh > gpurun_out/r4base_prof_attn.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4base_gen -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload generate --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4base_gen.log 2>&1)
f=$(find gpurun_out/r4base_gen -name "*kernel_stats.csv" | head -1)
python tools/prof_top.py "$f" 0 40 > gpurun_out/r4base_gen_summary.txt
rm -rf gpurun_out/r4base_gen
cat gpurun_out/r4base_attn_bench.txt; tail -40 gpurun_out/r4base_prof_attn.txt; head -30 gpurun_out/r4base_gen_summary.txt; tail -3 gpurun_out/r4base_gen.log
