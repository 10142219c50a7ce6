cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r03u bench prof > /dev/null 2>&1
cut -c1-260 gpurun_out/r03u_bench.json; tail -2 gpurun_out/r03u_bench.err; head -30 gpurun_out/r03u_prof_summary.txt | cut -c1-150; tail -14 gpurun_out/r03u_prof_summary.txt
