cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r03p testsall > /dev/null 2>&1
tail -6 gpurun_out/r03p_tests.log | cut -c1-300
bash tools/gpu_round.sh r03p bench prof > /dev/null 2>&1
cut -c1-300 gpurun_out/r03p_bench.json; tail -2 gpurun_out/r03p_bench.err; head -8 gpurun_out/r03p_prof_summary.txt; grep embed_ln_bwd gpurun_out/r03p_prof_summary.txt | cut -c1-120
