cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
bash tools/gpu_round.sh r03l testsall > /dev/null 2>&1
tail -12 gpurun_out/r03l_tests.log | cut -c1-300
bash tools/gpu_round.sh r03l bench prof > /dev/null 2>&1
cut -c1-300 gpurun_out/r03l_bench.json; tail -2 gpurun_out/r03l_bench.err; head -12 gpurun_out/r03l_prof_summary.txt
