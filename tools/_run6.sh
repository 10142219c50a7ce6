cd $GRAFT_REPO_ROOT
timeout 600 python tools/w4_stamps.py run > gpurun_out/r03f_stamps.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03f_stamps.txt
bash tools/gpu_round.sh r03f bench > /dev/null 2>&1
cut -c1-1500 gpurun_out/r03f_bench.json; tail -4 gpurun_out/r03f_bench.err
bash tools/gpu_round.sh r03f testsall > /dev/null 2>&1
tail -15 gpurun_out/r03f_tests.log
