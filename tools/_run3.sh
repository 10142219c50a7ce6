cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r03c testsall > /dev/null 2>&1
tail -12 gpurun_out/r03c_tests.log
bash tools/gpu_round.sh r03c bench > /dev/null 2>&1
MMSUM_WGRAD_STREAM=1 timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03c_bench_ws.json 2> gpurun_out/r03c_bench_ws.err
timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03c_bench_nows.json 2> gpurun_out/r03c_bench_nows.err
MMSUM_WGRAD_STREAM=1 timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03c_bench_ws2.json 2>> gpurun_out/r03c_bench_ws.err
bash tools/gpu_round.sh r03c prof > /dev/null 2>&1
for f in gpurun_out/r03c_bench.json gpurun_out/r03c_bench_ws.json gpurun_out/r03c_bench_nows.json gpurun_out/r03c_bench_ws2.json; do echo $f; cut -c1-330 $f; done
tail -3 gpurun_out/r03c_bench_ws.err
head -40 gpurun_out/r03c_prof_summary.txt
