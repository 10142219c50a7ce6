cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_epi_bench.py 129024 > gpurun_out/r03i_epi.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03i_epi.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -x -k "gemm or gelu or epilogue or nt_ or shapes" > gpurun_out/r03i_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03i_tests.log
tail -4 gpurun_out/r03i_tests.log
bash tools/gpu_round.sh r03i bench prof > /dev/null 2>&1
cut -c1-700 gpurun_out/r03i_bench.json; tail -3 gpurun_out/r03i_bench.err; head -14 gpurun_out/r03i_prof_summary.txt
