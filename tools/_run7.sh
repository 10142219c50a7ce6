cd $GRAFT_REPO_ROOT
timeout 600 python tools/w4_stamps.py run > gpurun_out/r03g_stamps.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03g_stamps.txt | cut -c1-420
timeout 300 python tools/gemm_epi_bench.py 129024 > gpurun_out/r03g_epi.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03g_epi.txt
timeout 1500 python -m pytest tests/test_timed_path_gpu.py tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -x -k "not generation" -s > gpurun_out/r03g_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03g_tests.log
grep -E "yardstick|hip .* yardstick|passed|failed|rc " gpurun_out/r03g_tests.log | cut -c1-400 | tail -30
