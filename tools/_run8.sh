cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_epi_bench.py 129024 > gpurun_out/r03h_epi.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03h_epi.txt
timeout 600 python tools/w4_stamps.py run 64512 1024 1024 > gpurun_out/r03h_stamps.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03h_stamps.txt | cut -c1-420
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py "tests/test_timed_path_gpu.py::test_text_table_step_bf16_config3" -q -x -k "not generation" -s > gpurun_out/r03h_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03h_tests.log
grep -E "yardstick|passed|failed|rc |Error" gpurun_out/r03h_tests.log | cut -c1-400 | tail -30
