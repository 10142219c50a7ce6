cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py::test_nt_ring_256_persistent_all_epilogues tests/test_bench_shapes_gpu.py::test_nt_ring_a2_split_and_lm_head tests/test_bench_shapes_gpu.py::test_tn_w4_weight_gradients_at_bench_sizes tests/test_bench_shapes_gpu.py::test_live_row_counts_at_bench_sizes tests/test_timed_path_gpu.py tests/test_ddp_rccl_gpu.py -q -x > gpurun_out/r03b_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03b_tests.log
tail -15 gpurun_out/r03b_tests.log
bash tools/gpu_ab.sh r03b > /dev/null 2>&1
bash tools/gpu_round.sh r03b bench prof
