cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py::test_tn_w4_weight_gradients_at_bench_sizes tests/test_bench_shapes_gpu.py::test_live_row_counts_at_bench_sizes tests/test_bench_shapes_gpu.py::test_wide_step_f32_and_bf16_vs_oracle tests/test_bench_shapes_gpu.py::test_nt_ring_256_persistent_all_epilogues tests/test_timed_path_gpu.py tests/test_modules_gpu.py -q > gpurun_out/r03d_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03d_tests.log
tail -8 gpurun_out/r03d_tests.log
for rep in 1 2; do
  (cd tools/build/r02tree && timeout 600 python bench.py --no-cpu-baseline --no-kernel-probe) > gpurun_out/r03d_ab_old$rep.json 2> gpurun_out/r03d_ab_old$rep.err
  timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03d_ab_new$rep.json 2> gpurun_out/r03d_ab_new$rep.err
done
timeout 900 python bench.py --batch 112 --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03d_b112.json 2> gpurun_out/r03d_b112.err
for f in gpurun_out/r03d_ab_old1.json gpurun_out/r03d_ab_new1.json gpurun_out/r03d_ab_old2.json gpurun_out/r03d_ab_new2.json gpurun_out/r03d_b112.json; do echo $f; cut -c1-260 $f; done
tail -2 gpurun_out/r03d_b112.err
bash tools/gpu_round.sh r03d prof > /dev/null 2>&1
head -30 gpurun_out/r03d_prof_summary.txt
