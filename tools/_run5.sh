cd $GRAFT_REPO_ROOT
: > gpurun_out/r03e_stagger.txt
for us in 0 4 8 12 16 0 8; do
  echo "== MMSUM_W4_STAGGER_US=$us" >> gpurun_out/r03e_stagger.txt
  MMSUM_W4_STAGGER_US=$us timeout 300 python tools/gemm_epi_bench.py 129024 >> gpurun_out/r03e_stagger.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r03e_stagger.txt
timeout 2400 python -m pytest tests/test_bench_shapes_gpu.py tests/test_timed_path_gpu.py tests/test_abi_cpu.py -q > gpurun_out/r03e_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03e_tests.log
tail -8 gpurun_out/r03e_tests.log
bash tools/gpu_round.sh r03e bench prof > /dev/null 2>&1
cut -c1-400 gpurun_out/r03e_bench.json; tail -2 gpurun_out/r03e_bench.err
head -25 gpurun_out/r03e_prof_summary.txt
bash tools/pmc_dominant.sh > gpurun_out/r03e_pmc_summary.txt 2>&1
cat gpurun_out/r03e_pmc_summary.txt
