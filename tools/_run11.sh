cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_epi_bench.py 129024 > gpurun_out/r03k_epi.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03k_epi.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -x -k "gemm or epilogue or shapes or nt_" > gpurun_out/r03k_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03k_tests.log
tail -3 gpurun_out/r03k_tests.log
# power / clock samples while the step runs (ordinary user: read-only queries)
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 1; done ) > gpurun_out/r03k_smi.txt 2>&1 &
SMI=$!
timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe --steps 40 --warmup 5 > gpurun_out/r03k_bench40.json 2> gpurun_out/r03k_bench40.err
kill $SMI 2>/dev/null
cut -c1-200 gpurun_out/r03k_bench40.json
head -40 gpurun_out/r03k_smi.txt | cut -c1-300
