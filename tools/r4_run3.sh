#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/debug_rs.py > gpurun_out/r4c_debug_rs.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "dec_gemm or decode_cross_attn" > gpurun_out/r4c_dec_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4c_dec_tests.log
timeout 1500 python -m pytest tests/test_generation_gpu.py tests/test_modules_gpu.py tests/test_timed_path_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -s -k "generation or beam" --durations=8 > gpurun_out/r4c_gen_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4c_gen_tests.log
timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4c_gen_bench.json 2> gpurun_out/r4c_gen_bench.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4c_gen -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload generate --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4c_gen.log 2>&1)
f=$(find gpurun_out/r4c_gen -name "*kernel_stats.csv" | head -1); python tools/prof_top.py "$f" 0 30 > gpurun_out/r4c_gen_summary.txt; rm -rf gpurun_out/r4c_gen
cat gpurun_out/r4c_debug_rs.txt | tail -20; tail -15 gpurun_out/r4c_dec_tests.log; tail -25 gpurun_out/r4c_gen_tests.log; cat gpurun_out/r4c_gen_bench.json | cut -c1-900; head -25 gpurun_out/r4c_gen_summary.txt
