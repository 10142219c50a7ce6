#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "attention" > gpurun_out/r4b_attn_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4b_attn_tests.log
ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1 ATTN_BENCH_B=128 timeout 300 python tools/attn_bench.py > gpurun_out/r4b_attn_bench.txt 2>&1
timeout 1500 python -m pytest tests/test_generation_gpu.py tests/test_ddp_rccl_gpu.py -m gpu -q -s --durations=10 --deselect tests/test_generation_gpu.py::test_generation_f32_max_length_128_at_bart_large_width > gpurun_out/r4b_gen_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4b_gen_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-also > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; echo "rc $?" >> gpurun_out/r4b_bench.err
tail -30 gpurun_out/r4b_attn_tests.log; cat gpurun_out/r4b_attn_bench.txt; tail -30 gpurun_out/r4b_gen_tests.log; tail -3 gpurun_out/r4b_bench.err; cut -c1-400 gpurun_out/r4b_bench.json
