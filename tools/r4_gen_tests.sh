#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_generation_gpu.py tests/test_ddp_rccl_gpu.py -m gpu -q -x -s --durations=10 > gpurun_out/r4_gen_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4_gen_tests.log
timeout 600 python -m pytest tests/test_timed_path_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "generation" -s > gpurun_out/r4_gen_tests2.log 2>&1; echo "rc $?" >> gpurun_out/r4_gen_tests2.log
timeout 900 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "rc $?" >> gpurun_out/r4a_bench.err
tail -25 gpurun_out/r4_gen_tests.log; tail -8 gpurun_out/r4_gen_tests2.log; tail -3 gpurun_out/r4a_bench.err; cut -c1-1500 gpurun_out/r4a_bench.json
