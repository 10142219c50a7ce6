#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/debug_rs2.py > gpurun_out/r4f_debug_rs2.txt 2>&1
timeout 2700 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r4f_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4f_tests.log
timeout 900 python bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err; echo "bench rc $?" >> gpurun_out/r4f_bench.err
grep -v "^\[\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r4f_debug_rs2.txt | tail -6; tail -30 gpurun_out/r4f_tests.log; tail -2 gpurun_out/r4f_bench.err; cut -c1-300 gpurun_out/r4f_bench.json
