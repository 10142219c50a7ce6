#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dec_gemm or decode or beam or gate or attention" > gpurun_out/r4e_kernel_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4e_kernel_tests.log
timeout 300 python tools/decode_kernels_bench.py > gpurun_out/r4e_decode_kernels_bench.txt 2>&1
timeout 1500 python -m pytest tests/test_generation_gpu.py tests/test_modules_gpu.py tests/test_timed_path_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -s -k "generation or beam" --durations=8 > gpurun_out/r4e_gen_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4e_gen_tests.log
timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4e_gen_bench.json 2> gpurun_out/r4e_gen_bench.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4e_gen -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload generate --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4e_gen.log 2>&1)
f=$(find gpurun_out/r4e_gen -name "*kernel_stats.csv" | head -1); python tools/prof_top.py "$f" 0 24 > gpurun_out/r4e_gen_summary.txt; rm -rf gpurun_out/r4e_gen
timeout 300 python tools/debug_ddp_rs.py > gpurun_out/r4e_debug_ddp.txt 2>&1
tail -12 gpurun_out/r4e_kernel_tests.log; cat gpurun_out/r4e_decode_kernels_bench.txt | tail -4; tail -12 gpurun_out/r4e_gen_tests.log; cut -c1-300 gpurun_out/r4e_gen_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4e_gen_bench.json')); print(round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/step', d.get('decode_hbm_frac'))"
head -22 gpurun_out/r4e_gen_summary.txt; grep -v "^\[\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r4e_debug_ddp.txt | tail -8
