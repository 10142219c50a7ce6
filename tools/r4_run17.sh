#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4p_gen_bench.json 2> gpurun_out/r4p_gen_bench.err
timeout 1500 python -m pytest tests/test_generation_gpu.py tests/test_modules_gpu.py tests/test_timed_path_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "generat or beam" > gpurun_out/r4p_gen_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4p_gen_tests.log
tail -4 gpurun_out/r4p_gen_tests.log
python -c "
import json; d=json.load(open('gpurun_out/r4p_gen_bench.json')); print('generate', round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/step')"
