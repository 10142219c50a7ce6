#!/usr/bin/env bash
# generation: decode tests, then bench.py --workload generate interleaved against tools/build/base/libmmsum_hip.so on one box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "decode" > gpurun_out/gen_ab_tests.log 2>&1; echo "rc $?" >> gpurun_out/gen_ab_tests.log
tail -3 gpurun_out/gen_ab_tests.log
for rep in 1 2 3; do
  timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/gen_new_$rep.json 2> gpurun_out/gen_new_$rep.err
  MMSUM_LIB=tools/build/base/libmmsum_hip.so timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/gen_base_$rep.json 2> gpurun_out/gen_base_$rep.err
done
for f in gpurun_out/gen_new_*.json gpurun_out/gen_base_*.json; do echo "$f $(python -c "
import json; d=json.load(open('$f')); print(round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/step')")"; done
