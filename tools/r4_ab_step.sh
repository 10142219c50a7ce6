#!/usr/bin/env bash
# usage: r4_ab_step.sh "<pytest -k expr>": kernel tests, then the step interleaved against tools/build/base/libmmsum_hip.so on one box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q -x -k "$1" > gpurun_out/ab_tests.log 2>&1; echo "rc $?" >> gpurun_out/ab_tests.log
tail -6 gpurun_out/ab_tests.log | cut -c1-250
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
python bench.py $F > gpurun_out/ab_new_$rep.json 2> gpurun_out/ab_new_$rep.err
MMSUM_LIB=tools/build/base/libmmsum_hip.so python bench.py $F > gpurun_out/ab_base_$rep.json 2> gpurun_out/ab_base_$rep.err
done
for f in gpurun_out/ab_new_*.json gpurun_out/ab_base_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2))")"; done
