#!/usr/bin/env bash
# BatchNorm kernels A/B on one box: tests, tools/bn_bench.py and the step, tree's library against tools/build/base/libmmsum_hip.so
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q -x -k "bn or resnet or image or img or f3 or handoff" > gpurun_out/bn_tests.log 2>&1; echo "rc $?" >> gpurun_out/bn_tests.log; tail -3 gpurun_out/bn_tests.log | cut -c1-200
echo "== new"; python tools/bn_bench.py 2>&1 | grep "^R="; echo "== base"; MMSUM_LIB=tools/build/base/libmmsum_hip.so python tools/bn_bench.py 2>&1 | grep "^R="
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
  python bench.py $F > gpurun_out/bn_new_$rep.json 2> gpurun_out/bn_new_$rep.err
  MMSUM_LIB=tools/build/base/libmmsum_hip.so python bench.py $F > gpurun_out/bn_base_$rep.json 2> gpurun_out/bn_base_$rep.err
done
for f in gpurun_out/bn_new_*.json gpurun_out/bn_base_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2))")"; done
