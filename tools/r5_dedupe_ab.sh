#!/usr/bin/env bash
# Live-image window (one representative of the empty image slots) on vs off, one box: the image-branch tests, then the step interleaved.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py tests/test_timed_path_gpu.py -m gpu -q -x -k "${1:-image or resnet or img or f3 or bn or conv or F8 or f8}" > gpurun_out/dd_tests.log 2>&1; echo "rc $?" >> gpurun_out/dd_tests.log
tail -15 gpurun_out/dd_tests.log | cut -c1-300
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
  python bench.py $F > gpurun_out/dd_on_$rep.json 2> gpurun_out/dd_on_$rep.err
  MMSUM_IMAGE_DEDUPE=0 python bench.py $F > gpurun_out/dd_off_$rep.json 2> gpurun_out/dd_off_$rep.err
done
for f in gpurun_out/dd_on_*.json gpurun_out/dd_off_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2), d.get('ms_per_step_p50'))")"; done
tail -3 gpurun_out/dd_on_1.err
