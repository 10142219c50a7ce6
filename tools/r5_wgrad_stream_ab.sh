#!/usr/bin/env bash
# Weight gradients on a side stream for small batches (engine.wgrad): module / graph tests, then the B = 8 and B = 1 steps with it on and off, one box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_modules_gpu.py tests/test_timed_path_gpu.py tests/test_ddp_rccl_gpu.py -m gpu -q -x > gpurun_out/ws_tests.log 2>&1; echo "rc $?" >> gpurun_out/ws_tests.log; tail -4 gpurun_out/ws_tests.log | cut -c1-200
for B in 8 1; do
  F="--batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-probe --no-also"
  for rep in 1 2; do
    python bench.py $F > gpurun_out/ws_on_${B}_$rep.json 2> gpurun_out/ws_on_${B}_$rep.err
    MMSUM_WGRAD_STREAM=0 python bench.py $F > gpurun_out/ws_off_${B}_$rep.json 2> gpurun_out/ws_off_${B}_$rep.err
    echo "B=$B rep $rep: on $(python -c "import json; d=json.load(open('gpurun_out/ws_on_${B}_$rep.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['final_loss'],5))")   off $(python -c "import json; d=json.load(open('gpurun_out/ws_off_${B}_$rep.json')); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['final_loss'],5))")"
  done
done
