#!/usr/bin/env bash
# What of the image branch is exposed in wall time: default vs MMSUM_SIDE_STREAM=0 vs the ResNet stubbed, one box.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
  python bench.py $F > gpurun_out/exp_default_$rep.json 2> gpurun_out/exp_default_$rep.err
  MMSUM_SIDE_STREAM=0 python bench.py $F > gpurun_out/exp_noside_$rep.json 2> gpurun_out/exp_noside_$rep.err
  python bench.py $F --diag-stub-resnet > gpurun_out/exp_stub_$rep.json 2> gpurun_out/exp_stub_$rep.err
done
for f in gpurun_out/exp_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2))")"; done
