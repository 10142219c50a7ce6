#!/usr/bin/env bash
# HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) against the default, interleaved on one box: decode step, B = 1 / B = 8 / B = 128 steps.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { python bench.py "$@" --no-cpu-baseline --no-kernel-probe --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.2f %s  %s' % (d['value'], d['unit'], d.get('ms_per_decode_step', d.get('ms_per_step'))))"; }
for rep in 1 2; do
  for k in 1 0; do
    export HIP_FORCE_DEV_KERNARG=$k
    echo "kernarg=$k generate  $(run --workload generate --steps 3 --warmup 1)"
    echo "kernarg=$k B=1       $(run --batch 1 --steps 40 --warmup 5)"
    echo "kernarg=$k B=8       $(run --batch 8 --steps 30 --warmup 5)"
  done
done
for k in 1 0; do export HIP_FORCE_DEV_KERNARG=$k; echo "kernarg=$k B=128     $(run --steps 6 --warmup 2)"; done
