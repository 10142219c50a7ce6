#!/usr/bin/env bash
# B = 8 / B = 1 step with and without the side stream (fork / join edges in the captured graphs), interleaved on one box.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do
  for b in 8 1; do
    for side in 1 0; do
      v=$(MMSUM_SIDE_STREAM=$side timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-probe --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.1f businesses/s  %.2f ms/step  p50 %.2f' % (d['value'], d['ms_per_step'], d['ms_per_step_p50']))")
      echo "B=$b side=$side  $v"
    done
  done
done
