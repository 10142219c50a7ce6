#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do
  timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4q_gen_iograph_$rep.json 2> gpurun_out/r4q_gen_iograph_$rep.err
  MMSUM_DECODE_IO_GRAPH=0 timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4q_gen_iohost_$rep.json 2> gpurun_out/r4q_gen_iohost_$rep.err
done
for f in gpurun_out/r4q_gen_*.json; do echo "$f $(python -c "
import json; d=json.load(open('$f')); print(round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/step')")"; done
