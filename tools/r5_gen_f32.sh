#!/usr/bin/env bash
# f32 decode path: kernel tests, generation tests, then the f32 / bf16 generation timings (bench.py --workload generate)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "skinny or decode" > gpurun_out/gf_ktests.log 2>&1; echo "rc $?" >> gpurun_out/gf_ktests.log; tail -3 gpurun_out/gf_ktests.log | cut -c1-200
timeout 2400 python -m pytest tests/test_generation_gpu.py -m gpu -q -x -s > gpurun_out/gf_gtests.log 2>&1; echo "rc $?" >> gpurun_out/gf_gtests.log; grep -E "config 5|generation [0-9]|passed|failed|rc |Error|assert" gpurun_out/gf_gtests.log | cut -c1-330 | tail -14
for dt in f32 bf16; do
  python bench.py --workload generate --dtype $dt --steps 2 --warmup 1 > gpurun_out/gf_gen_$dt.json 2> gpurun_out/gf_gen_$dt.err
  python -c "import json; d=json.load(open('gpurun_out/gf_gen_$dt.json')); print('$dt', round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/decode step', round(d['decode_hbm_frac'],3))"
done
