#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_ddp_rccl_gpu.py -m gpu -q -x > gpurun_out/r4n_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4n_tests.log
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
python bench.py $F > gpurun_out/r4n_bench_1.json 2> gpurun_out/r4n_bench_1.err
python bench.py $F > gpurun_out/r4n_bench_2.json 2> gpurun_out/r4n_bench_2.err
python bench.py --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-probe --no-also > gpurun_out/r4n_bench_b8.json 2> gpurun_out/r4n_bench_b8.err
tail -3 gpurun_out/r4n_tests.log
for f in gpurun_out/r4n_bench_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2), d.get('peak_hbm_gb'))")"; done
