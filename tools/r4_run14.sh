#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
bash tools/r4_attn_ab.sh cross_text > gpurun_out/r4m_attn_ab.txt 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/r4m_attn_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4m_attn_tests.log
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
python bench.py $F > gpurun_out/r4m_bench_1.json 2> gpurun_out/r4m_bench_1.err
MMSUM_LIB=tools/build/base/libmmsum_hip.so python bench.py $F > gpurun_out/r4m_bench_base_1.json 2> gpurun_out/r4m_bench_base_1.err
python bench.py $F > gpurun_out/r4m_bench_2.json 2> gpurun_out/r4m_bench_2.err
MMSUM_LIB=tools/build/base/libmmsum_hip.so python bench.py $F > gpurun_out/r4m_bench_base_2.json 2> gpurun_out/r4m_bench_base_2.err
python bench.py --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-probe --no-also > gpurun_out/r4m_bench_b8.json 2> gpurun_out/r4m_bench_b8.err
cat gpurun_out/r4m_attn_ab.txt; tail -3 gpurun_out/r4m_attn_tests.log
for f in gpurun_out/r4m_bench_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2), d.get('peak_hbm_gb'))")"; done
