#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/r4r_attn_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4r_attn_tests.log
tail -5 gpurun_out/r4r_attn_tests.log
bash tools/r4_attn_ab.sh cross_text > gpurun_out/r4r_attn_ab.txt 2>&1
ATTN_BENCH_PADS=0 bash tools/r4_attn_ab.sh cross_img4 > gpurun_out/r4r_attn_ab_img.txt 2>&1
bash tools/r4_attn_ab.sh cross_table > gpurun_out/r4r_attn_ab_table.txt 2>&1
cat gpurun_out/r4r_attn_ab.txt gpurun_out/r4r_attn_ab_img.txt gpurun_out/r4r_attn_ab_table.txt
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
python bench.py $F > gpurun_out/r4r_bench_new_$rep.json 2> gpurun_out/r4r_bench_new_$rep.err
MMSUM_LIB=tools/build/base/libmmsum_hip.so python bench.py $F > gpurun_out/r4r_bench_base_$rep.json 2> gpurun_out/r4r_bench_base_$rep.err
done
for f in gpurun_out/r4r_bench_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2))")"; done
