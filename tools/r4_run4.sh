#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "dec_gemm or decode_cross_attn" > gpurun_out/r4d_dec_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4d_dec_tests.log
timeout 300 python tools/decode_kernels_bench.py > gpurun_out/r4d_decode_kernels_bench.txt 2>&1
export ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1 ATTN_BENCH_B=128
timeout 200 python tools/attn_bench.py cross_img4 > gpurun_out/r4d_attn_img4_new.txt 2>&1
MMSUM_LIB=$PWD/tools/build/nochunk/multimodalsum_amd/csrc/libmmsum_hip.so timeout 200 python tools/attn_bench.py cross_img4 > gpurun_out/r4d_attn_img4_old.txt 2>&1
timeout 200 python tools/attn_bench.py cross_img4 >> gpurun_out/r4d_attn_img4_new.txt 2>&1
MMSUM_LIB=$PWD/tools/build/nochunk/multimodalsum_amd/csrc/libmmsum_hip.so timeout 200 python tools/attn_bench.py cross_img4 >> gpurun_out/r4d_attn_img4_old.txt 2>&1
unset ATTN_BENCH_PADS ATTN_BENCH_MAPS ATTN_BENCH_B
timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4d_gen_bench.json 2> gpurun_out/r4d_gen_bench.err
MMSUM_DECODE_FAST=0 timeout 600 python bench.py --workload generate --steps 3 --warmup 2 > gpurun_out/r4d_gen_bench_oldpath.json 2> gpurun_out/r4d_gen_bench_oldpath.err
tail -12 gpurun_out/r4d_dec_tests.log; cat gpurun_out/r4d_decode_kernels_bench.txt; echo NEW; cat gpurun_out/r4d_attn_img4_new.txt; echo OLD; cat gpurun_out/r4d_attn_img4_old.txt
python -c "
import json
for f in ('r4d_gen_bench','r4d_gen_bench_oldpath'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],2), 'summaries/s', round(d['ms_per_decode_step'],3), 'ms/step')"
