#!/usr/bin/env bash
# Attention A/B on one box: the attention tests, then tools/attn_bench.py (bench batch, trailing pads, compact K/V) with the tree's library
# and with tools/build/base/libmmsum_hip.so, interleaved.  usage: r5_attn_ab.sh [cases...]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "attn or attention" > gpurun_out/aab_tests.log 2>&1; echo "rc $?" >> gpurun_out/aab_tests.log
tail -3 gpurun_out/aab_tests.log | cut -c1-200
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
for rep in 1 2; do
  for c in ${@:-cross_text cross_img4 cross_table self_causal}; do
    echo "new  $(python tools/attn_bench.py $c 2>&1 | grep "^$c")"
    echo "base $(MMSUM_LIB=tools/build/base/libmmsum_hip.so python tools/attn_bench.py $c 2>&1 | grep "^$c")"
  done
done
