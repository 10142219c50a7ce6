#!/usr/bin/env bash
# Attention A/B on one box: the attention tests, then tools/attn_bench.py (bench batch, trailing pads, compact K/V) with the tree's library
# and with tools/build/base/libmmsum_hip.so, interleaved.  Round 6: 'new' was a build with EXTRA="-DMMSUM_ATTN_W64=1 -mllvm -amdgpu-mfma-vgpr-form" (the
# one-wave-per-SIMD forward), 'base' the default build.  usage: r6_attn_ab.sh [cases...]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_long_sequences_gpu.py -m gpu -q -x -k "attn or attention or encoder" > gpurun_out/r6_aab_tests.log 2>&1; echo "rc $?" >> gpurun_out/r6_aab_tests.log
tail -4 gpurun_out/r6_aab_tests.log | cut -c1-300
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
for rep in 1 2; do
  for c in ${@:-cross_text}; do
    echo "new  $(python tools/attn_bench.py $c 2>&1 | grep "^$c")"
    echo "base $(MMSUM_LIB=tools/build/base/libmmsum_hip.so python tools/attn_bench.py $c 2>&1 | grep "^$c")"
  done
done 2>&1 | tee gpurun_out/r6_attn_ab.txt
