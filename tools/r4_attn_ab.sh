#!/usr/bin/env bash
# interleaved A/B of two library builds on the attention micro-benchmark (text memory with padding, compact K / V through row maps)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=${ATTN_BENCH_PADS:-1} ATTN_BENCH_MAPS=1
for rep in 1 2; do
  echo NEW; python tools/attn_bench.py ${1:-cross_text} 2>&1 | grep -v amdgpu.ids
  echo BASE; MMSUM_LIB=tools/build/base/libmmsum_hip.so python tools/attn_bench.py ${1:-cross_text} 2>&1 | grep -v amdgpu.ids
done
