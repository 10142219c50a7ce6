#!/usr/bin/env bash
# Attention with s_setprio 1 around the MFMA bursts (S^T and P V) against the tree's library, interleaved on one box.
# needs tools/build/prio/libmmsum_hip.so (csrc built with -DATTN_PRIO=1)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
for rep in 1 2 3; do
  for c in ${@:-cross_text cross_img4 self_causal}; do
    echo "prio $(MMSUM_LIB=tools/build/prio/libmmsum_hip.so python tools/attn_bench.py $c 2>&1 | grep "^$c")"
    echo "tree $(python tools/attn_bench.py $c 2>&1 | grep "^$c")"
  done
done
