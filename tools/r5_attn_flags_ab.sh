#!/usr/bin/env bash
# attention.hip compiled with other scheduler flags (tools/build/att_<tag>/libmmsum_hip.so) against the tree's library, interleaved on one box.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
for rep in 1 2; do
  for c in cross_text cross_img4 self_causal; do
    for tag in tree "$@"; do
      if [ $tag = tree ]; then lib=""; else lib="tools/build/att_$tag/libmmsum_hip.so"; fi
      echo "$tag $(MMSUM_LIB=$lib python tools/attn_bench.py $c 2>&1 | grep "^$c" | cut -c1-100)"
    done
  done
done
