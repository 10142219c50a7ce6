#!/bin/bash
# GEMM A/B session on the GPU box: the step's products at the bench batch with the shipped library and with tools/ builds
# (tools/build/*.so, built by hand from csrc with -DMMSUM_EXP_* / -DMMSUM_DIAG_*; never shipped).
cd "$(dirname "$0")/.."
out=gpurun_out/exp_gemm.log
: > $out
for lib in default noepi 2wg1 2wg2 nt1 nt2; do
  for M in 64512 38912; do
    echo "=== lib=$lib M=$M" >> $out
    if [ $lib = default ]; then python tools/gemm_epi_bench.py $M >> $out 2>&1
    else MMSUM_LIB=$PWD/tools/build/libmmsum_$lib.so python tools/gemm_epi_bench.py $M >> $out 2>&1; fi
  done
done
for lib in 2wg1 2wg2 nt1; do
  echo "=== parity lib=$lib" >> $out
  MMSUM_LIB=$PWD/tools/build/libmmsum_$lib.so python -m pytest tests/test_bench_shapes_gpu.py -q -k "nt_ring" -p no:cacheprovider 2>&1 | tail -3 >> $out
done
