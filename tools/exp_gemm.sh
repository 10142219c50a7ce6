#!/bin/bash
# GEMM A/B session on the GPU box: the step's products at the bench batch with the shipped library and with tools/ builds
# (tools/build/*.so, built by hand from csrc with -DMMSUM_* switches; never shipped).
cd "$(dirname "$0")/.."
out=gpurun_out/exp_gemm.log
: > $out
for rep in 1 2; do
for lib in default nostagger; do
    echo "=== lib=$lib rep=$rep" >> $out
    if [ $lib = default ]; then python tools/gemm_epi_bench.py 64512 2>&1 | grep -v amdgpu >> $out
    else MMSUM_LIB=$PWD/tools/build/libmmsum_$lib.so python tools/gemm_epi_bench.py 64512 2>&1 | grep -v amdgpu >> $out; fi
done
done
