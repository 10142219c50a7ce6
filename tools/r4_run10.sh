#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/debug_rs3.py > gpurun_out/r4i_debug_rs3.txt 2>&1
grep "reduce_scatter:" gpurun_out/r4i_debug_rs3.txt | cut -c1-400
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv3x3 or batchnorm or conv_im2col or statistics" > gpurun_out/r4i_kernel_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4i_kernel_tests.log
tail -5 gpurun_out/r4i_kernel_tests.log
