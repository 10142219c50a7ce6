#!/usr/bin/env bash
# NT GEMM A/B on one box: kernel tests, then tools/gemm_epi_bench.py with the tree's library and with tools/build/base/libmmsum_hip.so, interleaved.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "${1:-gemm or nt or ring or conv or epilog}" > gpurun_out/gab_tests.log 2>&1; echo "rc $?" >> gpurun_out/gab_tests.log
tail -4 gpurun_out/gab_tests.log | cut -c1-200
M="${2:-147456}"
for rep in 1 2; do
  python tools/gemm_epi_bench.py $M > gpurun_out/gab_new_$rep.txt 2>&1
  MMSUM_LIB=tools/build/base/libmmsum_hip.so python tools/gemm_epi_bench.py $M > gpurun_out/gab_base_$rep.txt 2>&1
done
paste gpurun_out/gab_new_1.txt gpurun_out/gab_base_1.txt | cut -c1-75,100-175
echo ---- ; paste gpurun_out/gab_new_2.txt gpurun_out/gab_base_2.txt | cut -c1-75,100-175
