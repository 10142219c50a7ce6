#!/usr/bin/env bash
# A/B of the four-wave NT kernel's epilogue (MMSUM_W4_DIRECT) and tile order (MMSUM_RASTER) at the step's shapes: every
# combination twice, interleaved, each in its own process (the library reads the switches once).  usage: tools/gpu_ab.sh <tag>
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-ab}"
cd "$R"; mkdir -p gpurun_out
: > gpurun_out/${tag}_ab.txt
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo "== rep $rep MMSUM_W4_DIRECT=$1 MMSUM_RASTER=$2" >> gpurun_out/${tag}_ab.txt
    MMSUM_W4_DIRECT=$1 MMSUM_RASTER=$2 timeout 300 python tools/gemm_epi_bench.py 64512 >> gpurun_out/${tag}_ab.txt 2>&1
  done
done
echo "== ksweep direct" >> gpurun_out/${tag}_ab.txt
MMSUM_W4_DIRECT=1 timeout 200 python tools/gemm_ksweep.py 64512 1024 >> gpurun_out/${tag}_ab.txt 2>&1
echo "== ksweep staged" >> gpurun_out/${tag}_ab.txt
MMSUM_W4_DIRECT=0 timeout 200 python tools/gemm_ksweep.py 64512 1024 >> gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
