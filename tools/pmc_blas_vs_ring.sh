#!/usr/bin/env bash
# Hardware counters of one NT product (M N K = $1 $2 $3, default the FFN down-projection of the step) through the ring kernel and
# through hipBLASLt: clock under load (GRBM_GUI_ACTIVE / duration), matrix-pipe busy share, instruction mix per launch.
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the repository root (gpurun exports GRAFT_REPO_ROOT)
cd /tmp && export TMPDIR=/tmp
M=${1:-64512}; N=${2:-1024}; K=${3:-4096}
for mode in nt blas; do
  i=0
  for ctr in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
             "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --output-format csv --kernel-trace --pmc $ctr -d "$R"/gpurun_out/pbr_${mode}_$i -o p -- python3 "$R"/tools/gemm_one.py $M $N $K $mode 6 > "$R"/gpurun_out/pbr_${mode}_$i.log 2>&1
  done
done
cd "$R"
echo "== mmsum_gemm (gemm_nt_w4_kernel for 256x256 tiles)"; python tools/pmc_summary.py gemm_nt_ gpurun_out/pbr_nt_1/p_counter_collection.csv gpurun_out/pbr_nt_2/p_counter_collection.csv gpurun_out/pbr_nt_3/p_counter_collection.csv
echo "== hipBLASLt"; python tools/pmc_summary.py Cijk gpurun_out/pbr_blas_1/p_counter_collection.csv gpurun_out/pbr_blas_2/p_counter_collection.csv gpurun_out/pbr_blas_3/p_counter_collection.csv
grep -h "Cijk" gpurun_out/pbr_blas_1/p_kernel_trace.csv | head -1 | cut -c1-400
