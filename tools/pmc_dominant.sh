#!/usr/bin/env bash
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the repository root (gpurun exports GRAFT_REPO_ROOT)
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES:s" "TCC_HIT_sum TCC_MISS_sum:t"; do
  ctr="${c%%:*}"; tag="${c##*:}"
  rocprofv3 --output-format csv --kernel-trace --pmc $ctr -d "$R"/gpurun_out/pmcF_$tag -o $tag -- python3 "$R"/tools/gemm_one.py ${PMC_M:-129024} 4096 1024 gelu 6 > "$R"/gpurun_out/pmcF_$tag.log 2>&1
done
cd "$R"
python tools/pmc_summary.py gemm_nt_w4 gpurun_out/pmcF_f/f_counter_collection.csv gpurun_out/pmcF_w/w_counter_collection.csv gpurun_out/pmcF_s/s_counter_collection.csv gpurun_out/pmcF_t/t_counter_collection.csv
python tools/pmc_dominant_json.py ${PMC_ROUND:-4} ${PMC_M:-129024} 4096 1024 gpurun_out/pmcF_dominant.json gpurun_out/pmcF_f/f_counter_collection.csv gpurun_out/pmcF_w/w_counter_collection.csv gpurun_out/pmcF_s/s_counter_collection.csv gpurun_out/pmcF_t/t_counter_collection.csv
