#!/usr/bin/env bash
# LDS counters of the step's GEMM kernels (bank conflicts / LDS-array cycles), one --pmc pass per shape.  usage: r5_gemm_lds_pmc.sh
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for shape in "147456 4096 1024 gelu" "147456 1024 1024 nt" "147456 1024 4096 nt" "1024 1024 147456 tn" "4096 1024 147456 tn" "147456 50265 1024 nt"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$R"/gpurun_out/gl_$i -o p -- python3 "$R"/tools/gemm_one.py $shape 4 > "$R"/gpurun_out/gl_$i.log 2>&1
  echo "== $shape"
  (cd "$R" && python tools/pmc_summary.py gemm_ gpurun_out/gl_$i/p_counter_collection.csv)
done
