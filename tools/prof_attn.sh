#!/usr/bin/env bash
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the repository root (gpurun exports GRAFT_REPO_ROOT)
cd /tmp && export TMPDIR=/tmp
export ATTN_BENCH_B=${ATTN_BENCH_B:-128} ATTN_BENCH_PADS=1
rocprofv3 --output-format csv --kernel-trace --stats -d "$R"/gpurun_out/pa_k -o k -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/pa_k.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$R"/gpurun_out/pa_1 -o p -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/pa_1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d "$R"/gpurun_out/pa_2 -o p -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/pa_2.log 2>&1
cd "$R"
python tools/prof_top.py gpurun_out/pa_k/k_kernel_stats.csv 1 8
for k in attn_tr_fwd attn_tr_bwd_dq attn_tr_bwd_dkv; do echo "== $k"; python tools/pmc_summary.py $k gpurun_out/pa_1/p_counter_collection.csv gpurun_out/pa_2/p_counter_collection.csv; done
