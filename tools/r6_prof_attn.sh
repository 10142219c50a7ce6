#!/usr/bin/env bash
# PMC counters of the text cross-attention forward: the tree's library (attn_w64_fwd, round 6) and tools/build/base (attn_tr_fwd, round 5), same box.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
export ATTN_BENCH_B=${ATTN_BENCH_B:-128} ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
for v in new base; do
  if [ $v = base ]; then export MMSUM_LIB="$R"/tools/build/base/libmmsum_hip.so; else unset MMSUM_LIB; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d "$R"/gpurun_out/r6pa_${v}_k -o k -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/r6pa_${v}_k.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$R"/gpurun_out/r6pa_${v}_1 -o p -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/r6pa_${v}_1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d "$R"/gpurun_out/r6pa_${v}_2 -o p -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/r6pa_${v}_2.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAVES -d "$R"/gpurun_out/r6pa_${v}_3 -o p -- python3 "$R"/tools/attn_bench.py cross_text > "$R"/gpurun_out/r6pa_${v}_3.log 2>&1
done
cd "$R"
for v in new base; do
  echo "======== $v"
  python tools/prof_top.py gpurun_out/r6pa_${v}_k/k_kernel_stats.csv 1 4
  k=attn_w64_fwd; [ $v = base ] && k=attn_tr_fwd
  python tools/pmc_summary.py $k gpurun_out/r6pa_${v}_1/p_counter_collection.csv gpurun_out/r6pa_${v}_2/p_counter_collection.csv gpurun_out/r6pa_${v}_3/p_counter_collection.csv
done > gpurun_out/r6_attention_fwd_pmc.txt 2>&1
rm -rf gpurun_out/r6pa_*_k gpurun_out/r6pa_*_1 gpurun_out/r6pa_*_2 gpurun_out/r6pa_*_3
cat gpurun_out/r6_attention_fwd_pmc.txt
