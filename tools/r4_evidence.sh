#!/usr/bin/env bash
# Round-4 evidence on ONE box: default bench line, rocprofv3 summaries of the step and of generation, PMC passes of the dominant GEMM
# (-> the JSON roofline.traffic reads) and of the text cross-attention kernels.  Everything lands in gpurun_out/r04_*.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out
PMC_ROUND=4 PMC_M=147456 bash tools/pmc_dominant.sh > gpurun_out/r04_pmc_dominant.txt 2>&1
cp gpurun_out/pmcF_dominant.json gpurun_out/r04_dominant_gemm_pmc_B128.json 2>/dev/null
cp gpurun_out/r04_dominant_gemm_pmc_B128.json profiles/ 2>/dev/null          # so that the bench line below reads THIS box's counters
timeout 1200 python bench.py > gpurun_out/r04_bench_B128.json 2> gpurun_out/r04_bench_B128.err
bash tools/gpu_round.sh r04 prof profgen > /dev/null 2>&1
ATTN_BENCH_MAPS=1 bash tools/prof_attn.sh > gpurun_out/r04_attention_pmc.txt 2>&1
cat gpurun_out/r04_bench_B128.json | cut -c1-300; tail -3 gpurun_out/r04_pmc_dominant.txt; head -12 gpurun_out/r04_prof_summary.txt; tail -30 gpurun_out/r04_attention_pmc.txt
