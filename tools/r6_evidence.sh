#!/usr/bin/env bash
# The round's evidence on ONE box: full GPU test suite, the bench line, the steady-state rocprofv3 summaries (side stream on / off), the
# generation profile, the dominant kernel's PMC passes.  usage: r6_evidence.sh [tests] [bench] [prof] [pmc]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
what="${*:-tests bench prof pmc}"
for w in $what; do
  case $w in
    tests) timeout 3000 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r06_gpu_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r06_gpu_tests.log
           timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_gpu_tests.log 2>&1; echo "smoke rc $?" >> gpurun_out/r06_gpu_tests.log
           tail -25 gpurun_out/r06_gpu_tests.log | cut -c1-200 ;;
    pmc) PMC_M=147456 PMC_ROUND=6 bash tools/pmc_dominant.sh > gpurun_out/r06_pmc.log 2>&1; tail -3 gpurun_out/r06_pmc.log | cut -c1-300
         cp gpurun_out/pmcF_dominant.json profiles/r06_dominant_gemm_pmc_B128.json 2>/dev/null; cp gpurun_out/pmcF_dominant.json gpurun_out/r06_dominant_gemm_pmc_B128.json 2>/dev/null ;;
    bench) timeout 1500 python bench.py > gpurun_out/r06_bench_B128.json 2> gpurun_out/r06_bench_B128.err; echo "bench rc $?"; cut -c1-400 gpurun_out/r06_bench_B128.json ;;
    prof) bash tools/r5_prof.sh r06_step_B128 > /dev/null 2>&1; bash tools/r5_prof.sh r06_step_B128_noside noside > /dev/null 2>&1
          bash tools/gpu_round.sh r06_generate_B8 profgen > /dev/null 2>&1
          head -12 gpurun_out/r06_step_B128_summary.txt; tail -20 gpurun_out/r06_step_B128_noside_summary.txt ;;
  esac
done
