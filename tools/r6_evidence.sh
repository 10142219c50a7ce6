#!/usr/bin/env bash
# The round's evidence on ONE box: full GPU test suite, the bench line, the steady-state rocprofv3 summaries (side stream on / off), the
# generation profile, the dominant kernel's PMC passes.  usage: r6_evidence.sh [tests] [bench] [prof] [pmc] [small]
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
    small) # the small-batch evidence: the step's products one by one, one-round products against hipBLASLt, the host's timeline, the B = 8 / B = 1 profiles
          { timeout 300 python tools/gemm_small_bench.py 640 1152 2304 5000 9216; } 2>&1 | grep "NT\|TN" > gpurun_out/r06_gemm_small_products.txt
          { echo "# one tile alone: t(K) (tools/gemm_ksweep.py)"; for sh in "128 128" "256 128" "256 256"; do timeout 120 python tools/gemm_ksweep.py $sh; done
            echo "# time against the number of 256x256 tiles (tools/gemm_rounds_bench.py)"; timeout 200 python tools/gemm_rounds_bench.py 4096 1024
            echo "# t(K) of one and two rounds (tools/gemm_round_ksweep.py)"; timeout 200 python tools/gemm_round_ksweep.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_gemm_rounds.txt
          { timeout 300 python tools/host_timeline.py 1 40; timeout 300 python tools/host_timeline.py 8 30; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_host_timeline.txt
          for b in 8 1; do
            ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/bprof" -o r --output-format csv -- python3 "$R/bench.py" --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-probe --no-also > "$R/gpurun_out/bprof.log" 2>&1 )
            t=$(find gpurun_out/bprof -name "*kernel_trace.csv" | head -1)
            python tools/prof_gaps.py "$t" 20 > gpurun_out/r06_step_B${b}_gaps.txt 2>&1; python tools/prof_steady.py "$t" 20 70 > gpurun_out/r06_step_B${b}_summary.txt 2>&1; rm -rf gpurun_out/bprof
          done
          cp gpurun_out/r06_gemm_small_products.txt gpurun_out/r06_gemm_rounds.txt gpurun_out/r06_host_timeline.txt gpurun_out/r06_step_B8_*.txt gpurun_out/r06_step_B1_*.txt profiles/ 2>/dev/null
          head -4 gpurun_out/r06_host_timeline.txt; head -3 gpurun_out/r06_step_B8_gaps.txt | cut -c1-120 ;;
  esac
done
