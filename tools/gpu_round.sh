#!/usr/bin/env bash
# One GPU session of a development round: the -m gpu tests, the default bench line, and a rocprofv3 kernel-stats profile of the
# step (everything lands under gpurun_out/<tag>_*).  usage: tools/gpu_round.sh <tag> [tests|bench|prof ...]
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-run}"; shift || true
what="${*:-tests bench prof}"
cd "$R"; mkdir -p gpurun_out
for w in $what; do
  case $w in
    tests) timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/${tag}_tests.log ;;
    testsall) timeout 2700 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/${tag}_tests.log ;;
    bench) timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc $?" >> gpurun_out/${tag}_bench.err ;;
    benchfast) timeout 600 python bench.py --no-cpu-baseline --no-also > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc $?" >> gpurun_out/${tag}_bench.err ;;
    profgen) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R"/gpurun_out/${tag}_profgen -o r --output-format csv -- python3 "$R"/bench.py --workload generate --steps 2 --warmup 1 > "$R"/gpurun_out/${tag}_profgen.log 2>&1)
          f=$(find gpurun_out/${tag}_profgen -name "*kernel_stats.csv" | head -1); t=$(find gpurun_out/${tag}_profgen -name "*kernel_trace.csv" | head -1)
          if [ -n "$t" ]; then python tools/prof_gaps.py "$t" 0 > gpurun_out/${tag}_profgen_gaps.txt 2>&1; fi
          if [ -n "$f" ]; then python tools/prof_top.py "$f" 378 40 > gpurun_out/${tag}_profgen_summary.txt; fi
          rm -rf gpurun_out/${tag}_profgen ;;
    prof8) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R"/gpurun_out/${tag}_prof8 -o r --output-format csv -- python3 "$R"/bench.py --batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/${tag}_prof8.log 2>&1)
          f=$(find gpurun_out/${tag}_prof8 -name "*kernel_stats.csv" | head -1); t=$(find gpurun_out/${tag}_prof8 -name "*kernel_trace.csv" | head -1)
          if [ -n "$t" ]; then python tools/prof_gaps.py "$t" 20 > gpurun_out/${tag}_prof8_gaps.txt 2>&1; fi
          if [ -n "$f" ]; then python tools/prof_top.py "$f" 0 60 > gpurun_out/${tag}_prof8_summary.txt; fi
          rm -rf gpurun_out/${tag}_prof8 ;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R"/gpurun_out/${tag}_prof -o r --output-format csv -- python3 "$R"/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/${tag}_prof.log 2>&1)
          f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
          t=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
          if [ -n "$t" ]; then python tools/prof_gaps.py "$t" 6 > gpurun_out/${tag}_prof_gaps.txt 2>&1; fi
          if [ -n "$f" ]; then python tools/prof_top.py "$f" 0 60 > gpurun_out/${tag}_prof_summary.txt; python tools/prof_summary.py "$f" 0 >> gpurun_out/${tag}_prof_summary.txt; cp "$f" gpurun_out/${tag}_kernel_stats.csv; fi
          rm -rf gpurun_out/${tag}_prof ;;
  esac
done
tail -5 gpurun_out/${tag}_tests.log 2>/dev/null; cat gpurun_out/${tag}_bench.json 2>/dev/null | cut -c1-600; head -30 gpurun_out/${tag}_prof_summary.txt 2>/dev/null
