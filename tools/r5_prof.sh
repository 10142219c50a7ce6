#!/usr/bin/env bash
# usage: r5_prof.sh <tag> [noside]: rocprofv3 kernel trace of the step -> steady-state per-step table (tools/prof_steady.py) + idle gaps
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
tag="${1:-r05}"; [ "${2:-}" = "noside" ] && export MMSUM_SIDE_STREAM=0
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R"/gpurun_out/${tag}_prof -o r --output-format csv -- python3 "$R"/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/${tag}_prof.log 2>&1)
t=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
python tools/prof_steady.py "$t" 6 70 > gpurun_out/${tag}_summary.txt 2>&1
python tools/prof_gaps.py "$t" 6 > gpurun_out/${tag}_gaps.txt 2>&1
cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
head -100 gpurun_out/${tag}_summary.txt
