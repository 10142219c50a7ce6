#!/usr/bin/env bash
# rocprofv3 kernel stats of the step with the image / table branch on the main stream (nothing runs beside it: the image kernels' own durations)
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out
export MMSUM_SIDE_STREAM=0
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R"/gpurun_out/noside_prof -o r --output-format csv -- python3 "$R"/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/noside_prof.log 2>&1)
f=$(find gpurun_out/noside_prof -name "*kernel_stats.csv" | head -1)
python tools/prof_top.py "$f" 0 70 > gpurun_out/noside_prof_summary.txt; python tools/prof_summary.py "$f" 0 >> gpurun_out/noside_prof_summary.txt; cp "$f" gpurun_out/noside_kernel_stats.csv
rm -rf gpurun_out/noside_prof
cat gpurun_out/noside_prof_summary.txt | head -90
