#!/usr/bin/env bash
# Interleaved A/B of two builds of the C-ABI library on ONE box (boxes of the pool differ by 2-3 %, more than most kernel changes):
# the shipped library against another build loaded through MMSUM_LIB, twice each, alternating; the isolated GEMM launches of the step
# (tools/gemm_epi_bench.py) and the whole step (bench.py without its probe / also / CPU legs).
# usage: tools/gpu_ab.sh <tag> /path/to/other/libmmsum_hip.so [rows]
#   the other build: e.g.  git archive <commit> multimodalsum_amd/csrc include | tar -x -C tools/build/prev && make -C tools/build/prev/multimodalsum_amd/csrc
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-ab}"; other="${2:?path of the other libmmsum_hip.so}"; rows="${3:-129024}"
cd "$R"; mkdir -p gpurun_out
out=gpurun_out/${tag}_ab.txt; : > "$out"
for rep in 1 2; do
  for which in other shipped; do
    if [ $which = other ]; then export MMSUM_LIB="$other"; else unset MMSUM_LIB; fi
    echo "== rep $rep $which: gemm_epi_bench $rows" >> "$out"
    timeout 300 python tools/gemm_epi_bench.py "$rows" 2>&1 | grep -v amdgpu.ids >> "$out"
    echo "== rep $rep $which: bench.py" >> "$out"
    timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f businesses/s  %.2f ms/step' % (d['value'], d['ms_per_step']))" >> "$out"
  done
done
unset MMSUM_LIB
cat "$out"
