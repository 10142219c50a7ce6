#!/usr/bin/env bash
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
bash tools/r4_evidence.sh > gpurun_out/r04_evidence.log 2>&1
timeout 2700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r04_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r04_smoke.log
tail -40 gpurun_out/r04_evidence.log | cut -c1-250; tail -4 gpurun_out/r04_tests.log; tail -3 gpurun_out/r04_smoke.log
