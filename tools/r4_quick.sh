#!/usr/bin/env bash
# usage: r4_quick.sh "<pytest -k expression>" [files...]: a subset of the GPU tests
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
k="$1"; shift
timeout 1500 python -m pytest ${@:-tests} -m gpu -q -x -k "$k" > gpurun_out/quick_tests.log 2>&1; echo "rc $?" >> gpurun_out/quick_tests.log
tail -30 gpurun_out/quick_tests.log | cut -c1-300
