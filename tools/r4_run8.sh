#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/debug_ddp_rs.py > gpurun_out/r4h_debug_ddp.txt 2>&1
grep -v "^\[\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r4h_debug_ddp.txt | tail -24 | cut -c1-600
