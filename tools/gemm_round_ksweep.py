#!/usr/bin/env python3
"""t(K) of ONE round of 256x256 tiles (R = N = 4096: 256 tiles) and of two rounds (R = 8192), graph-captured with distinct operands per
launch, mmsum_gemm next to hipBLASLt: slope = the main loop per 1,024 of K with every CU busy, intercept = launch + fill + epilogue.
usage: python tools/gemm_round_ksweep.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_small_bench import graph_time, REPS

dt = torch.bfloat16
for R, N in ((4096, 4096), (8192, 4096)):
    for K in (256, 512, 1024, 2048, 4096):
        xs = [torch.randn(R, K, device="cuda").to(dt) for _ in range(REPS)]
        ws = [(torch.randn(N, K, device="cuda") * 0.02).to(dt) for _ in range(REPS)]
        ys = [torch.empty(R, N, device="cuda", dtype=dt) for _ in range(REPS)]

        def ours():
            for x, w, y in zip(xs, ws, ys):
                kn.gemm(x, w, y)

        def blas():
            for x, w, y in zip(xs, ws, ys):
                torch.mm(x, w.t(), out=y)
        t0, t1 = graph_time(ours), graph_time(blas)
        print("R=%5d N=%d K=%4d   mmsum_gemm %6.1f us (%6.1f TF/s)   hipBLASLt %6.1f us (%6.1f TF/s)"
              % (R, N, K, t0, 2.0 * R * N * K / t0 / 1e6, t1, 2.0 * R * N * K / t1 / 1e6), flush=True)
