#!/usr/bin/env python3
"""Time of the bf16 NT product against the number of 256x256 tiles (N, K fixed; rows = 256 x tiles / (N / 256)), graph-captured with
distinct operands per launch, next to hipBLASLt: what one round of the CUs costs when 16 .. 512 tiles run at once.
usage: python tools/gemm_rounds_bench.py [N K]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_small_bench import graph_time, REPS

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 1024)
dt = torch.bfloat16
for tiles in (16, 32, 64, 128, 192, 256, 288, 320, 384, 512, 576, 768):
    R = 256 * tiles // (N // 256)
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
    print("R=%5d N=%d K=%d  256^2 tiles %4d   mmsum_gemm %6.1f us (%6.1f TF/s)   hipBLASLt %6.1f us (%6.1f TF/s)   plan %s"
          % (R, N, K, tiles, t0, 2.0 * R * N * K / t0 / 1e6, t1, 2.0 * R * N * K / t1 / 1e6, kn.gemm_plan(xs[0], ws[0], ys[0])), flush=True)
