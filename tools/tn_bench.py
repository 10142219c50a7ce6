#!/usr/bin/env python3
"""The step's weight-gradient products through the TN kernel as engine.wgrad launches them: dW[N_out, K_in] += dy[R, N_out]^T x[R, K_in],
split-K slabs + slab_reduce (R = 64,512 decoder rows at B = 56), and the tied-embedding gradient (one launch, accumulating)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64512
for M, N in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (2048, 1024)]:
    dy = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    x = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = max(1, min(32, -(-768 // tiles), R // 64 // 4))
    ws = torch.empty(sk * M, N, device="cuda")
    def f():
        kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
        kn.slab_reduce(ws, sk, out, accumulate=True)
    us = timeit(f) * 1e3
    us1 = timeit(lambda: kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)) * 1e3
    print("dW[%4d,%4d] R=%6d sk=%2d  %6.0f us with reduce (%5.0f TF/s)   product alone %6.0f us (%5.0f TF/s)  plan %s" % (
        M, N, R, sk, us, 2.0 * M * N * R / us / 1e6, us1, 2.0 * M * N * R / us1 / 1e6,
        kn.gemm_plan(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)), flush=True)
V, Vpad, D = 50265, 50304, 1024
dl = torch.randn(R, Vpad, device="cuda").to(torch.bfloat16)
h = torch.randn(R, D, device="cuda").to(torch.bfloat16)
out = torch.zeros(V, D, device="cuda")
us = timeit(lambda: kn.gemm(dl[:, :V], h, out, a_t=True, b_t=True, accumulate=True), iters=3) * 1e3
print("dE[%d,%d] R=%d accumulate  %6.0f us (%5.0f TF/s)" % (V, D, R, us, 2.0 * V * D * R / us / 1e6))
