#!/usr/bin/env python3
"""Split count of the step's weight-gradient products (dW[N_out, K_in] += dy[R, N_out]^T x[R, K_in], R = 64,512): time of the TN product
+ slab reduction against the number of k slices; the workgroup count is tiles x slices on 256 CUs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64512
for M, N in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (2048, 1024), (1024, 2048)]:
    dy = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    x = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    line = "dW[%4d,%4d] R=%6d " % (M, N, R)
    for sk in (2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
        ws = torch.empty(sk * M, N, device="cuda")
        def f():
            kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
            kn.slab_reduce(ws, sk, out, accumulate=True)
        us = timeit(f) * 1e3
        plan = kn.gemm_plan(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
        line += " sk%-2d %4.0fus(%d)" % (sk, us, plan[3])
    print(line, flush=True)
