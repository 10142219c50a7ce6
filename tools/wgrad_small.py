#!/usr/bin/env python3
"""Weight-gradient products with a small output and a very long reduction (the ResNet layer3 convolutions of the step: dW[Cout, K] =
dy[R, Cout]^T x[R, K], R = 224 * 14 * 14): TN kernel at several split counts, with the slab reduction."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

R = int(sys.argv[1]) if len(sys.argv) > 1 else 43904
for M, N in [(256, 1024), (256, 2304), (1024, 256), (512, 1024), (1024, 1024)]:
    dy = torch.randn(R, M, device="cuda").to(torch.bfloat16)
    x = torch.randn(R, N, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    line = "dW[%4d,%4d] R=%6d " % (M, N, R)
    for sk in (1, 2, 4, 8, 16, 32, 64):
        ws = torch.empty(sk * M, N, device="cuda")
        def g():
            if sk == 1:
                kn.gemm(dy, x, out, a_t=True, b_t=True, accumulate=True)
            else:
                kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
                kn.slab_reduce(ws, sk, out, accumulate=True)
        us = timeit(g) * 1e3
        plan = kn.gemm_plan(dy, x, ws if sk > 1 else out, a_t=True, b_t=True, splitk=sk, slabs=sk > 1, accumulate=sk == 1)
        line += " sk%-2d %4.0fus %4.0fTF %s|" % (sk, us, 2.0 * M * N * R / us / 1e6, "x".join(str(v) for v in plan[1:3]))
    print(line, flush=True)
