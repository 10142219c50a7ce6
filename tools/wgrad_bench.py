#!/usr/bin/env python3
"""wgrad-shaped products (small output, long reduction): split-K slabs at several split counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

KB = int(sys.argv[1]) if len(sys.argv) > 1 else 16128
for M, N, K in [(1024, 1024, KB), (3072, 1024, KB), (4096, 1024, KB), (2048, 1024, KB * 55552 // 32256)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    line = "M=%5d N=%5d K=%6d " % (M, N, K)
    for sk in (1, 2, 4, 8, 16):
        ws = torch.empty(sk * M, N, device="cuda")
        def f():
            if sk == 1:
                kn.gemm(a, b, out, accumulate=True)
            else:
                kn.gemm(a, b, ws, splitk=sk, slabs=True)
                kn.slab_reduce(ws, sk, out, accumulate=True)
        us = timeit(f) * 1e3
        line += " sk%-2d %5.0fus %5.0fTF |" % (sk, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
    # the same product straight from reduction-major operands (TN kernel, no transposed copies)
    at, bt = a.t().contiguous(), b.t().contiguous()
    line = "   TN (dy[K,M], x[K,N])      "
    for sk in (1, 2, 4, 8, 16):
        ws = torch.empty(sk * M, N, device="cuda")
        def g():
            if sk == 1:
                kn.gemm(at, bt, out, a_t=True, b_t=True, accumulate=True)
            else:
                kn.gemm(at, bt, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
                kn.slab_reduce(ws, sk, out, accumulate=True)
        us = timeit(g) * 1e3
        line += " sk%-2d %5.0fus %5.0fTF |" % (sk, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
