#!/usr/bin/env python3
"""t(K) of the bf16 NT GEMM at fixed M,N: the slope is the main-loop rate, the intercept the per-tile fixed cost
(launch + pipeline fill + epilogue).  usage: gemm_ksweep.py M N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

M, N = int(sys.argv[1]), int(sys.argv[2])
for K in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: kn.gemm(a, b, out), iters=20)
    print("M=%d N=%d K=%5d  %8.1f us  %7.1f TFLOP/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
