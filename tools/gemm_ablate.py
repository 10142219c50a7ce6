#!/usr/bin/env python3
"""Attributes the launch time of the 256x256 ring GEMM by ablation (MMSUM_GEMM_ABLATE, results are wrong by construction):
1 = no DMA after the first ring fill, 2 = no MFMA, 3 = no epilogue.  usage: MMSUM_GEMM_ABLATE=n python tools/gemm_ablate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

for M, N, K in [(32256, 1024, 1024), (32256, 4096, 1024), (32256, 1024, 4096), (16128, 1024, 50304)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: kn.gemm(a, b, out), iters=20)
    print("ablate=%s M=%d N=%d K=%5d  %8.1f us" % (os.environ.get("MMSUM_GEMM_ABLATE", "0"), M, N, K, ms * 1e3), flush=True)
