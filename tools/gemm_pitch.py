#!/usr/bin/env python3
"""Does a power-of-two row pitch cost bandwidth (L2 channel hot spots)?  The same NT GEMM with operands / result at their natural
leading dimension and padded by `pad` elements.  usage: gemm_pitch.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit

M, N, K = [int(v) for v in sys.argv[1:4]]
for pa, pb, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 32), (136, 136, 136)]:
    a = torch.randn(M, K + pa, device="cuda").to(torch.bfloat16)[:, :K]
    b = torch.randn(N, K + pb, device="cuda").to(torch.bfloat16)[:, :K]
    out = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16)[:, :N]
    ms = timeit(lambda: kn.gemm(a, b, out), iters=20)
    print("M=%d N=%d K=%d  pad A/B/C = %3d/%3d/%3d  %8.1f us  %7.1f TFLOP/s" % (M, N, K, pa, pb, pc, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
