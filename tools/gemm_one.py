#!/usr/bin/env python3
"""Runs ONE bf16 GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K [tn] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tn = len(sys.argv) > 4 and sys.argv[4] == "tn"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
if tn:
    a = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16)
else:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    kn.gemm(a, b, out, a_t=tn, b_t=tn)
torch.cuda.synchronize()
