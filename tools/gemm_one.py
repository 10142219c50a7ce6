#!/usr/bin/env python3
"""Runs ONE bf16 GEMM shape a few times (for rocprofv3 --pmc passes).
usage: gemm_one.py M N K [nt|tn|gelu|blas] [iters]     gelu = x W^T + bias, GELU, pre-activation saved (the step's dominant kernel);
blas = the same NT product through torch.matmul (hipBLASLt), for counter comparisons"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
tn = mode == "tn"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
if tn:
    a = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16)
else:
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if mode == "gelu" else None
bias = torch.randn(N, device="cuda", generator=g) if mode == "gelu" else None
for _ in range(iters):
    if mode == "gelu":
        kn.gemm(a, b, out, bias=bias, epi=kn.EPI_GELU, aux=aux)
    elif mode == "blas":
        torch.matmul(a, b.t(), out=out)
    else:
        kn.gemm(a, b, out, a_t=tn, b_t=tn)
torch.cuda.synchronize()
