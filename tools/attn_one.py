#!/usr/bin/env python3
"""One attention case, a few launches (for rocprofv3 --pmc / --kernel-trace).  usage: attn_one.py [cross_text|self|cross_img|cross_table] [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

name = sys.argv[1] if len(sys.argv) > 1 else "cross_text"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 28
CASES = {"cross_text": (NB, 9, 9, 128, 128, True, False), "self": (9 * NB, 1, 1, 128, 128, False, False),
         "cross_img": (NB, 9, 4, 196, 128, False, False), "cross_table": (NB, 9, 1, 47, 128, False, False)}
B, qpb, N, S, T, excl, causal = CASES[name]
H, D, dt = 16, 1024, torch.bfloat16
nq = B * qpb
q = torch.randn(nq * T, D, device="cuda").to(dt)
kv = torch.randn(B * N * S, 2 * D, device="cuda").to(dt)
out = torch.empty(nq * T, D, device="cuda", dtype=dt)
pad = torch.zeros(B * N * S, dtype=torch.uint8, device="cuda")
null = torch.zeros(B * N, dtype=torch.uint8, device="cuda")
desc = kn.make_attn_desc(q, kv[:, :D], kv[:, D:], out, pad, null, nq, T, qpb, N, S, H, excl, causal, 0.125)
dout = torch.randn(nq * T, D, device="cuda").to(dt)
dq = torch.empty_like(q)
dkv = torch.empty_like(kv)
stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device="cuda")
for _ in range(4):
    kn.attn_fwd(desc, q)
    kn.attn_bwd(desc, dout, dq, False, dkv[:, :D], dkv[:, D:], stats)
torch.cuda.synchronize()
