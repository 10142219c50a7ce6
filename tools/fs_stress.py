#!/usr/bin/env python3
"""Stress of mmsum_gemm's fused split (gemm_nt_ring_kernel<..., FS>): products with alternating operand sets queued back to back, a second
stream running large products beside them; every output must equal, bit for bit, the one the same operands gave the first time.
usage: python tools/fs_stress.py [iterations]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

it = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dt = torch.bfloat16
shapes = [(1152, 1024, 4096), (640, 1024, 4096), (1400, 1024, 4096), (1400, 4096, 1024), (200, 136, 4608), (1152, 1024, 2304)]
sets = []
for i, (M, N, K) in enumerate(shapes):
    for s in range(2):
        g = torch.Generator(device="cuda").manual_seed(100 * i + s)
        a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.5).to(dt)
        b = torch.randn(N, device="cuda", generator=g)
        sets.append((a, w, b))
want = []
for a, w, b in sets:
    o = torch.empty(a.shape[0], w.shape[0], device="cuda", dtype=dt)
    kn.gemm(a, w, o, bias=b)
    want.append(o.clone())
    print("plan", tuple(a.shape), tuple(w.shape), kn.gemm_plan(a, w, o, bias=b), flush=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()
big_a = torch.randn(16384, 4096, device="cuda").to(dt)
big_w = torch.randn(4096, 4096, device="cuda").to(dt)
big_o = torch.empty(16384, 4096, device="cuda", dtype=dt)
outs = [torch.empty_like(o) for o in want]
bad = 0
for mode in ("alone", "beside large products"):
    for rep in range(it // 50):
        if mode != "alone":
            with torch.cuda.stream(side):
                for _ in range(6):
                    kn.gemm(big_a, big_w, big_o)
        for _ in range(50):
            for (a, w, b), o in zip(sets, outs):
                kn.gemm(a, w, o, bias=b)
        torch.cuda.synchronize()
        for i, (o, wnt) in enumerate(zip(outs, want)):
            if not torch.equal(o, wnt):
                bad += 1
                d = (o.float() - wnt.float()).abs()
                print("MISMATCH", mode, "rep", rep, "set", i, "max diff", float(d.max()), "elements", int((d > 0).sum()), flush=True)
    print(mode, "done, mismatches so far:", bad, flush=True)
sys.exit(1 if bad else 0)
