#!/usr/bin/env python3
"""The decode step's weight-streaming products (M = businesses x beams rows), one by one: time per launch inside a captured HIP graph
(the way the decode step issues them) and the weight bytes per second they reach."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16
cases = [("qkv", M, 3072, 1024, 0), ("proj", M, 1024, 1024, 0), ("cross out (3 modalities)", 3 * M, 1024, 1024, 0), ("alpha/beta (A2)", M, 1024, 2048, 0),
         ("fc1+gelu", M, 4096, 1024, kn.EPI_GELU), ("fc2", M, 1024, 4096, 0), ("lm head", M, 50265, 1024, 0)]
for name, m, N, K, epi in cases:
    reps = 20
    xs = [torch.randn(m, K, device="cuda").to(dt) for _ in range(reps)]
    ws = [(torch.randn(N, K, device="cuda") * 0.02).to(dt) for _ in range(reps)]          # distinct weights per launch: nothing stays in cache
    b = torch.zeros(N, device="cuda")
    ys = [torch.empty(m, N, device="cuda", dtype=dt) for _ in range(reps)]
    def run():
        for x, w, y in zip(xs, ws, ys):
            kn.gemm(x, w, y, bias=b, epi=epi)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * reps) * 1e3
    print("%-26s M=%3d N=%5d K=%4d  %6.1f us/launch  %5.2f TB/s of weights" % (name, m, N, K, us, N * K * 2 / us / 1e6), flush=True)
