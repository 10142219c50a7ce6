#!/usr/bin/env python3
"""mmsum_gemm with the epilogues of the training step (bias, GELU + saved pre-activation, GELU', column sums), HIP-event
timed on rotating buffers so that the operands come from HBM as they do in the step.  A/B: MMSUM_LIB=/path/to/another/libmmsum_hip.so."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from multimodalsum_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None
CASES = [("fc1+bias", 4096, 1024, "bias"), ("proj+bias", 1024, 1024, "bias"), ("qkv+bias", 3072, 1024, "bias"), ("fc1+bias+gelu", 4096, 1024, "gelu"),
         ("fc2+bias", 1024, 4096, "bias"), ("dgrad plain", 1024, 1024, "none"), ("dgrad fc2 gelu'", 4096, 1024, "gelu_bwd"),
         ("dgrad fc1", 1024, 4096, "none"), ("dgrad accumulate", 1024, 1024, "acc"), ("dgrad fc1 accum", 1024, 4096, "acc"),
         ("kv+bias", 2048, 1024, "bias"), ("dgrad qkv accum", 1024, 3072, "acc")]


def main():
    dt = torch.bfloat16
    nbuf = 6
    tot = 0.0
    for name, N, K, kind in CASES:
        if ONLY is not None and name not in ONLY:
            continue
        a = [torch.randn(M, K, device="cuda").to(dt) for _ in range(nbuf)]
        b = [torch.randn(N, K, device="cuda").to(dt) * 0.03 for _ in range(nbuf)]
        out = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nbuf)]
        aux = [torch.randn(M, N, device="cuda").to(dt) for _ in range(nbuf)]
        bias = torch.randn(N, device="cuda")
        cs = torch.zeros(N, device="cuda")

        def run(i):
            j = i % nbuf
            if kind == "bias":
                kn.gemm(a[j], b[j], out[j], bias=bias)
            elif kind == "gelu":
                kn.gemm(a[j], b[j], out[j], bias=bias, epi=_lib.EPI_GELU, aux=aux[j])
            elif kind == "gelu_bwd":
                kn.gemm(a[j], b[j], out[j], epi=_lib.EPI_GELU_BWD, aux=aux[j], colsum=cs)
            elif kind == "acc":
                kn.gemm(a[j], b[j], out[j], accumulate=True)
            else:
                kn.gemm(a[j], b[j], out[j])
        for i in range(3):
            run(i)
        iters = 24
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tot += ms
        print("%-18s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
    print("sum %.1f us" % (tot * 1e3))


main()
