#!/usr/bin/env python3
"""Micro-benchmark of mmsum_gemm on the shapes of the training step (HIP-event timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

SHAPES14 = [(16128, 1024, 1024), (16128, 3072, 1024), (16128, 4096, 1024), (16128, 1024, 4096), (16128, 1024, 2048), (27762, 2048, 1024),
            (48384, 1024, 1024), (16128, 50265, 1024), (16128, 1024, 50304)]
SHAPES = [(9216, 1024, 1024), (9216, 3072, 1024), (9216, 4096, 1024), (9216, 1024, 4096), (15864, 2048, 1024),
          (27648, 1024, 1024), (9216, 50265, 1024), (1024, 1024, 9216), (4096, 1024, 9216), (50265, 1024, 9216)]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dt = torch.bfloat16
    shapes = SHAPES14 if (len(sys.argv) > 1 and sys.argv[1] == 'b14') else SHAPES
    if len(sys.argv) > 3:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for M, N, K in shapes:
        a = torch.randn(M, K, device="cuda").to(dt)
        b = torch.randn(N, K, device="cuda").to(dt)
        ld = (N + 127) // 128 * 128
        out = torch.empty(M, ld, device="cuda", dtype=dt)[:, :N]
        ms = timeit(lambda: kn.gemm(a, b, out))
        line = "NT  M=%6d N=%6d K=%5d  %8.3f ms  %7.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9)
        if M * N <= 4096 * 4096 and K >= 4096:
            outf = torch.zeros(M, N, device="cuda")
            for sk in (1, 4, 8):
                ms = timeit(lambda: kn.gemm(a, b, outf, accumulate=True, splitk=sk))
                line += "  | f32-accum splitk=%d %7.1f" % (sk, 2.0 * M * N * K / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
