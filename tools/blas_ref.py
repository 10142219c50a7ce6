#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the step's dominant shapes, beside mmsum_gemm: a calibration of
how far the ring kernels are from the best known code for this chip, not a dependency of the product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dt = torch.bfloat16
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64512            # 129024 = the bench batch (B = 112)
    for M, N, K in [(rows, 4096, 1024), (rows, 1024, 4096), (rows, 1024, 1024), (rows, 3072, 1024), (rows * 38912 // 64512, 4096, 1024)]:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randn(N, K, device="cuda").to(dt) * 0.02
        y = torch.empty(M, N, device="cuda", dtype=dt)
        t_blas = timeit(lambda: torch.matmul(x, w.t(), out=y))
        t_ours = timeit(lambda: kn.gemm(x, w, y))
        fl = 2.0 * M * N * K
        print("NT  M=%6d N=%5d K=%5d   hipBLASLt %7.1f us (%6.1f TF/s)   mmsum %7.1f us (%6.1f TF/s)" % (M, N, K, t_blas, fl / t_blas / 1e6, t_ours, fl / t_ours / 1e6), flush=True)
    for R, N, K in [(rows, 4096, 1024), (rows, 1024, 4096), (rows, 1024, 1024)]:        # wgrad: dW[N,K] = dy[R,N]^T x[R,K]
        dy = torch.randn(R, N, device="cuda").to(dt)
        x = torch.randn(R, K, device="cuda").to(dt)
        dw = torch.empty(N, K, device="cuda", dtype=torch.float32)
        dwb = torch.empty(N, K, device="cuda", dtype=dt)
        t_blas = timeit(lambda: torch.matmul(dy.t(), x, out=dwb))
        fl = 2.0 * R * N * K
        print("TN  R=%6d N=%5d K=%5d   hipBLASLt %7.1f us (%6.1f TF/s)" % (R, N, K, t_blas, fl / t_blas / 1e6), flush=True)


if __name__ == "__main__":
    main()
