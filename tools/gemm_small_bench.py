#!/usr/bin/env python3
"""The small-batch step's products (B = 1: 1,152 decoder rows, ~640 encoder rows; B = 8: 9,216 / ~5,000) inside a captured graph with
distinct operands per launch (nothing stays in cache between launches), next to hipBLASLt through torch.mm on the same operands:
  NT: y[R, N] = x[R, K] w[N, K]^T     (forward and dgrad products)
  TN: dW[N, K] = dy[R, N]^T x[R, K]   (weight gradients, split-K slabs + slab_reduce as engine.wgrad issues them)
usage: python tools/gemm_small_bench.py [rows ...]   (default 640 1152 5000 9216)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from multimodalsum_amd.engine import splitk_rule

rows = [int(a) for a in sys.argv[1:]] or [640, 1152, 5000, 9216]
dt = torch.bfloat16
REPS = 16


def graph_time(run):
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
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


def main():
    for R in rows:
        for N, K in [(1024, 1024), (2048, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]:
            xs = [torch.randn(R, K, device="cuda").to(dt) for _ in range(REPS)]
            ws = [(torch.randn(N, K, device="cuda") * 0.02).to(dt) for _ in range(REPS)]
            ys = [torch.empty(R, N, device="cuda", dtype=dt) for _ in range(REPS)]
            b = torch.zeros(N, device="cuda")

            def ours():
                for x, w, y in zip(xs, ws, ys):
                    kn.gemm(x, w, y, bias=b)

            def blas():
                for x, w, y in zip(xs, ws, ys):
                    torch.mm(x, w.t(), out=y)
            ours(); blas_ref = torch.mm(xs[0].float(), ws[0].float().t())
            err = float((ys[0].float() - blas_ref).abs().max() / blas_ref.abs().max())
            assert err < 1e-2, err
            t0, t1 = graph_time(ours), graph_time(blas)
            print("NT R=%5d N=%4d K=%4d   mmsum_gemm %6.1f us (%6.1f TF/s)   hipBLASLt %6.1f us   plan %s"
                  % (R, N, K, t0, 2.0 * R * N * K / t0 / 1e6, t1, kn.gemm_plan(xs[0], ws[0], ys[0], bias=b)), flush=True)
        for N, K in [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)]:
            dys = [torch.randn(R, N, device="cuda").to(dt) for _ in range(REPS)]
            xs = [torch.randn(R, K, device="cuda").to(dt) for _ in range(REPS)]
            outs = [torch.zeros(N, K, device="cuda") for _ in range(REPS)]
            for sk in sorted({1, 2, 4, splitk_rule(N, K, R)}):
              slab = torch.empty(max(sk, 1) * N, K, device="cuda")

              def ours():
                  for dy, x, o in zip(dys, xs, outs):
                      if sk > 1:
                          kn.gemm(dy, x, slab, a_t=True, b_t=True, splitk=sk, slabs=True)
                          kn.slab_reduce(slab, sk, o, accumulate=True)
                      else:
                          kn.gemm(dy, x, o, a_t=True, b_t=True, accumulate=True)

              for o in outs:
                  o.zero_()
              ours()
              ref = torch.mm(dys[0].float().t(), xs[0].float())
              err = float((outs[0] - ref).abs().max() / ref.abs().max())
              assert err < 2e-3, err
              t0 = graph_time(ours)
              tmp = [torch.empty(N, K, device="cuda", dtype=dt) for _ in range(REPS)]

              def blas2():
                  for dy, x, o in zip(dys, xs, tmp):
                      torch.mm(dy.t(), x, out=o)
              t1 = graph_time(blas2)
              print("TN R=%5d dW[%4d,%4d] split %2d%s  mmsum_gemm + slab_reduce %6.1f us (%6.1f TF/s)   hipBLASLt (bf16 out, no accumulate) %6.1f us"
                    % (R, N, K, sk, "*" if sk == splitk_rule(N, K, R) else " ", t0, 2.0 * R * N * K / t0 / 1e6, t1), flush=True)


if __name__ == "__main__":
    main()
