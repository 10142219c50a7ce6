#!/usr/bin/env python3
"""Runs ONE representative launch (a few repeats) of a row-streaming kernel family at the bench batch's sizes (B = 128: 147,456 decoder rows), for
rocprofv3 --pmc passes (tools/r6_family_pmc.sh).  usage: family_one.py <ln_fwd|ln_bwd|gate_fwd|gate_bwd|loss|adamw|bn_apply|bn_bwd|slab_reduce> [iters]
Prints the algorithmic bytes of one launch (operands once) as `ALG <name> <kernel substring> <bytes>`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bf = torch.bfloat16
R, D, V = 147456, 1024, 50265
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, dtype=bf: torch.randn(*s, device=dev, generator=g).to(dtype)      # noqa: E731

if name in ("ln_fwd", "ln_bwd"):
    x, res, y, dy, dx, dres = rnd(R, D), rnd(R, D), torch.empty(R, D, device=dev, dtype=bf), rnd(R, D), torch.empty(R, D, device=dev, dtype=bf), torch.empty(R, D, device=dev, dtype=bf)
    gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
    dg, db, dxs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    kn.add_ln_fwd(x, res, gamma, beta, y, mean, rstd, 1e-5, 0.1, 1234)
    if name == "ln_fwd":
        run = lambda: kn.add_ln_fwd(x, res, gamma, beta, y, mean, rstd, 1e-5, 0.1, 1234)      # noqa: E731
        alg, sub = 3 * R * D * 2 + 8 * R, "add_ln_fwd"
    else:
        run = lambda: kn.add_ln_bwd(dy, x, res, gamma, mean, rstd, dx, dres, False, dg, db, 0.1, 1234, dxsum=dxs)      # noqa: E731
        alg, sub = 5 * R * D * 2 + 8 * R, "add_ln_bwd"
elif name in ("gate_fwd", "gate_bwd"):
    pa, pb, yt, ytab, yimg, out, dout = (rnd(R, D) for _ in range(7))
    B = R // (9 * 128)
    no_table, no_img = torch.zeros(B, dtype=torch.uint8, device=dev), torch.zeros(B, dtype=torch.uint8, device=dev)
    dpa, dpb, dyt, dytab, dyimg = (torch.empty(R, D, device=dev, dtype=bf) for _ in range(5))
    s0, s1 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    if name == "gate_fwd":
        run = lambda: kn.gate_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, out, 9 * 128)      # noqa: E731
        alg, sub = 6 * R * D * 2, "gate_fwd"
    else:
        run = lambda: kn.gate_bwd(dout, pa, pb, ytab, yimg, no_table, no_img, dpa, dpb, dyt, dytab, dyimg, 9 * 128, sums=(s0, s1))      # noqa: E731
        alg, sub = 10 * R * D * 2, "gate_bwd"
elif name == "loss":
    ld = (V + 127) // 128 * 128
    logits = torch.empty(R, ld, device=dev, dtype=bf)
    for r0 in range(0, R, 16384):
        logits[r0:r0 + 16384].copy_(rnd(min(16384, R - r0), ld))
    keep = logits.clone()
    tgt = torch.randint(0, V, (R,), device=dev, generator=g)
    rl = torch.empty(R, device=dev)

    def run():
        logits.copy_(keep)
        kn.ls_loss(logits, tgt, rl, V, 0.1, 1.0 / R)
    alg, sub = 2 * R * V * 2, "ls_loss"
elif name == "adamw":
    n = 486_900_000 // 64 * 64
    p, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    sh = torch.zeros(n, device=dev, dtype=bf)
    hyper = torch.tensor([1e-5, 1e-7, 0.0, 0.0], device=dev)
    run = lambda: kn.adamw(p, gr, m, v, sh, hyper, None, 0.9, 0.999, 1e-6)      # noqa: E731
    alg, sub = n * 30, "adamw"
elif name in ("bn_apply", "bn_bwd"):
    Rb, C = 244 * 14 * 14, 1024                      # layer3's bn3 + residual at the ~244 images a B = 128 batch runs
    x, res, dy = rnd(Rb, C), rnd(Rb, C), rnd(Rb, C)
    y, dx, dres = (torch.empty(Rb, C, device=dev, dtype=bf) for _ in range(3))
    gamma, beta, rm, rv = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sums, dsums, dg, db = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    kn.bn_reduce(x, sums)
    kn.bn_apply(x, sums, gamma, beta, res, y, rm, rv, 1e-5, 0.1, True, True)
    kn.bn_bwd_reduce(dy, y, x, sums, dsums, 1e-5, True)
    if name == "bn_apply":
        run = lambda: kn.bn_apply(x, sums, gamma, beta, res, y, rm, rv, 1e-5, 0.1, True, True)      # noqa: E731
        alg, sub = 3 * Rb * C * 2, "bn_apply"
    else:
        run = lambda: kn.bn_bwd_apply(dy, y, x, sums, dsums, gamma, dx, dres, dg, db, 1e-5, True)      # noqa: E731
        alg, sub = 5 * Rb * C * 2, "bn_bwd_apply"
elif name == "slab_reduce":
    rows, cols, ns = 4096, 1024, 4
    ws = torch.randn(ns * rows, cols, device=dev, generator=g)
    out = torch.zeros(rows, cols, device=dev)
    run = lambda: kn.slab_reduce(ws, ns, out, accumulate=True)      # noqa: E731
    alg, sub = (ns + 2) * rows * cols * 4, "slab_reduce"
else:
    raise SystemExit("unknown family " + name)
for _ in range(iters):
    run()
torch.cuda.synchronize()
print("ALG %s %s %d" % (name, sub, alg), flush=True)
