#!/usr/bin/env python3
"""The decode step's kernels one by one, inside a captured HIP graph (the way the step issues them), old against new:
  products: mmsum_gemm's weight-streaming kernels (K split over the waves of a 16- / 32-column workgroup) against mmsum_dec_gemm (K
            split over one-wave workgroups, last-arriver reduction);
  cross-attention: three mmsum_attn_fwd launches (the training kernel, one workgroup per business and head) against one
            mmsum_decode_cross_attn launch (one workgroup per entity and head).
usage: python tools/decode_kernels_bench.py [rows = businesses x beams, default 32]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16


def graph_time(run, reps):
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
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


cases = [("qkv", M, 3072, 1024, 0, 0), ("proj", M, 1024, 1024, 0, 0), ("cross out (3 modalities)", 3 * M, 1024, 1024, 0, 0),
         ("alpha/beta (two tensors)", M, 1024, 2048, 0, 1024), ("fc1+gelu", M, 4096, 1024, kn.EPI_GELU, 0), ("fc2", M, 1024, 4096, 0, 0)]
ws = kn.dec_gemm_workspace(96, 4096, 4096, "cuda")
for name, m, N, K, epi, k2 in cases:
    reps = 20
    xs = [torch.randn(m, K, device="cuda").to(dt) for _ in range(reps)]
    wts = [(torch.randn(N, K, device="cuda") * 0.02).to(dt) for _ in range(reps)]          # distinct weights per launch: nothing stays in cache
    b = torch.zeros(N, device="cuda")
    ys = [torch.empty(m, N, device="cuda", dtype=dt) for _ in range(reps)]

    def old():
        for x, w, y in zip(xs, wts, ys):
            kn.gemm(x[:, :k2] if k2 else x, w, y, bias=b, epi=epi, a2=x[:, k2:] if k2 else None)

    def new():
        for x, w, y in zip(xs, wts, ys):
            kn.dec_gemm(x[:, :k2] if k2 else x, w, y, ws, bias=b, epi=epi, x2=x[:, k2:] if k2 else None)
    t_old, t_new = graph_time(old, reps), graph_time(new, reps)
    print("%-26s M=%3d N=%5d K=%4d   mmsum_gemm %6.1f us (%5.2f TB/s of weights)   mmsum_dec_gemm %6.1f us (%5.2f TB/s)"
          % (name, m, N, K, t_old, N * K * 2 / t_old / 1e6, t_new, N * K * 2 / t_new / 1e6), flush=True)

# ---- the LM head: f32 rows (the un-rounded last LayerNorm) x bf16 weights [50265, 1024] -> f32 logits
V_ = 50265
reps = 4
x32s = [torch.randn(M, 1024, device="cuda") for _ in range(reps)]
wvs = [(torch.randn(V_, 1024, device="cuda") * 0.02).to(dt) for _ in range(reps)]
lgs = [torch.empty(M, 50304, device="cuda") for _ in range(reps)]
bz = torch.zeros(V_, device="cuda")
wsv = kn.dec_gemm_workspace(96, V_, 1024, "cuda")


def lm_old():
    for x, w, y in zip(x32s, wvs, lgs):
        for r0 in range(0, M, 64):
            kn.gemm(x[r0:r0 + 64], w, y[r0:r0 + 64, :V_], bias=bz)


def lm_new():
    for x, w, y in zip(x32s, wvs, lgs):
        kn.dec_gemm(x, w, y[:, :V_], wsv, bias=bz)


lm_old(); r0_ = lgs[0].clone(); lm_new(); torch.cuda.synchronize()
print("LM head outputs: max |old - new| = %.3e" % float((r0_[:, :V_] - lgs[0][:, :V_]).abs().max()))
t_old, t_new = graph_time(lm_old, reps), graph_time(lm_new, reps)
print("LM head M=%3d N=%5d K=1024 (f32 rows)   mmsum_gemm %6.1f us (%5.2f TB/s of weights)   mmsum_dec_gemm %6.1f us (%5.2f TB/s)"
      % (M, V_, t_old, V_ * 1024 * 2 / t_old / 1e6, t_new, V_ * 1024 * 2 / t_new / 1e6), flush=True)

# ---- cross-attention over the cached K / V: B businesses x qpb hypotheses, text 8 x 128 (trailing pads), table 1 x 47, images 4 x 196 (U{0..4} live)
B, qpb, H, D = max(1, M // 4), 4, 16, 1024
R = B * qpb
mods_shape = [(8, 128), (1, 47), (4, 196)]
g = torch.Generator().manual_seed(0)
reps = 6
rows = sum(B * N * S for N, S in mods_shape)
kvs = [torch.randn(rows, 2 * D, device="cuda").to(dt) for _ in range(reps)]
q = torch.randn(R, D, device="cuda").to(dt)
pads, nulls = [], []
for mi, (N, S) in enumerate(mods_shape):
    pad = torch.zeros(B, N, S, dtype=torch.bool)
    if mi == 0:
        lens = (torch.randn(B, N, generator=g) * 20 + 75).round().clamp(32, S).long()
        pad = torch.arange(S).view(1, 1, S) >= lens.unsqueeze(-1)
    if mi == 2:
        nv = torch.randint(0, N + 1, (B,), generator=g)
        pad = (torch.arange(N).unsqueeze(0) >= nv.unsqueeze(1)).unsqueeze(-1).expand(B, N, S).clone()
    pu = pad.to(torch.uint8).cuda().contiguous()
    nul = torch.empty(B * N, dtype=torch.uint8, device="cuda")
    kn.entity_null(pu, nul, B * N, S)
    pads.append(pu)
    nulls.append(nul)
heads = torch.empty(3 * R, D, device="cuda", dtype=dt)
xws = kn.decode_cross_attn_workspace(sum(B * N for N, S in mods_shape), H, qpb, B, 3, "cuda")


def mods_of(kv):
    out, off = [], 0
    for (N, S), pu, nul in zip(mods_shape, pads, nulls):
        sl = slice(off, off + B * N * S)
        out.append((kv[sl, :D], kv[sl, D:], pu, nul, N, S))
        off += B * N * S
    return out


def old_x():
    for kv in kvs:
        for m, (k, v, pu, nul, N, S) in enumerate(mods_of(kv)):
            d = kn.make_attn_desc(q, k, v, heads[m * R:(m + 1) * R], pu, nul, B, qpb, 1, N, S, H, False, False, 0.125)
            kn.attn_fwd(d, q)


def new_x():
    for kv in kvs:
        kn.decode_cross_attn(q, mods_of(kv), heads, xws, B, qpb, H, 0.125)


old_x()
ref = heads.clone()
new_x()
torch.cuda.synchronize()
print("cross-attention outputs: max |old - new| = %.3e (max |old| %.3e)" % (float((ref.float() - heads.float()).abs().max()), float(ref.float().abs().max())))
nbytes = rows * 2 * D * 2
t_old, t_new = graph_time(old_x, reps), graph_time(new_x, reps)
print("cross-attention B=%d x %d hypotheses, %.1f MB of cached K / V:   3 x mmsum_attn_fwd %6.1f us (%5.2f TB/s)   mmsum_decode_cross_attn %6.1f us (%5.2f TB/s)"
      % (B, qpb, nbytes / 1e6, t_old, nbytes / t_old / 1e6, t_new, nbytes / t_new / 1e6))
