#!/usr/bin/env python3
"""Micro-benchmark of the entity-attention kernels at the training-step shapes (B=8; ATTN_BENCH_B=28 for the bench batch).
bwd+ = backward that accumulates into dQ (second and third modality of the decoder's cross-attention)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn

NB = int(os.environ.get("ATTN_BENCH_B", "8"))
CASES = [("cross_text", NB, 9, 9, 128, 128, True, False), ("self_causal", 9 * NB, 1, 1, 128, 128, False, True),
         ("cross_img", NB, 9, 1, 196, 128, False, False), ("cross_table", NB, 9, 1, 47, 128, False, False),
         # the step's image memory: 4 image slots per business, U{0..4} of them filled (the others are null entities: skipped)
         ("cross_img4", NB, 9, 4, 196, 128, False, False)]


def timeit(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    H, D, dt = 16, 1024, torch.bfloat16
    for name, B, qpb, N, S, T, excl, causal in CASES:
        if only and only != name:
            continue
        nq = B * qpb
        q = torch.randn(nq * T, D, device="cuda").to(dt)
        kv = torch.randn(B * N * S, 2 * D, device="cuda").to(dt)
        k, v = kv[:, :D], kv[:, D:]
        out = torch.empty(nq * T, D, device="cuda", dtype=dt)
        pad = torch.zeros(B * N * S, dtype=torch.uint8, device="cuda")
        if os.environ.get("ATTN_BENCH_PADS") == "1" and name == "cross_text":      # review lengths ~ clamp(N(75,20),32,128): trailing pads
            g = torch.Generator().manual_seed(0)
            lens = (torch.randn(B * N, generator=g) * 20 + 75).round().clamp(32, S).long()
            pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8).reshape(-1).cuda()
        null = torch.zeros(B * N, dtype=torch.uint8, device="cuda")
        live_frac = 1.0
        if name == "cross_img4":
            g = torch.Generator().manual_seed(1)
            nv = torch.randint(0, N + 1, (B,), generator=g)
            gone = (torch.arange(N).unsqueeze(0) >= nv.unsqueeze(1))                       # [B, N]
            pad = gone.unsqueeze(-1).expand(B, N, S).reshape(-1).to(torch.uint8).cuda().contiguous()
            null = gone.reshape(-1).to(torch.uint8).cuda().contiguous()
            live_frac = float((~gone).float().mean())
        kv_rows = None
        if os.environ.get("ATTN_BENCH_MAPS") == "1" and not causal:     # compacted memory: K / V hold the unmasked rows only, read through a row map
            keep = pad.eq(0)
            nlive = int(keep.sum())
            kv_rows = torch.full((B * N * S,), -1, dtype=torch.int32, device="cuda")
            kv_rows[keep] = torch.arange(nlive, dtype=torch.int32, device="cuda")
            kv = kv[keep].contiguous()
            k, v = kv[:, :D], kv[:, D:]
        if os.environ.get("ATTN_BENCH_HOT") == "1" and kv_rows is not None:     # DIAGNOSTIC: every entity reads the same S rows (K / V always cache-hot)
            kv_rows = (torch.arange(B * N * S, device="cuda") % S).to(torch.int32)
            kv_rows[pad.ne(0)] = -1
        desc = kn.make_attn_desc(q, k, v, out, pad, null, nq, T, qpb, N, S, H, excl, causal, 0.125, kv_rows=kv_rows)
        dout = torch.randn(nq * T, D, device="cuda").to(dt)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device="cuda")
        ents = (N - 1) if excl else N
        fl = 4.0 * nq * T * S * 64 * H * ents
        tf = timeit(lambda: kn.attn_fwd(desc, q))
        tb = timeit(lambda: kn.attn_bwd(desc, dout, dq, False, dkv[:, :D], dkv[:, D:], stats))
        ta = timeit(lambda: kn.attn_bwd(desc, dout, dq, True, dkv[:, :D], dkv[:, D:], stats))
        print("%-12s fwd %7.0f us (%6.1f TF/s)   bwd %7.0f us (%6.1f TF/s)   bwd+ %7.0f us%s" % (name, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6, ta,
              "   (%.0f %% of the entities live: TF/s at the canonical count of ALL entities)" % (100 * live_frac) if live_frac < 1 else ""), flush=True)


if __name__ == "__main__":
    main()
