"""GPU: every C-ABI kernel against a plain PyTorch fp32 statement of the same op (and the oracle /
golden vectors where one exists).  f32 instantiations are held to ~1e-4, bf16 ones to bf16 rounding."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multimodalsum_amd import kernels as kn
    from multimodalsum_amd import _lib

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return (2e-4, 2e-4) if dtype == torch.float32 else (3e-2, 3e-2)


def close(a, b, dtype, scale=1.0, what=""):
    rt, at = tol(dtype)
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert math.isfinite(err), "%s: non-finite" % what
    assert err <= scale * (at + rt * ref), "%s: max err %.3e (ref max %.3e)" % (what, err, ref)


def rnd(*shape, dtype=torch.float32, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(DEV).to(dtype)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (128, 128, 64), (47, 1024, 2048), (300, 72, 160), (200, 1024, 26), (64, 72, 77), (33, 40, 8)])
def test_gemm_layouts(dtype, a_t, b_t, M, N, K):
    if a_t and M % 8:
        M = M + (8 - M % 8)      # a transposed operand needs an aligned leading dimension (all call sites have it)
    if K % 8 and not (a_t and b_t):
        pytest.skip("K-contiguous operands are read in 16-byte chunks: ragged K exists only as a reduction over rows (both transposed)")
    a = rnd(M, K, dtype=dtype, seed=1)
    b = rnd(N, K, dtype=dtype, seed=2)
    ref = a.float() @ b.float().t()
    A = a.t().contiguous() if a_t else a
    B = b.t().contiguous() if b_t else b
    out = torch.full((M, N), float("nan"), device=DEV, dtype=dtype)
    kn.gemm(A, B, out, a_t=a_t, b_t=b_t)
    close(out, ref, dtype, scale=math.sqrt(K / 64), what="gemm")


@pytest.mark.parametrize("M,N,K", [(32, 1024, 1024), (12, 3072, 1024), (64, 1024, 4096), (33, 4096, 1024), (1, 256, 128), (96, 1024, 1024), (128, 2048, 1024)])
def test_gemm_skinny_decode_shapes(M, N, K):
    """gemm_skinny_kernel (M <= 128 bf16: the decode step's weight-streaming products): bias, GELU, K split over two operands."""
    dt = torch.bfloat16
    x, w = rnd(M, K, dtype=dt, seed=50, std=0.5), rnd(N, K, dtype=dt, seed=51, std=0.5)
    bias = rnd(N, seed=52)
    ref = x.float() @ w.float().t() + bias
    out = torch.full((M, N), float("nan"), device=DEV, dtype=dt)
    kn.gemm(x, w, out, bias=bias)
    close(out, ref, dt, scale=math.sqrt(K / 64), what="skinny bias")
    kn.gemm(x, w, out, bias=bias, epi=kn.EPI_GELU, alpha=0.125)
    close(out, F.gelu((x.float() @ w.float().t()) * 0.125 + bias), dt, scale=math.sqrt(K / 64), what="skinny gelu")
    if K >= 256:
        x2 = rnd(M, 128, dtype=dt, seed=53, std=0.5)
        w2 = rnd(N, K + 128, dtype=dt, seed=54, std=0.5)
        kn.gemm(x, w2, out, a2=x2, bias=bias)
        close(out, torch.cat([x.float(), x2.float()], 1) @ w2.float().t() + bias, dt, scale=math.sqrt(K / 64), what="skinny a2")


@pytest.mark.parametrize("M,N,K", [(32, 1024, 1024), (12, 3072, 1024), (64, 1024, 4096), (33, 4096, 1024), (1, 256, 1024), (32, 50265, 1024), (40, 2048, 2048), (96, 1024, 1024)])
def test_gemm_skinny_f32_decode_shapes(M, N, K):
    """gemm_skinny_f32_kernel (f32 x, W, out, M <= 96: the decode step of the f32 compute mode, whose token ids are held to the
    reference's): bias, erf-GELU, K split over two operands, against an f64 product; the plan names the weight-streaming kernel."""
    from multimodalsum_amd import _lib
    dt = torch.float32
    x, w = rnd(M, K, seed=60, std=0.5), rnd(N, K, seed=61, std=0.5)
    bias = rnd(N, seed=62)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=dt)
    assert kn.gemm_plan(x, w, out, bias=bias)[0] == _lib.PLAN_SKINNY
    kn.gemm(x, w, out, bias=bias)
    close(out, x.double() @ w.double().t() + bias.double(), dt, scale=math.sqrt(K / 64), what="skinny f32 bias")
    kn.gemm(x, w, out, bias=bias, epi=kn.EPI_GELU, alpha=0.125)
    close(out, F.gelu((x.double() @ w.double().t()) * 0.125 + bias.double()), dt, scale=math.sqrt(K / 64), what="skinny f32 gelu")
    if K == 1024:                      # the alpha / beta products: [yt | ytab] against a [N, 2 K] weight
        x2 = rnd(M, K, seed=63, std=0.5)
        w2 = rnd(N, 2 * K, seed=64, std=0.5)
        assert kn.gemm_plan(x, w2, out, a2=x2, bias=bias)[0] == _lib.PLAN_SKINNY
        kn.gemm(x, w2, out, a2=x2, bias=bias)
        close(out, torch.cat([x.double(), x2.double()], 1) @ w2.double().t() + bias.double(), dt, scale=math.sqrt(K / 32), what="skinny f32 a2")
    # a strided view of the output (the logits buffer is padded to 128 columns) and rows the kernel must not touch
    big = torch.full((M + 3, N + 7), float("nan"), device=DEV, dtype=dt)
    kn.gemm(x, w, big[:M, :N], bias=bias)
    close(big[:M, :N], x.double() @ w.double().t() + bias.double(), dt, scale=math.sqrt(K / 64), what="skinny f32 strided")
    assert torch.isnan(big[M:]).all() and torch.isnan(big[:, N:]).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dtype):
    M, N, Kd = 160, 200, 128
    a, b = rnd(M, Kd, dtype=dtype, seed=3, std=0.3), rnd(N, Kd, dtype=dtype, seed=4, std=0.3)
    bias = rnd(N, seed=5)
    pre = a.float() @ b.float().t() * 0.5 + bias
    # bias + alpha + GELU with aux
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    aux = torch.empty(M, N, device=DEV, dtype=dtype)
    kn.gemm(a, b, out, bias=bias, alpha=0.5, epi=kn.EPI_GELU, aux=aux)
    close(aux, pre, dtype, what="gelu aux")
    close(out, F.gelu(pre), dtype, what="gelu out")
    # GELU backward: v * gelu'(aux)
    u = rnd(M, N, dtype=dtype, seed=6)
    uf = u.float().requires_grad_(True)
    F.gelu(uf).backward(a.float() @ b.float().t())
    out2 = torch.empty(M, N, device=DEV, dtype=dtype)
    kn.gemm(a, b, out2, epi=kn.EPI_GELU_BWD, aux=u)
    close(out2, uf.grad, dtype, what="gelu bwd")
    # ReLU fwd / bwd
    kn.gemm(a, b, out, bias=bias, epi=kn.EPI_RELU)
    close(out, F.relu(a.float() @ b.float().t() + bias), dtype, what="relu")
    kn.gemm(a, b, out2, epi=kn.EPI_RELU_BWD, aux=out)
    close(out2, (a.float() @ b.float().t()) * (out.float() > 0), dtype, what="relu bwd")
    # accumulate into f32 output (+ split-K atomics)
    acc = rnd(M, N, seed=7)
    ref = acc + a.float() @ b.float().t()
    o1 = acc.clone()
    kn.gemm(a, b, o1, accumulate=True)
    close(o1, ref, dtype, what="accum f32")
    o2 = acc.clone()
    kn.gemm(a, b, o2, accumulate=True, splitk=2)
    close(o2, ref, dtype, what="splitk")
    # accumulate into an output of the operand dtype; ragged N (unaligned rows: scalar stores) with bias
    o6 = rnd(M, N, dtype=dtype, seed=11)
    ref6 = o6.float() + a.float() @ b.float().t()
    kn.gemm(a, b, o6, accumulate=True)
    close(o6, ref6, dtype, what="accum same dtype")
    for Nr in (203, 264):
        br, biasr = rnd(Nr, Kd, dtype=dtype, seed=12, std=0.3), rnd(Nr, seed=13)
        o7 = rnd(300, Nr, dtype=dtype, seed=14)
        ar = rnd(300, Kd, dtype=dtype, seed=15, std=0.3)
        ref7 = o7.float() + ar.float() @ br.float().t() + biasr
        kn.gemm(ar, br, o7, bias=biasr, accumulate=True)
        close(o7, ref7, dtype, what="accum + bias, N=%d" % Nr)
        o8 = rnd(300, Nr, seed=16)
        ref8 = o8 + ar.float() @ br.float().t() + biasr
        kn.gemm(ar, br, o8, bias=biasr, accumulate=True)
        close(o8, ref8, dtype, what="f32 accum + bias, N=%d" % Nr)
    # split-K into partial slabs + deterministic reduce (no atomics)
    ws = torch.full((3 * M, N), float("nan"), device=DEV)
    kn.gemm(a, b, ws, splitk=3, slabs=True)
    o5 = acc.clone()
    kn.slab_reduce(ws, 3, o5, accumulate=True)
    close(o5, ref, dtype, what="splitk slabs")
    # epilogue column sums (bias gradient of the layer below) on the bf16 fast path; rejected elsewhere
    if dtype == torch.bfloat16:
        for (Mc, Nc) in ((160, 200), (1000, 520)):
            ac, bc = rnd(Mc, Kd, dtype=dtype, seed=30, std=0.3), rnd(Nc, Kd, dtype=dtype, seed=31, std=0.3)
            uc = rnd(Mc, Nc, dtype=dtype, seed=32)
            oc = torch.empty(Mc, Nc, device=DEV, dtype=dtype)
            cs = torch.ones(Nc, device=DEV)
            kn.gemm(ac, bc, oc, epi=kn.EPI_GELU_BWD, aux=uc, colsum=cs)
            ref_o = torch.empty(Mc, Nc, device=DEV, dtype=dtype)
            kn.gemm(ac, bc, ref_o, epi=kn.EPI_GELU_BWD, aux=uc)
            assert torch.equal(oc, ref_o)
            refc = 1.0 + ref_o.double().sum(0)
            assert (cs.double() - refc).abs().max().item() <= 1e-4 * (1 + refc.abs().max().item())
    else:
        with pytest.raises(AssertionError):
            kn.gemm(a, b, torch.empty(M, N, device=DEV, dtype=dtype), colsum=torch.zeros(N, device=DEV))
    # K split over two A operands
    a2 = rnd(M, 64, dtype=dtype, seed=8, std=0.3)
    b2 = rnd(N, Kd + 64, dtype=dtype, seed=9, std=0.3)
    o3 = torch.empty(M, N, device=DEV, dtype=dtype)
    kn.gemm(a, b2, o3, a2=a2)
    close(o3, torch.cat([a.float(), a2.float()], 1) @ b2.float().t(), dtype, what="a2 split")
    # strided views (sub-matrix of a wider buffer)
    wide = rnd(M, 3 * Kd, dtype=dtype, seed=10, std=0.3)
    o4 = torch.empty(M, N, device=DEV, dtype=dtype)
    kn.gemm(wide[:, Kd:2 * Kd], b, o4)
    close(o4, wide[:, Kd:2 * Kd].float() @ b.float().t(), dtype, what="strided A")


@pytest.mark.parametrize("rows,n_out,k_in,sk", [(512, 256, 256, 1), (1000, 1024, 512, 4), (2066, 2048, 1024, 8), (96, 136, 72, 1),
                                               (16128, 1024, 1024, 8), (18, 256, 128, 1), (4000, 3072, 1024, 3)])
def test_gemm_wgrad_layout(rows, n_out, k_in, sk):
    """dW[n_out,k_in] = dy[rows,n_out]^T x[rows,k_in] straight from reduction-major bf16 operands
    (gemm_tn_ring_kernel): bf16 result, f32 accumulate, atomics and split-K slabs; any row count."""
    dy = rnd(rows, n_out, dtype=torch.bfloat16, seed=11, std=0.5)
    x = rnd(rows, k_in, dtype=torch.bfloat16, seed=12, std=0.5)
    ref = dy.double().t() @ x.double()
    scale = math.sqrt(max(rows, 64) / 64)
    acc = rnd(n_out, k_in, seed=13)
    o1 = acc.clone()
    kn.gemm(dy, x, o1, a_t=True, b_t=True, accumulate=True)
    close(o1, acc.double() + ref, torch.float32, scale=4 * scale, what="tn accumulate")
    o2 = torch.full((n_out, k_in), float("nan"), device=DEV, dtype=torch.bfloat16)
    kn.gemm(dy, x, o2, a_t=True, b_t=True)
    close(o2, ref, torch.bfloat16, scale=scale, what="tn bf16 out")
    if sk > 1:
        ws = torch.full((sk * n_out, k_in), float("nan"), device=DEV)
        kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
        o3 = acc.clone()
        kn.slab_reduce(ws, sk, o3, accumulate=True)
        close(o3, acc.double() + ref, torch.float32, scale=4 * scale, what="tn slabs")
        o4 = acc.clone()
        kn.gemm(dy, x, o4, a_t=True, b_t=True, accumulate=True, splitk=sk)
        close(o4, acc.double() + ref, torch.float32, scale=4 * scale, what="tn atomics")
    # strided operands (column windows of wider buffers, as the fused q/k/v gradients are)
    wide = rnd(rows, n_out + 64, dtype=torch.bfloat16, seed=14, std=0.5)
    o5 = torch.zeros(n_out, k_in, device=DEV)
    kn.gemm(wide[:, 64:], x, o5, a_t=True, b_t=True, accumulate=True)
    close(o5, wide[:, 64:].double().t() @ x.double(), torch.float32, scale=4 * scale, what="tn strided")
    # ragged output rows inside a padded leading dimension (the tied-embedding gradient: V = 50265 of 50304 columns)
    if n_out > 8:
        o6 = torch.zeros(n_out - 3, k_in, device=DEV)
        kn.gemm(dy[:, :n_out - 3], x, o6, a_t=True, b_t=True, accumulate=True)
        close(o6, ref[:n_out - 3], torch.float32, scale=4 * scale, what="tn ragged M")


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum(dtype):
    x = rnd(1000, 200, dtype=dtype, seed=1)
    out = torch.ones(200, device=DEV)
    kn.colsum(x, out, accumulate=True)
    close(out, 1 + x.float().sum(0), dtype, scale=4, what="colsum")


@pytest.mark.parametrize("rows,cols,ld_extra", [(64, 64, 0), (200, 136, 0), (333, 1024, 0), (130, 72, 8), (1000, 3072, 0), (77, 40, 0)])
def test_transpose_colsum(rows, cols, ld_extra):
    """mmsum_transpose_bf16: bit-exact transpose, zero padding up to rows_pad, fused column sums (+=)."""
    src_full = rnd(rows, cols + ld_extra, dtype=torch.bfloat16, seed=rows + cols)
    src = src_full[:, :cols]
    rp = (rows + 63) // 64 * 64
    dst = torch.full((cols, rp), 7.0, device=DEV, dtype=torch.bfloat16)
    kn.transpose(src, dst, rp)
    assert torch.equal(dst[:, :rows], src.t())
    assert (dst[:, rows:] == 0).all()
    dst2 = torch.full((cols, rp), 7.0, device=DEV, dtype=torch.bfloat16)
    cs = torch.ones(cols, device=DEV, dtype=torch.float32)
    kn.transpose(src, dst2, rp, colsum=cs)
    assert torch.equal(dst2, dst)
    ref = 1.0 + src.double().sum(0)
    assert (cs.double() - ref).abs().max().item() <= 1e-4 * (1 + ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [256, 1024])
def test_add_ln(dtype, D):
    R = 77
    x, res, dy = rnd(R, D, dtype=dtype, seed=1), rnd(R, D, dtype=dtype, seed=2), rnd(R, D, dtype=dtype, seed=3)
    gamma, beta = (1 + 0.1 * rnd(D, seed=4)), 0.1 * rnd(D, seed=5)
    xf, rf = x.float().requires_grad_(True), res.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xf + rf, (D,), gf, bf, 1e-5)
    ref.backward(dy.float())
    y = torch.empty_like(x)
    mean, rstd = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    kn.add_ln_fwd(x, res, gamma, beta, y, mean, rstd, 1e-5, 0.0, 0)
    close(y, ref, dtype, what="ln fwd")
    dx, dres = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    kn.add_ln_bwd(dy, x, res, gamma, mean, rstd, dx, dres, False, dg, db, 0.0, 0)
    close(dx, xf.grad, dtype, what="ln dx")
    close(dres, rf.grad, dtype, what="ln dres")
    close(dg, gf.grad, dtype, scale=4, what="ln dgamma")
    close(db, bf.grad, dtype, scale=4, what="ln dbeta")
    # fused column sums of dx (the bias gradient of the Linear that produced x), with dropout
    dxs = torch.ones(D, device=DEV)
    kn.add_ln_bwd(dy, x, res, gamma, mean, rstd, dx, dres, False, dg, db, 0.25, 77, dxsum=dxs)
    close(dxs, 1.0 + dx.double().sum(0), dtype, scale=4, what="ln dx column sums")
    kn.add_ln_fwd(x, res, gamma, beta, y, mean, rstd, 1e-5, 0.0, 0)
    # dropout: statistical + fwd/bwd mask consistency
    kn.add_ln_fwd(x, torch.zeros_like(res), torch.ones_like(gamma), torch.zeros_like(beta), y, mean, rstd, 1e-5, 0.5, 1234)
    kn.add_ln_bwd(dy, x, torch.zeros_like(res), torch.ones_like(gamma), mean, rstd, dx, dres, False, dg, db, 0.5, 1234)
    frac_zero = (dx.float() == 0).float().mean().item()
    assert 0.4 < frac_zero < 0.6, frac_zero


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_ln(dtype):
    V, D, nseq, T = 50, 256, 6, 10
    E, P = rnd(V, D, dtype=dtype, seed=1), rnd(T + 2, D, dtype=dtype, seed=2)
    rvec = rnd(D, dtype=dtype, seed=3)
    rd = rnd(nseq, seed=4)
    gamma, beta = (1 + 0.1 * rnd(D, seed=5)), 0.1 * rnd(D, seed=6)
    ids = torch.randint(0, V, (nseq, T), generator=torch.Generator().manual_seed(7)).to(DEV)
    ids[0, 3] = 1
    ids[2, 5:] = 1
    dy = rnd(nseq * T, D, dtype=dtype, seed=8)
    Ef, Pf, rf = E.float().requires_grad_(True), P.float().requires_grad_(True), rvec.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.embedding(ids, Ef, padding_idx=1) + Pf[torch.arange(T, device=DEV) + 2] + (rd[:, None] * rf)[:, None, :]
    ref = F.layer_norm(z, (D,), gf, bf, 1e-5).view(-1, D)
    ref.backward(dy.float())
    y = torch.empty(nseq * T, D, device=DEV, dtype=dtype)
    mean, rstd = torch.empty(nseq * T, device=DEV), torch.empty(nseq * T, device=DEV)
    kn.embed_ln_fwd(ids, E, P, rd, rvec, gamma, beta, y, mean, rstd, nseq, T, 2, 1e-5, 0.0, 0)
    close(y, ref, dtype, what="embed fwd")
    dE, dP = torch.zeros(V, D, device=DEV), torch.zeros(T + 2, D, device=DEV)
    dr, dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    kn.embed_ln_bwd(dy, ids, E, P, rd, rvec, gamma, mean, rstd, dE, dP, dr, dg, db, nseq, T, 2, 1, 0.0, 0)
    close(dE, Ef.grad, dtype, scale=4, what="dE")
    close(dP, Pf.grad, dtype, scale=4, what="dP")
    close(dr, rf.grad, dtype, scale=4, what="drvec")
    close(dg, gf.grad, dtype, scale=4, what="dgamma")
    close(db, bf.grad, dtype, scale=4, what="dbeta")
    assert dE[1].abs().max().item() == 0.0  # padding_idx row gets no lookup gradient


# ------------------------------------------------------------------------------------------------
def attn_reference(q, k, v, pad, nq, T, qpb, N, S, H, exclude, causal, scale, mm=None, q0=0):
    """q [nq*T, H*64]; k/v [B*N*S, H*64]; pad [B,N,S] bool.  Returns out [nq*T, H*64] (autograd, in the dtype of q).
    mm: the two attention products (default torch.matmul; the bf16 yardstick passes the oracle's bf16-operand product)."""
    mm = mm or torch.matmul
    B = nq // qpb
    qh = q.view(nq, T, H, 64).permute(0, 2, 1, 3)                       # [nq,H,T,64]
    kh = k.view(B, N, S, H, 64).permute(0, 1, 3, 2, 4)                   # [B,N,H,S,64]
    vh = v.view(B, N, S, H, 64).permute(0, 1, 3, 2, 4)
    outs = []
    for qb in range(nq):
        b, i = qb // qpb, qb % qpb
        acc, cnt = 0, 0
        for n in range(N):
            if exclude and n == i:
                continue
            if pad is not None and bool(pad[b, n].all()):
                continue
            s = mm(qh[qb], kh[b, n].transpose(-1, -2)) * scale
            if pad is not None:
                s = s.masked_fill(pad[b, n][None, None, :], float("-inf"))
            if causal:
                s = s + torch.triu(torch.full((T, S), float("-inf"), device=s.device, dtype=s.dtype), 1 + q0)     # key s masked when s > q0 + t
            acc = acc + mm(torch.softmax(s, -1), vh[b, n])
            cnt += 1
        if cnt == 0:
            outs.append(torch.zeros(H, T, 64, device=q.device, dtype=q.dtype) + 0 * qh[qb])
        else:
            outs.append(acc / cnt)
    return torch.stack(outs).permute(0, 2, 1, 3).reshape(nq * T, H * 64)


ATTN_CASES = [
    # name, B, qpb, N, S, T, H, exclude, causal, self
    ("enc_self", 3, 1, 1, 40, 40, 2, False, False, True),
    ("dec_self", 2, 1, 1, 128, 128, 2, False, True, True),
    ("dec_self_short", 2, 1, 1, 10, 10, 1, False, True, True),
    ("cross_text_loo", 2, 3, 3, 20, 12, 2, True, False, False),
    ("cross_text_full", 1, 9, 9, 128, 128, 1, True, False, False),
    ("cross_table", 2, 3, 1, 47, 12, 2, False, False, False),
    ("cross_img", 2, 2, 3, 196, 33, 1, False, False, False),
    ("cross_text_holes", 2, 3, 3, 128, 70, 2, True, False, False),      # masked keys anywhere, not only trailing padding
    ("cross_img_holes", 1, 2, 2, 196, 128, 1, False, False, False),
    ("cross_img_one_entity", 2, 9, 1, 196, 128, 1, False, False, False),   # the training layout: one image / table entity per business,
    ("cross_table_walk", 3, 4, 1, 47, 40, 2, False, False, False),         # shared by its query blocks (a workgroup walks several of them)
    # review lengths 5 .. 128: one, two, three and four live key blocks per entity -- every way the dK / dV kernel shares a live block's
    # query sweep with the waves whose own block is all padding (three helpers; two; one that moves from block to block; none)
    ("cross_text_lengths", 1, 9, 9, 128, 64, 1, True, False, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(dtype, case):
    name, B, qpb, N, S, T, H, exclude, causal, is_self = case
    nq = B * qpb
    D = H * 64
    g = torch.Generator().manual_seed(11)
    pad = torch.zeros(B, N, S, dtype=torch.bool)
    for b in range(B):
        for n in range(N):
            L = int(torch.randint(max(1, S // 3), S + 1, (1,), generator=g))
            pad[b, n, L:] = True
    if name == "cross_text_lengths":
        for n, L in enumerate((5, 32, 33, 64, 65, 96, 97, 128, 20)):
            pad[0, n] = torch.arange(S) >= L
    if name.endswith("_holes"):
        pad = pad | (torch.rand(B, N, S, generator=g) < 0.3)
        pad[:, :, 5] = False
        pad[0, 0, :64] = False              # first masked key of this entity lies in its third key block
    if not is_self and N > 1 and name != "cross_text_lengths":
        pad[0, N - 1, :] = True            # a null entity
    if name in ("cross_table", "cross_table_walk"):
        pad[1, 0, :] = True                 # business without a table: output must be exactly 0
    if causal:
        pad[:, :, 0] = False
    pad = pad.to(DEV)
    if is_self:
        qkv = rnd(nq * T, 3 * D, dtype=dtype, seed=1)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q = rnd(nq * T, D, dtype=dtype, seed=1)
        kv = rnd(B * N * S, 2 * D, dtype=dtype, seed=2)
        k, v = kv[:, :D], kv[:, D:]
    dout = rnd(nq * T, D, dtype=dtype, seed=3)
    scale = 0.125
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
    ref = attn_reference(qf, kf, vf, pad, nq, T, qpb, N, S, H, exclude, causal, scale)
    ref.backward(dout.float())

    pad_u8 = pad.to(torch.uint8).contiguous()
    null = torch.empty(B * N, dtype=torch.uint8, device=DEV)
    kn.entity_null(pad_u8, null, B * N, S)
    assert torch.equal(null.bool().cpu(), pad.view(B * N, S).all(-1).cpu())
    out = torch.full((nq * T, D), float("nan"), device=DEV, dtype=dtype)
    desc = kn.make_attn_desc(q, k, v, out, pad_u8, null, nq, T, qpb, N, S, H, exclude, causal, scale)
    kn.attn_fwd(desc, q)
    close(out, ref, dtype, what=name + " fwd")

    dq = torch.full((nq * T, D), float("nan"), device=DEV, dtype=dtype)
    dk = torch.full((B * N * S, D), float("nan"), device=DEV, dtype=dtype)
    dv = torch.full((B * N * S, D), float("nan"), device=DEV, dtype=dtype)
    stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device=DEV)
    kn.attn_bwd(desc, dout, dq, False, dk, dv, stats)
    close(dq, qf.grad, dtype, what=name + " dq")
    close(dk, kf.grad, dtype, what=name + " dk")
    close(dv, vf.grad, dtype, what=name + " dv")

    # accumulation into dQ (second and third modality of the decoder's cross-attention), and outputs whose row pitch is not
    # a multiple of 16 bytes (the kernels then leave the 16-byte row path): same values
    base = rnd(nq * T, D, dtype=dtype, seed=4)
    dq2 = base.clone()
    kn.attn_bwd(desc, dout, dq2, True, dk, dv, stats)
    close(dq2, base.float() + qf.grad, dtype, what=name + " dq accumulate")
    odd = lambda rows: torch.full((rows, D + 2), float("nan"), device=DEV, dtype=dtype)[:, :D]      # noqa: E731
    out_o, dq_o, dk_o, dv_o = odd(nq * T), odd(nq * T), odd(B * N * S), odd(B * N * S)
    desc_o = kn.make_attn_desc(q, k, v, out_o, pad_u8, null, nq, T, qpb, N, S, H, exclude, causal, scale)
    kn.attn_fwd(desc_o, q)
    assert torch.equal(out_o, out)
    dq_o.copy_(base)
    kn.attn_bwd(desc_o, dout, dq_o, True, dk_o, dv_o, stats)
    assert torch.equal(dq_o, dq2) and torch.equal(dk_o, dk) and torch.equal(dv_o, dv)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,T,q0", [(158, 30, 128), (224, 96, 128), (129, 1, 128), (141, 13, 128), (96, 32, 64), (200, 128, 64)])
def test_attention_causal_query_offset(dtype, S, T, q0):
    """mmsum_attn_desc.causal_q0: a causal query block whose first row sits at key q0 (the second query block of a decoder sequence of
    129 .. 224 positions, engine._causal_long_fwd): key s is masked for query row t when s > q0 + t.  Forward, dQ and dK / dV (every key
    row, the ones no query sees included: zeros) against the fp32 reference, with trailing key padding; bad offsets are refused."""
    B, H = 3, 2
    D = H * 64
    pad = torch.zeros(B, 1, S, dtype=torch.bool)
    pad[1, 0, S - 7:] = True
    pad[2, 0, q0 + max(1, T // 2):] = True          # the padding starts inside the diagonal region
    pad = pad.to(DEV)
    q = rnd(B * T, D, dtype=dtype, seed=1)
    kv = rnd(B * S, 2 * D, dtype=dtype, seed=2)
    k, v = kv[:, :D], kv[:, D:]
    dout = rnd(B * T, D, dtype=dtype, seed=3)
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
    ref = attn_reference(qf, kf, vf, pad, B, T, 1, 1, S, H, False, True, 0.125, q0=q0)
    ref.backward(dout.float())
    pad_u8 = pad.to(torch.uint8).contiguous()
    out = torch.full((B * T, D), float("nan"), device=DEV, dtype=dtype)
    desc = kn.make_attn_desc(q, k, v, out, pad_u8, None, B, T, 1, 1, S, H, False, True, 0.125, causal_q0=q0)
    kn.attn_fwd(desc, q)
    close(out, ref, dtype, what="fwd")
    dq = torch.full((B * T, D), float("nan"), device=DEV, dtype=dtype)
    dk = torch.full((B * S, D), float("nan"), device=DEV, dtype=dtype)
    dv = torch.full((B * S, D), float("nan"), device=DEV, dtype=dtype)
    stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device=DEV)
    kn.attn_bwd(desc, dout, dq, False, dk, dv, stats)
    close(dq, qf.grad, dtype, what="dq")
    close(dk, kf.grad, dtype, what="dk")
    close(dv, vf.grad, dtype, what="dv")
    for bad in (kn.make_attn_desc(q, k, v, out, pad_u8, None, B, T, 1, 1, S, H, False, True, 0.125, causal_q0=q0 + 8),       # not a multiple of 32
                kn.make_attn_desc(q, k, v, out, pad_u8, None, B, T, 1, 1, S, H, False, False, 0.125, causal_q0=q0),          # not causal
                kn.make_attn_desc(q, k, v, out, pad_u8, None, B, T, 1, 1, S, H, False, True, 0.125, causal_q0=S)):           # past the keys
        with pytest.raises(RuntimeError):
            kn.attn_fwd(bad, q)


@pytest.mark.parametrize("std", [1.0, 0.25], ids=["peaked", "flat"])
@pytest.mark.parametrize("case", [c for c in ATTN_CASES if c[0] in ("enc_self", "dec_self", "cross_text_full", "cross_img_one_entity", "cross_img")], ids=lambda c: c[0])
def test_attention_bf16_error_against_the_bf16_yardstick(case, std):
    """How far the bf16 attention kernels are from exact arithmetic, beside ANY bf16 matrix-core attention: the f64 attention of the
    bf16 inputs is the truth; the yardstick is the same algorithm in f32 with both products on bf16 operands (probabilities and
    score gradients rounded, oracle/bart_oracle._QuantMatmul).  Relative L2 error of out / dQ / dK / dV of the HIP kernels must
    stay within 1.5x the yardstick's + 1e-3.  `flat` = small scores (nearly uniform probabilities: dS = P (dP - delta) is then a
    difference of nearly equal numbers, the regime of a freshly initialised model), `peaked` = unit-variance q, k.
    `cross_img` (several 196-key entities per business) runs the chunked forward: running softmax over two chunks of keys
    (attn_tr_fwd_chunk_kernel)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import bart_oracle as bo
    name, B, qpb, N, S, T, H, exclude, causal, is_self = case
    nq, D, dtype = B * qpb, H * 64, torch.bfloat16
    g = torch.Generator().manual_seed(5)
    pad = torch.zeros(B, N, S, dtype=torch.bool)
    for b in range(B):
        for n in range(N):
            pad[b, n, int(torch.randint(max(1, S // 3), S + 1, (1,), generator=g)):] = True
    pad = pad.to(DEV)
    if is_self:
        qkv = rnd(nq * T, 3 * D, dtype=dtype, seed=1, std=std)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q = rnd(nq * T, D, dtype=dtype, seed=1, std=std)
        kv = rnd(B * N * S, 2 * D, dtype=dtype, seed=2, std=std)
        k, v = kv[:, :D], kv[:, D:]
    dout = rnd(nq * T, D, dtype=dtype, seed=3)

    def run(ft, mm):
        qf, kf, vf = (t.to(ft).contiguous().requires_grad_(True) for t in (q, k, v))
        o = attn_reference(qf, kf, vf, pad, nq, T, qpb, N, S, H, exclude, causal, 0.125, mm=mm)
        o.backward(dout.to(ft))
        return [t.detach().double() for t in (o, qf.grad, kf.grad, vf.grad)]

    truth = run(torch.float64, None)
    emu = [t.to(torch.bfloat16).double() for t in run(torch.float32, bo._QuantMatmul.apply)]      # results stored in bf16, like the kernels'
    pad_u8 = pad.to(torch.uint8).contiguous()
    null = torch.empty(B * N, dtype=torch.uint8, device=DEV)
    kn.entity_null(pad_u8, null, B * N, S)
    out = torch.zeros(nq * T, D, device=DEV, dtype=dtype)
    desc = kn.make_attn_desc(q, k, v, out, pad_u8, null, nq, T, qpb, N, S, H, exclude, causal, 0.125)
    kn.attn_fwd(desc, q)
    dq, dk, dv = torch.zeros_like(out), torch.zeros(B * N * S, D, device=DEV, dtype=dtype), torch.zeros(B * N * S, D, device=DEV, dtype=dtype)
    stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device=DEV)
    kn.attn_bwd(desc, dout, dq, False, dk, dv, stats)
    rel = lambda a, t: float((a.double() - t).norm() / t.norm().clamp_min(1e-30))      # noqa: E731
    rows = []
    for what, h, e, t in zip(("out", "dq", "dk", "dv"), (out, dq, dk, dv), emu, truth):
        rows.append((what, rel(h, t), rel(e, t)))
    print(name, std, " ".join("%s hip %.2e yardstick %.2e" % r for r in rows))
    for what, eh, ee in rows:
        assert eh <= 1.5 * ee + 1e-3, (name, std, what, eh, ee, rows)


MAPPED_CASES = [c for c in ATTN_CASES if c[0] in ("enc_self", "cross_text_loo", "cross_text_full", "cross_table_walk", "cross_img", "cross_text_holes")]


@pytest.mark.parametrize("case", MAPPED_CASES, ids=[c[0] for c in MAPPED_CASES])
def test_attention_row_maps(case):
    """Compact operands read through row maps (the padding-free encoder's [live, 3D] q/k/v, the compacted memory's K/V): the
    matrices hold only the rows that exist, in a shuffled order; same values as the padded layout, nothing written elsewhere."""
    dtype = torch.bfloat16
    name, B, qpb, N, S, T, H, exclude, causal, is_self = case
    nq = B * qpb
    D = H * 64
    g = torch.Generator().manual_seed(12)
    pad = torch.zeros(B, N, S, dtype=torch.bool)
    for b in range(B):
        for n in range(N):
            L = int(torch.randint(max(1, S // 3), S + 1, (1,), generator=g))
            pad[b, n, L:] = True
    if name.endswith("_holes"):
        pad = pad | (torch.rand(B, N, S, generator=g) < 0.3)
        pad[:, :, 5] = False
    if not is_self and N > 1:
        pad[0, N - 1, :] = True
    if name == "cross_table_walk":
        pad[1, 0, :] = True
    pad = pad.to(DEV)
    keep = (~pad).reshape(-1)
    nlive = int(keep.sum())
    order = torch.randperm(nlive, generator=g).to(DEV)                  # physical position of the i-th live row
    kv_rows = torch.full((B * N * S,), -1, dtype=torch.int32, device=DEV)
    kv_rows[keep] = order.to(torch.int32)
    if is_self:
        qkv = rnd(nq * T, 3 * D, dtype=dtype, seed=1)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        q_rows = kv_rows
        qkeep = keep
    else:
        q = rnd(nq * T, D, dtype=dtype, seed=1)
        kv = rnd(B * N * S, 2 * D, dtype=dtype, seed=2)
        k, v = kv[:, :D], kv[:, D:]
        q_rows, qkeep = None, torch.ones(nq * T, dtype=torch.bool, device=DEV)
    dout = rnd(nq * T, D, dtype=dtype, seed=3) * qkeep.unsqueeze(1).to(dtype)            # absent query rows carry no gradient
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
    ref = attn_reference(qf, kf, vf, pad, nq, T, qpb, N, S, H, exclude, causal, 0.125)
    ref.backward(dout.float())

    def compact(t, rows, n):                                             # [logical rows, W] -> [n, W] at the mapped positions
        out = torch.full((n, t.shape[1]), float("nan"), device=DEV, dtype=t.dtype)
        sel = rows >= 0
        out[rows[sel].long()] = t[sel]
        return out
    if is_self:
        qkv_c = compact(qkv, kv_rows, nlive)
        qc, kc, vc = qkv_c[:, :D], qkv_c[:, D:2 * D], qkv_c[:, 2 * D:]
        dout_c = compact(dout, q_rows, nlive)
        nqrows = nlive
    else:
        qc, dout_c, nqrows = q, dout, nq * T
        kv_c = compact(torch.cat([k, v], 1), kv_rows, nlive)
        kc, vc = kv_c[:, :D], kv_c[:, D:]
    pad_u8 = pad.to(torch.uint8).contiguous()
    null = torch.empty(B * N, dtype=torch.uint8, device=DEV)
    kn.entity_null(pad_u8, null, B * N, S)
    out = torch.full((nqrows, D), float("nan"), device=DEV, dtype=dtype)
    desc = kn.make_attn_desc(qc, kc, vc, out, pad_u8, null, nq, T, qpb, N, S, H, exclude, causal, 0.125, q_rows=q_rows, kv_rows=kv_rows)
    kn.attn_fwd(desc, qc)
    ref_c = compact(ref.detach(), q_rows, nlive) if is_self else ref
    close(out, ref_c, dtype, what=name + " mapped fwd")
    dq = torch.full((nqrows, D), float("nan"), device=DEV, dtype=dtype)
    dk = torch.full((nlive, D), float("nan"), device=DEV, dtype=dtype)
    dv = torch.full((nlive, D), float("nan"), device=DEV, dtype=dtype)
    stats = torch.empty(kn.attn_bwd_workspace(desc) // 4, device=DEV)
    kn.attn_bwd(desc, dout_c, dq, False, dk, dv, stats)
    close(dq, compact(qf.grad, q_rows, nlive) if is_self else qf.grad, dtype, what=name + " mapped dq")
    close(dk, compact(kf.grad, kv_rows, nlive), dtype, what=name + " mapped dk")
    close(dv, compact(vf.grad, kv_rows, nlive), dtype, what=name + " mapped dv")
    # the same launch on the padded layout gives the same bits
    out_p = torch.full((nq * T, D), float("nan"), device=DEV, dtype=dtype)
    desc_p = kn.make_attn_desc(q, k, v, out_p, pad_u8, null, nq, T, qpb, N, S, H, exclude, causal, 0.125)
    kn.attn_fwd(desc_p, q)
    sel = (q_rows >= 0) if is_self else torch.ones(nq * T, dtype=torch.bool, device=DEV)
    assert torch.equal(out_p[sel], out[q_rows[sel].long()] if is_self else out)
    with pytest.raises(RuntimeError, match="mmsum_attn_fwd"):           # row maps are a bf16 feature: f32 refuses them
        kn.attn_fwd(kn.make_attn_desc(qc.float(), kc.float(), vc.float(), out.float(), pad_u8, null, nq, T, qpb, N, S, H, exclude, causal,
                                      0.125, q_rows=q_rows, kv_rows=kv_rows), qc.float())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gate(dtype):
    Bq, rows, D = 4, 6, 256
    R = Bq * rows
    ts = [rnd(R, D, dtype=dtype, seed=s) for s in range(1, 7)]
    pa, pb, yt, ytab, yimg, dout = ts
    no_table = torch.tensor([0, 1, 0, 0], dtype=torch.uint8, device=DEV)
    no_img = torch.tensor([0, 0, 1, 0], dtype=torch.uint8, device=DEV)
    fl = [t.float().requires_grad_(True) for t in (pa, pb, yt, ytab, yimg)]
    ma = (1 - no_table.float()).repeat_interleave(rows)[:, None]
    mb = (1 - no_img.float()).repeat_interleave(rows)[:, None]
    ref = fl[2] + ma * F.relu(torch.tanh(fl[0])) * fl[3] + mb * F.relu(torch.tanh(fl[1])) * fl[4]
    ref.backward(dout.float())
    out = torch.empty_like(yt)
    kn.gate_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, out, rows)
    close(out, ref, dtype, what="gate fwd")
    outs = [torch.empty_like(yt) for _ in range(5)]
    kn.gate_bwd(dout, pa, pb, ytab, yimg, no_table, no_img, *outs, rows)
    for o, f, nm in zip(outs, fl, ["dpa", "dpb", "dyt", "dytab", "dyimg"]):
        close(o, f.grad, dtype, what=nm)
    # the form with the two bias gradients summed on the way (alpha_proj, beta_proj): same five outputs, sums of the values AS
    # STORED, added to what the buffers held -- at the step's width too
    for R2, D2 in ((R, D), (4 * 130, 1024)):
        ts2 = [rnd(R2, D2, dtype=dtype, seed=10 + s_) for s_ in range(6)]
        nt2 = torch.zeros(4, dtype=torch.uint8, device=DEV)
        nt2[1] = 1
        ni2 = torch.zeros(4, dtype=torch.uint8, device=DEV)
        ni2[2] = 1
        plain = [torch.empty_like(ts2[0]) for _ in range(5)]
        kn.gate_bwd(ts2[5], ts2[0], ts2[1], ts2[3], ts2[4], nt2, ni2, *plain, R2 // 4)
        fused = [torch.empty_like(ts2[0]) for _ in range(5)]
        sums = [torch.full((D2,), 0.5, device=DEV) for _ in range(2)]
        kn.gate_bwd(ts2[5], ts2[0], ts2[1], ts2[3], ts2[4], nt2, ni2, *fused, R2 // 4, sums=sums)
        for a_, b_ in zip(plain, fused):           # two kernels, the same arithmetic: equal to the last bit or one FMA contraction apart
            close(b_, a_.float(), dtype, what="gate bwd with sums")
        want = [plain[0].double().sum(0), plain[1].double().sum(0)]
        for got, w_, nm in zip(sums, want, ["sum dpa", "sum dpb"]):
            assert float((got.double() - 0.5 - w_).abs().max()) <= 1e-4 * float(w_.abs().max()) + 1e-4, nm


def test_gemm_epilogue_statistics_small_tiles():
    """Column sums and sums of squares of the stored result in the GEMM epilogue (MMSUM_GEMM_COLSUM | COLSUM2: the BatchNorm
    statistics of a convolution's output) on the smaller-tile ring kernels and on ragged edges: N = 64 / 128 (256 x 128 tiles,
    the early ResNet stages), rows that end inside a tile, and the statistics -> {mean, variance} + running-statistics kernel."""
    for M, N, K in ((1000, 64, 192), (5000, 128, 576), (3000, 256, 64), (700, 1024, 256)):
        a, w = rnd(M, K, dtype=torch.bfloat16, seed=1, std=0.5), rnd(N, K, dtype=torch.bfloat16, seed=2, std=0.5)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        raw = torch.zeros(2 * N, device=DEV)
        kn.gemm(a, w, out, colsum=raw, colsum_sq=True)
        od = out.double()
        close(out, a.float() @ w.float().t(), torch.bfloat16, what="gemm with statistics")
        assert float((raw[:N].double() - od.sum(0)).abs().max()) <= 1e-4 * float(od.abs().sum(0).max()) + 1e-3, (M, N, K)
        assert float((raw[N:].double() - (od * od).sum(0)).abs().max()) <= 1e-4 * float((od * od).sum(0).max()) + 1e-3, (M, N, K)
        sums = torch.empty(2 * N, device=DEV)
        rm, rv = torch.full((N,), 0.5, device=DEV), torch.full((N,), 2.0, device=DEV)
        kn.bn_stats_from_sums(raw, M, sums, rm, rv, 0.1)
        mean, var = od.mean(0), od.var(0, unbiased=False)
        assert float((sums[:N].double() - mean).abs().max()) <= 1e-4 * float(mean.abs().max()) + 1e-5
        assert float((sums[N:].double() - var).abs().max()) <= 1e-3 * float(var.max()) + 1e-5
        assert float((rm.double() - (0.45 + 0.1 * mean)).abs().max()) <= 1e-4 and float((rv.double() - (1.8 + 0.1 * var * M / (M - 1))).abs().max()) <= 1e-3 * float(var.max()) + 1e-4
        # the form the bf16 step takes: mmsum_bn_apply derives the statistics from `raw` itself, writes `sums` and updates the running
        # statistics in the same launch -- equal to the two-launch form above to the last bit, with a residual and in the padded layout too
        gamma, beta = 1 + 0.1 * rnd(N, seed=3), 0.1 * rnd(N, seed=4)
        res = rnd(M, N, dtype=torch.bfloat16, seed=5)
        y_two = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        kn.bn_apply(out, sums, gamma, beta, res, y_two, None, None, 1e-5, 0.1, True, True)
        sums2 = torch.full((2 * N,), float("nan"), device=DEV)
        rm2, rv2 = torch.full((N,), 0.5, device=DEV), torch.full((N,), 2.0, device=DEV)
        y_one = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        kn.bn_apply(out, sums2, gamma, beta, res, y_one, rm2, rv2, 1e-5, 0.1, True, True, raw=raw)
        assert torch.equal(sums2, sums) and torch.equal(rm2, rm) and torch.equal(rv2, rv) and torch.equal(y_one, y_two), (M, N, K)
        ref_y = torch.relu((od - mean) / (var + 1e-5).sqrt() * gamma.double() + beta.double() + res.double())
        assert float((y_one.double() - ref_y).abs().max()) <= 2e-2 * float(ref_y.abs().max())


def test_bn_statistics_from_raw_sums_with_large_means():
    """var = E[y^2] - E[y]^2 from the GEMM epilogue's plain f32 sums cancels when |mean| >> std.  How much: columns whose mean is
    10 x their standard deviation (far beyond what a convolution with zero-mean weights produces) keep the variance to 2e-3 relative and the
    mean to 1e-5 against the f64 statistics of the SAME stored bf16 values -- below the 2^-9 rounding of the values themselves."""
    M, N, K = 20000, 256, 64
    a = rnd(M, K, dtype=torch.bfloat16, seed=1, std=0.25)
    a[:, 0] = 1.0                                                  # a constant input channel ...
    w = rnd(N, K, dtype=torch.bfloat16, seed=2, std=0.5)
    w[:, 0] = 20.0                                                 # ... whose weight puts a mean of 20 on every output column (std = 0.5 * 0.25 * 8 = 1 -> 2 with the rest)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    raw = torch.zeros(2 * N, device=DEV)
    kn.gemm(a, w, out, colsum=raw, colsum_sq=True)
    od = out.double()
    mean, var = od.mean(0), od.var(0, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 8.0
    sums = torch.empty(2 * N, device=DEV)
    kn.bn_stats_from_sums(raw, M, sums, None, None, 0.1)
    assert float(((sums[:N].double() - mean) / mean).abs().max()) <= 1e-5
    assert float(((sums[N:].double() - var) / var).abs().max()) <= 2e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_ls_loss(dtype, golden_dir):
    from oracle import bart_oracle as bo
    from multimodalsum_amd.formula_init import formula_tensor
    g = np.load(os.path.join(golden_dir, "f5_loss.npz"))
    V, ld = 50265, 50304
    target = torch.from_numpy(g["target"]).to(DEV)
    logits_cpu = formula_tensor("f5.logits", (8, V), std=2.0)
    buf = torch.zeros(8, ld, device=DEV, dtype=dtype)
    buf[:, :V] = logits_cpu.to(DEV).to(dtype)
    src = buf[:, :V].float().cpu().requires_grad_(True)
    ref = bo.label_smoothing_loss(src, target.cpu(), V, 0.1)
    ref.backward()
    rows = torch.empty(8, device=DEV)
    kn.ls_loss(buf, target, rows, V, 0.1, 1.0 / 8, True)
    tot = torch.empty(1, device=DEV)
    kn.segment_sum(rows, tot, 1, 8, 1.0 / 8)
    assert abs(tot.item() - ref.item()) < 1e-4 * abs(ref.item()) + 1e-5
    if dtype == torch.float32:
        assert abs(tot.item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
        close(buf[:, :64], torch.from_numpy(g["grad_sample"]), dtype, scale=0.01, what="golden grad")
    close(buf[:, :V], src.grad, dtype, scale=0.05, what="dlogits")
    assert buf[:, V:].abs().max().item() == 0.0
    # smoothing 0 == cross entropy
    buf[:, :V] = logits_cpu.to(DEV).to(dtype)
    kn.ls_loss(buf, target, rows, V, 0.0, 1.0, False)
    ce = F.cross_entropy(buf[:, :V].float(), target, reduction="none")
    close(rows, ce, torch.float32, scale=5, what="ce rows")


def test_optimizer_kernels():
    from oracle import step_oracle as so
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2, std=0.01)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pc, gc, mc, vc = p.cpu().clone(), g.cpu().clone(), m.cpu().clone(), v.cpu().clone()
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    nsq = torch.empty(1, device=DEV)
    kn.l2norm_sq(g, nsq)
    ref_n = (gc.double() ** 2).sum().item()
    assert abs(nsq.item() - ref_n) < 1e-5 * ref_n
    for step in range(1, 4):
        lr, wd = 1e-3, 0.01
        step_size = lr * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        hyper = torch.tensor([step_size, lr * wd, 0.5, 0.0], device=DEV)
        kn.adamw(p, g, m, v, shadow, hyper, nsq, 0.9, 0.999, 1e-6)
        gcl = gc.clone()
        so.clip_grad_norm([gcl], 0.5)
        so.adamw_step(pc, gcl, mc, vc, step, lr, weight_decay=wd)
    close(p, pc, torch.float32, scale=0.1, what="adamw p")
    close(m, mc, torch.float32, scale=0.1, what="adamw m")
    close(shadow, pc, torch.bfloat16, what="shadow")
    g2 = g.clone()
    kn.scale_by_clip(g2, nsq, 0.5)
    close(g2, gcl, torch.float32, scale=0.1, what="clip scale")
    x = rnd(1001, seed=5)
    xb = torch.empty(1001, device=DEV, dtype=torch.bfloat16)
    kn.cast(xb, x)
    assert torch.equal(xb, x.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 16, 8, 16, 3, 1, 1), (2, 16, 16, 8, 16, 3, 2, 1), (2, 20, 20, 3, 8, 7, 2, 3)])
def test_conv_im2col(dtype, cfg):
    N, H, W, C, Cout, KS, stride, pad = cfg
    Ho, Wo = (H + 2 * pad - KS) // stride + 1, (W + 2 * pad - KS) // stride + 1
    Kd = KS * KS * C
    Kpad = (Kd + 63) // 64 * 64
    x = rnd(N, C, H, W, seed=1)
    w = rnd(Cout, C, KS, KS, seed=2, std=0.2)
    xn = torch.empty(N * H * W, C, device=DEV, dtype=dtype)
    kn.nchw_to_nhwc(x, xn, N, C, H, W)
    close(xn, x.permute(0, 2, 3, 1).reshape(-1, C), dtype, what="nhwc")
    xr = xn.float().view(N, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wm = torch.empty(Cout, Kpad, device=DEV, dtype=dtype)
    kn.conv_weight_to_matrix(wm, w, Cout, C, KS, KS, Kpad)
    wr = wm[:, :Kd].float().reshape(Cout, KS, KS, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=stride, padding=pad)
    col = torch.full((N * Ho * Wo, Kpad), float("nan"), device=DEV, dtype=dtype)
    kn.im2col(xn, col, N, H, W, C, KS, KS, stride, pad, Ho, Wo, Kpad)
    y = torch.empty(N * Ho * Wo, Cout, device=DEV, dtype=dtype)
    kn.gemm(col, wm, y)
    close(y, ref.permute(0, 2, 3, 1).reshape(-1, Cout), dtype, scale=2, what="conv fwd")
    if C % 4 == 0:
        dy = rnd(N * Ho * Wo, Cout, dtype=dtype, seed=3)
        ref.backward(dy.float().view(N, Ho, Wo, Cout).permute(0, 3, 1, 2))
        dcol = torch.empty(N * Ho * Wo, Kpad, device=DEV, dtype=dtype)
        kn.gemm(dy, wm, dcol, b_t=True)
        dx = torch.empty(N * H * W, C, device=DEV, dtype=dtype)
        kn.col2im(dcol, dx, N, H, W, C, KS, KS, stride, pad, Ho, Wo, Kpad)
        close(dx, xr.grad.permute(0, 2, 3, 1).reshape(-1, C), dtype, scale=4, what="conv dgrad")
        dwm = torch.zeros(Cout, Kpad, device=DEV)
        kn.gemm(dy, col, dwm, a_t=True, b_t=True, accumulate=True)
        dw = torch.ones(Cout, C, KS, KS, device=DEV)
        kn.conv_matrix_grad_to_weight(dwm, dw, Cout, C, KS, KS, Kpad, True)
        close(dw, 1 + wr.grad, dtype, scale=4, what="conv wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2 * 14 * 14, 64), (3001, 256), (777, 1024), (40000, 128)], ids=lambda t: "%dx%d" % t)
def test_batchnorm_pool(dtype, shape):
    R, C = shape
    x, res, dy = rnd(R, C, dtype=dtype, seed=1), rnd(R, C, dtype=dtype, seed=2), rnd(R, C, dtype=dtype, seed=3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xf, rf = x.float().requires_grad_(True), res.float().requires_grad_(True)
    gf, bf = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(xf, rm_ref, rv_ref, gf, bf, True, 0.1, 1e-5) + rf)
    ref.backward(dy.float())
    sums = torch.empty(2 * C, device=DEV)
    kn.bn_reduce(x, sums)
    y = torch.empty_like(x)
    kn.bn_apply(x, sums, gamma, beta, res, y, rm, rv, 1e-5, 0.1, True, True)
    close(y, ref, dtype, what="bn fwd")
    close(rm, rm_ref, torch.float32, scale=4, what="running mean")
    close(rv, rv_ref, torch.float32, scale=4, what="running var")
    dsums = torch.empty(2 * C, device=DEV)
    kn.bn_bwd_reduce(dy, y, x, sums, dsums, 1e-5, True)
    dx, dres = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    kn.bn_bwd_apply(dy, y, x, sums, dsums, gamma, dx, dres, dg, db, 1e-5, True)
    close(dx, xf.grad, dtype, scale=2, what="bn dx")
    close(dres, rf.grad, dtype, what="bn dres")
    close(dg, gf.grad, dtype, scale=8, what="bn dgamma")
    close(db, bf.grad, dtype, scale=8, what="bn dbeta")
    # eval mode
    kn.bn_apply(x, sums, gamma, beta, None, y, rm, rv, 1e-5, 0.1, False, False)
    close(y, F.batch_norm(x.float(), rm, rv, gamma, beta, False, 0.1, 1e-5), dtype, what="bn eval")
    if R != 2 * 14 * 14:
        return
    # max-pool 3x3/2 pad 1
    N, H, W = 2, 14, 14
    yp = torch.empty(N * 7 * 7, C, device=DEV, dtype=dtype)
    kn.maxpool3x3s2(x, yp, N, H, W, C, 7, 7)
    refp = F.max_pool2d(x.float().view(N, H, W, C).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, C)
    close(yp, refp, dtype, what="maxpool")


@pytest.mark.parametrize("dtype", DTYPES)
def test_table_gather(dtype):
    from multimodalsum_amd import synthetic as syn
    B, V, D = 3, 200, 1024
    field, fv = syn.table_batch(B, V, seed=21)
    E = rnd(V, D, dtype=dtype, seed=1, std=0.05)
    wr, wh = rnd(D, 4, dtype=dtype, seed=2), rnd(D, 4, dtype=dtype, seed=3)
    Ef = E.float().cpu()
    name, category, str_cat, str_bool, rating, hours = fv

    def msum(ids, dim):
        return (F.embedding(ids, Ef) * ids.ne(1).unsqueeze(-1).float()).sum(dim=dim)

    names = msum(field, 1).unsqueeze(0).expand(B, -1, -1)
    cat_valid = category.ne(1).any(-1).unsqueeze(-1).float()
    vals = torch.cat([msum(name, 1).unsqueeze(1),
                      (msum(category, 2) * cat_valid).sum(1, keepdim=True) / (cat_valid.sum(1, keepdim=True) + 1e-6),
                      msum(str_cat, 2), F.embedding(str_bool.squeeze(-1), Ef) * str_bool.ne(1).float(),
                      F.linear(rating.float(), wr.float().cpu()).unsqueeze(1), F.linear(hours.float(), wh.float().cpu())], 1)
    ref = torch.cat([names, vals], -1)
    ones = torch.ones(B, 1, dtype=torch.bool)
    mref = torch.cat([ones, category[:, :1, 0].ne(1), str_cat[:, :, 0].ne(1), str_bool[:, :, 0].ne(1), ones, hours.sum(-1) != 0], 1)
    out = torch.empty(B * 47, 2 * D, device=DEV, dtype=dtype)
    mask = torch.empty(B, 47, dtype=torch.uint8, device=DEV)
    kn.table_gather(E, field.to(DEV), [t.to(DEV).contiguous() for t in fv], wr, wh, out, mask, B, 1)
    close(out, ref.view(B * 47, 2 * D), dtype, what="table gather")
    assert torch.equal(mask.bool().cpu(), mref)
    dall = rnd(B * 47, 2 * D, dtype=dtype, seed=4)
    dwr, dwh = torch.zeros(D, 4, device=DEV), torch.zeros(D, 4, device=DEV)
    kn.table_gather_bwd(dall, rating.to(DEV), hours.to(DEV), dwr, dwh, B, D)
    dv = dall.float().cpu().view(B, 47, 2 * D)[:, :, D:]
    close(dwr, torch.einsum("bk,bd->dk", rating.float(), dv[:, 39]), dtype, what="dw_rating")
    close(dwh, torch.einsum("bjk,bjd->dk", hours.float(), dv[:, 40:47]), dtype, scale=2, what="dw_hours")


@pytest.mark.parametrize("dtype", DTYPES)
def test_amazon_table_gather(dtype):
    """mmsum_amazon_table_gather(_bwd) against the oracle's statement of AmazonTableEncoder's gather (table_encoder.py:106-166)."""
    from multimodalsum_amd import synthetic as syn
    B, V, D = 3, 200, 1024
    field, fv = syn.amazon_table_batch(B, V, seed=22)
    E = rnd(V, D, dtype=dtype, seed=1, std=0.05)
    wp, wr = rnd(D, 11, dtype=dtype, seed=2), rnd(D, 4, dtype=dtype, seed=3)
    Ef = E.float().cpu()
    price, rating, brand, name, category, description = fv

    def msum(ids, dim):
        return (F.embedding(ids, Ef) * ids.ne(1).unsqueeze(-1).float()).sum(dim=dim)

    fn = F.embedding(field, Ef).squeeze(1)
    names = torch.cat([fn[:-1], fn[-1:].repeat(128, 1)]).unsqueeze(0).expand(B, -1, -1)
    rv = category.ne(1).any(-1)
    groups = (msum(category, 3) * rv.unsqueeze(-1).float()).sum(2) / (rv.float().sum(2, keepdim=True) + 1e-6)
    gv = rv.any(-1).unsqueeze(-1).float()
    cat = (groups * gv).sum(1, keepdim=True) / (gv.sum(1, keepdim=True) + 1e-6)
    vals = torch.cat([F.linear(price.float(), wp.float().cpu()).unsqueeze(1), F.linear(rating.float(), wr.float().cpu()).unsqueeze(1),
                      msum(brand, 1).unsqueeze(1), msum(name, 1).unsqueeze(1), cat, F.embedding(description, Ef)], 1)
    ref = torch.cat([names, vals], -1)
    ones = torch.ones(B, 1, dtype=torch.bool)
    mref = torch.cat([price.sum(1, keepdim=True) != 0, ones, brand[:, :1].ne(1), name[:, :1].ne(1), ones, description.ne(1)], 1)
    out = torch.empty(B * 133, 2 * D, device=DEV, dtype=dtype)
    mask = torch.empty(B, 133, dtype=torch.uint8, device=DEV)
    kn.amazon_table_gather(E, field.to(DEV), [t.to(DEV).contiguous() for t in fv], wp, wr, out, mask, B, 1)
    close(out, ref.reshape(B * 133, 2 * D), dtype, what="amazon table gather")
    assert torch.equal(mask.bool().cpu(), mref)
    dall = rnd(B * 133, 2 * D, dtype=dtype, seed=4)
    dwp, dwr = torch.ones(D, 11, device=DEV), torch.ones(D, 4, device=DEV)
    kn.amazon_table_gather_bwd(dall, price.to(DEV), rating.to(DEV), dwp, dwr, B, D)
    dv = dall.float().cpu().view(B, 133, 2 * D)[:, :, D:]
    close(dwp, 1 + torch.einsum("bk,bd->dk", price.float(), dv[:, 0]), dtype, what="d price weight")
    close(dwr, 1 + torch.einsum("bk,bd->dk", rating.float(), dv[:, 1]), dtype, what="d rating weight")


@pytest.mark.parametrize("dtype", DTYPES)
def test_rows_gather(dtype):
    """mmsum_rows_gather: compact (valid rows first, filler rows zero) and expand (padding rows zero), strided operands."""
    R, C, cap = 300, 1024, 200
    g = torch.Generator().manual_seed(3)
    mask = torch.rand(R, generator=g) < 0.6
    n = int(mask.sum())
    assert n <= cap
    src_wide = rnd(R, C + 64, dtype=dtype, seed=9)
    src = src_wide[:, 64:]
    pos = torch.cumsum(mask.long(), 0) - 1
    p2c = torch.where(mask, pos, torch.full_like(pos, -1)).to(DEV)
    c2p = torch.full((cap,), -1, dtype=torch.long)
    c2p[:n] = torch.nonzero(mask).flatten()
    c2p = c2p.to(DEV)
    comp = torch.full((cap, C), 7.0, device=DEV, dtype=dtype)
    kn.rows_gather(src, comp, c2p)
    assert torch.equal(comp[:n], src[mask.to(DEV)]) and (comp[n:] == 0).all()
    back = torch.full((R, C), 7.0, device=DEV, dtype=dtype)
    kn.rows_gather(comp, back, p2c)
    assert torch.equal(back[mask.to(DEV)], src[mask.to(DEV)]) and (back[~mask.to(DEV)] == 0).all()


def test_attention_rejects_more_than_32_entities():
    """Entity sets are 32-bit masks inside the kernels: N > 32 is a shape error, not a silent truncation."""
    D = 64
    q = torch.zeros(4, D, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(33 * 4, D, device=DEV, dtype=torch.bfloat16)
    pad = torch.zeros(33 * 4, dtype=torch.uint8, device=DEV)
    null = torch.zeros(33, dtype=torch.uint8, device=DEV)
    desc = kn.make_attn_desc(q, kv, kv, torch.empty_like(q), pad, null, 1, 4, 1, 33, 4, 1, False, False, 0.125)
    with pytest.raises(RuntimeError, match="mmsum_attn_fwd"):
        kn.attn_fwd(desc, q)


# ------------------------------------------------------------------------------------------------
# beam-search decode step (csrc/decode.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("num_beams,V", [(4, 50265), (2, 200), (8, 1031)])
def test_beam_topk(dtype, num_beams, V):
    """mmsum_beam_topk against the reference's sequence of torch ops (adjust_logits -> log_softmax -> bans -> + beam scores ->
    topk over [B, beams * V], modeling_multimodalsum.py:2874-2925): same candidate ids in the same order, scores to f32
    rounding; forced token, min-length ban and n-gram bans included."""
    B = 3
    R, K = B * num_beams, 2 * num_beams
    ld = (V + 127) // 128 * 128
    g = torch.Generator().manual_seed(7)
    for force, ban, with_bans in ((-1, -1, False), (-1, 2, True), (0, -1, False), (2, -1, True)):
        logits = (torch.randn(R, ld, generator=g) * 3).to(DEV).to(dtype)
        beam_scores = (torch.randn(R, generator=g) * 2).to(DEV)
        beam_scores[1] = -1e9
        banned = None
        if with_bans:
            banned = torch.full((R, 7), -1, dtype=torch.int32)
            top_tok = logits[:, :V].float().argmax(-1).cpu()
            for r in range(R):
                banned[r, 0] = int(top_tok[r])                    # ban each row's best token: the winner must change
                banned[r, 1] = (r * 13 + 5) % V                   # (the list is filled from the front: the first -1 ends it)
            banned = banned.to(DEV)
        ref = logits[:, :V].float().clone()
        if force >= 0:
            keep = ref[:, force].clone()
            ref.fill_(float("-inf"))
            ref[:, force] = keep
        sc = torch.log_softmax(ref, dim=-1)
        if ban >= 0:
            sc[:, ban] = float("-inf")
        if banned is not None:
            for r in range(R):
                for t in banned[r].tolist():
                    if t >= 0:
                        sc[r, t] = float("-inf")
        cand = (sc + beam_scores[:, None]).view(B, num_beams * V)
        want_s, want_i = torch.topk(cand, K, dim=1)
        out_s = torch.zeros(B, K, device=DEV)
        out_i = torch.zeros(B, K, dtype=torch.int64, device=DEV)
        kn.beam_topk(logits, V, beam_scores, banned, force, ban, num_beams, out_s, out_i)
        finite = torch.isfinite(want_s)
        assert torch.equal(torch.isfinite(out_s), finite), (force, ban)
        tol = 2e-4
        # the K best scores, best first ...
        assert float((out_s[finite] - want_s[finite]).abs().max()) <= tol, (force, ban)
        # ... each belonging to the candidate it names, no candidate twice (bf16 logits tie often: among equal scores the
        # kernel returns the lower index, torch.topk leaves the order open, so ids are compared only where f32 logits make
        # ties impossible)
        for b in range(B):
            ids = out_i[b][finite[b]]
            assert ids.unique().numel() == ids.numel()
            assert float((cand[b][ids] - out_s[b][finite[b]]).abs().max()) <= tol
        if dtype == torch.float32:
            assert torch.equal(out_i[finite], want_i[finite]), (force, ban, out_i, want_i)     # -inf ties have no defined order
        if banned is not None and force < 0:
            assert bool(torch.isinf(logits[0, int(banned[0, 0])].float()))                     # the ban is written into the logits


@pytest.mark.parametrize("dtype", DTYPES)
def test_decode_self_attn_through_ancestor_table(dtype):
    """mmsum_decode_self_attn: one query per hypothesis over cache rows reached through the ancestor table == softmax(q K^T) V
    over the gathered rows (what the reference gets after index_select-ing its caches, modeling_multimodalsum.py:3104-3115).
    bf16 takes the 256-thread kernel (every K row in registers, V rows by LDS-DMA): lengths across its 32-row slots up to Tmax = 256."""
    R, H, Tmax, D = 12, 16, 256, 1024
    g = torch.Generator().manual_seed(11)
    for length in (1, 7, 40, 65, 130, 256):
        q = torch.randn(R, 3 * D, generator=g).to(DEV).to(dtype)               # a [R, 3D] qkv buffer: the query is a strided view
        kc = torch.randn(R * Tmax, D, generator=g).to(DEV).to(dtype)
        vc = torch.randn(R * Tmax, D, generator=g).to(DEV).to(dtype)
        anc = torch.randint(0, R, (R, Tmax), generator=g, dtype=torch.int32).to(DEV)
        out = torch.full((R, D), float("nan"), device=DEV, dtype=dtype)
        kn.decode_self_attn(q[:, :D], kc, vc, anc, out, H, length, Tmax, 0.125)
        s = torch.arange(length, device=DEV)
        ref = torch.empty(R, D, device=DEV)
        for r in range(R):
            phys = anc[r, :length].long() * Tmax + s
            k = kc.float()[phys].view(length, H, 64)
            v = vc.float()[phys].view(length, H, 64)
            p = torch.softmax(torch.einsum("hd,shd->hs", q[r, :D].float().view(H, 64) * 0.125, k), dim=-1)
            ref[r] = torch.einsum("hs,shd->hd", p, v).reshape(D)
        close(out, ref, dtype, what="decode self-attention, length %d" % length)
        # the step's own K / V handed in: appended to the caches by the kernel and used for position length - 1
        anc2 = anc.clone()
        anc2[:, length - 1] = torch.arange(R, dtype=torch.int32, device=DEV)
        kv_new = torch.randn(R, 2 * D, generator=g).to(DEV).to(dtype)
        kc2, vc2 = kc.clone(), vc.clone()
        out2 = torch.full((R, D), float("nan"), device=DEV, dtype=dtype)
        kn.decode_self_attn(q[:, :D], kc2, vc2, anc2, out2, H, length, Tmax, 0.125, k_new=kv_new[:, :D], v_new=kv_new[:, D:])
        kc3, vc3 = kc.clone(), vc.clone()
        kc3.view(R, Tmax, D)[:, length - 1] = kv_new[:, :D]
        vc3.view(R, Tmax, D)[:, length - 1] = kv_new[:, D:]
        assert torch.equal(kc2, kc3) and torch.equal(vc2, vc3)
        out3 = torch.full((R, D), float("nan"), device=DEV, dtype=dtype)
        kn.decode_self_attn(q[:, :D], kc3, vc3, anc2, out3, H, length, Tmax, 0.125)
        assert torch.equal(out2, out3)


# ------------------------------------------------------------------------------------------------
# the decode step's own kernels (round 4)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,epi,k2,f32", [
    (32, 1024, 1024, 0, 0, False), (96, 1024, 1024, 0, 0, False), (32, 3072, 1024, 0, 0, False), (32, 4096, 1024, 1, 0, False),
    (32, 1024, 4096, 0, 0, False), (32, 1024, 2048, 0, 1024, False), (12, 1024, 1024, 0, 0, False), (8, 264, 256, 0, 0, False),
    (8, 256, 512, 0, 256, False), (32, 5001, 1024, 0, 0, True), (7, 4999, 1024, 0, 0, True)],
    ids=["out", "out3R", "qkv", "fc1_gelu", "fc2", "alpha_two_tensors", "R12", "tiny_ragged_N", "tiny_two_tensors", "lm_head_f32", "lm_head_f32_R7"])
def test_dec_gemm(M, N, K, epi, k2, f32):
    """mmsum_dec_gemm (the reduction split over one-wave workgroups, the tile's last arriver reduces and runs the epilogue) against an
    f64 product of the same bf16 operands; the same workspace serves two different products one after the other and a REPEAT of the
    first (the ticket words are back at zero after every launch); rows past M and columns past N of a padded output stay untouched."""
    bf = torch.bfloat16
    x = rnd(M, K, dtype=torch.float32 if f32 else bf, seed=1)
    w = rnd(N, K, dtype=bf, seed=2, std=0.05)
    bias = rnd(N, seed=3)
    ws = kn.dec_gemm_workspace(96, max(N, 4096), 4096, DEV)
    out = torch.full((M + 1, N + 3), float("nan"), device=DEV, dtype=torch.float32 if f32 else bf)
    x1, x2 = (x[:, :k2].contiguous(), x[:, k2:].contiguous()) if k2 else (x, None)

    def run():
        kn.dec_gemm(x1, w, out[:M, :N], ws, bias=bias, epi=epi, x2=x2)
    run()
    ref = x.double() @ w.double().t() + bias.double()
    if epi:
        ref = torch.nn.functional.gelu(ref)
    first = out.clone()
    assert torch.isnan(out[M]).all() and torch.isnan(out[:, N:]).all()
    tol_ = 1e-5 + (1e-5 if f32 else 1e-2) * float(ref.abs().max())        # f32 x: bf16 hi + lo parts carry 16 bits of x; the weights are bf16
    assert float((out[:M, :N].double() - ref).abs().max()) <= tol_, float((out[:M, :N].double() - ref).abs().max())
    # another product through the same workspace, then the first one again: bit-identical
    y = rnd(16, 1024, dtype=bf, seed=5)
    wy = rnd(2048, 1024, dtype=bf, seed=6, std=0.05)
    oy = torch.empty(16, 2048, device=DEV, dtype=bf)
    kn.dec_gemm(y, wy, oy, ws)
    assert float((oy.double() - y.double() @ wy.double().t()).abs().max()) <= 1e-2 * float((y.double() @ wy.double().t()).abs().max()) + 1e-5
    out.fill_(float("nan"))
    run()
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(first, nan=7.0))


@pytest.mark.parametrize("on_logits", [False, True], ids=["beam", "greedy"])
@pytest.mark.parametrize("num_beams,V", [(1, 50265), (4, 50265), (3, 1000)])
def test_beam_topk_repetition_penalty(num_beams, V, on_logits):
    """mmsum_beam_topk with the repetition penalty (enforce_repetition_penalty_, generation_utils.py:47-55) in the reference's order --
    log_softmax, penalty on the log-probabilities, bans (beam search, :2874-2900), or penalty and bans on the raw logits (greedy
    decoding, :2749-2783: only the order within a row matters there) -- against the same torch ops: each row's best token is
    penalised (the winner must change), one banned token is also listed (the ban wins), f32 logits."""
    B = 3
    R, K = B * num_beams, 2 * num_beams
    ld = (V + 127) // 128 * 128
    g = torch.Generator().manual_seed(17)
    for penalty in (1.7, 0.6):
        logits = (torch.randn(R, ld, generator=g) * 3).to(DEV)
        beam_scores = torch.zeros(R, device=DEV) if on_logits else (torch.randn(R, generator=g) * 2).to(DEV)
        order = logits[:, :V].argsort(-1, descending=True).cpu()
        pen = torch.full((R, 9), -1, dtype=torch.int32)
        banned = torch.full((R, 3), -1, dtype=torch.int32)
        for r in range(R):
            pen[r, 0], pen[r, 1], pen[r, 2], pen[r, 3] = int(order[r, 0]), int(order[r, 2]), int(order[r, 40]), (r * 13 + 5) % V
            banned[r, 0] = int(order[r, 2])                      # listed AND banned: the ban comes after the penalty
        ref = logits[:, :V].float().clone()
        sc = ref if on_logits else torch.log_softmax(ref, dim=-1)
        for r in range(R):
            for t in pen[r].tolist():
                if t >= 0:
                    sc[r, t] = sc[r, t] * penalty if sc[r, t] < 0 else sc[r, t] / penalty
        if on_logits:
            sc = sc - torch.logsumexp(logits[:, :V].float(), dim=-1, keepdim=True)
        for r in range(R):
            sc[r, int(banned[r, 0])] = float("-inf")
        cand = (sc + beam_scores[:, None]).view(B, num_beams * V)
        want_s, want_i = torch.topk(cand, K, dim=1)
        out_s = torch.zeros(B, K, device=DEV)
        out_i = torch.zeros(B, K, dtype=torch.int64, device=DEV)
        kn.beam_topk(logits, V, beam_scores, banned.to(DEV), -1, -1, num_beams, out_s, out_i, penalized=pen.to(DEV), penalty=penalty,
                     penalty_on_logits=on_logits)
        assert float((out_s - want_s).abs().max()) <= 2e-4, (penalty, float((out_s - want_s).abs().max()))
        gaps = (want_s[:, :-1] - want_s[:, 1:]).min()
        if float(gaps) > 1e-3:
            assert torch.equal(out_i, want_i), (penalty, out_i, want_i)
        for b in range(B):
            assert float((cand[b][out_i[b]] - out_s[b]).abs().max()) <= 2e-4


@pytest.mark.parametrize("ncand,V", [(21, 50265), (64, 50265), (41, 1000), (3, 100)])
def test_beam_topk_candidate_lists(ncand, V):
    """mmsum_beam_topk with an explicit candidate count (ABI 10; sampling asks for the top_k + 1 best of every row, num_beams = 1, no
    forced token): the ncand best post-processed logits of a row, best first, ties by lower token -- with a repetition penalty, an EOS
    ban and banned tokens -- against torch.topk on the same processed scores; more than 64 candidates, or candidate lists with several
    beams past the merge wave's capacity, are refused."""
    B = 4
    ld = (V + 127) // 128 * 128
    g = torch.Generator().manual_seed(23)
    logits = (torch.randn(B, ld, generator=g) * 3).to(DEV)
    zeros = torch.zeros(B, device=DEV)
    order = logits[:, :V].argsort(-1, descending=True).cpu()
    pen = torch.full((B, 5), -1, dtype=torch.int32)
    banned = torch.full((B, 3), -1, dtype=torch.int32)
    for r in range(B):
        pen[r, 0], pen[r, 1] = int(order[r, 0]), int(order[r, 3])
        banned[r, 0] = int(order[r, 1])
    sc = logits[:, :V].float().clone()
    for r in range(B):
        for t in pen[r].tolist():
            if t >= 0:
                sc[r, t] = sc[r, t] * 1.4 if sc[r, t] < 0 else sc[r, t] / 1.4
    sc = sc - torch.logsumexp(logits[:, :V].float(), dim=-1, keepdim=True)
    sc[:, 2] = float("-inf")                                  # the EOS ban (cur_len < min_length)
    for r in range(B):
        sc[r, int(banned[r, 0])] = float("-inf")
    want_s, want_i = torch.topk(sc, ncand, dim=1)
    out_s = torch.zeros(B, ncand, device=DEV)
    out_i = torch.zeros(B, ncand, dtype=torch.int64, device=DEV)
    kn.beam_topk(logits, V, zeros, banned.to(DEV), -1, 2, 1, out_s, out_i, penalized=pen.to(DEV), penalty=1.4, penalty_on_logits=True, ncand=ncand)
    assert float((out_s - want_s).abs().max()) <= 2e-4
    if float((want_s[:, :-1] - want_s[:, 1:]).min()) > 1e-4:
        assert torch.equal(out_i, want_i)
    for b in range(B):
        assert float((sc[b][out_i[b]] - out_s[b]).abs().max()) <= 2e-4
        assert bool((out_s[b][:-1] >= out_s[b][1:]).all())
    big_s, big_i = torch.zeros(B, 65, device=DEV), torch.zeros(B, 65, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        kn.beam_topk(logits, V, zeros, None, -1, -1, 1, big_s, big_i, ncand=65)
    if V > 1000:
        s4, i4 = torch.zeros(1, 40, device=DEV), torch.zeros(1, 40, dtype=torch.int64, device=DEV)
        with pytest.raises(RuntimeError):                     # 4 beams x 8 chunks x 40 candidates > the merge wave's 1,024 slots
            kn.beam_topk(logits, V, zeros, None, -1, -1, 4, s4, i4, ncand=40)


def test_handoff_stress():
    """The in-launch hand-offs: mmsum_dec_gemm's split-K slabs and mmsum_decode_cross_attn's entity mean (bf16 AND f32 kernels) meet in
    the LAST ARRIVER through write-through stores + a relaxed ticket, without release / acquire fences -- an ordering that rests on
    this target's code generation (csrc/check_handoff.py holds the generated code to it at build time; this test holds the hardware).
    Back-to-back launches on ONE workspace ALTERNATE BETWEEN TWO INPUT SETS (with identical inputs a last arriver that reads
    a stale partial of the previous launch gets the same value and nothing shows; in a decode loop the inputs change every step): 600
    launches of a product with 16 slices per column tile (512 one-wave workgroups: slices of a tile on different CUs and XCDs), 300 of
    each cross-attention kernel; every launch must be BIT-identical to the first run of ITS OWN input set (the reduction order is
    fixed), and those first runs must be right; a lost or stale slab, or a ticket left non-zero, shows as a mismatch."""
    bf = torch.bfloat16
    M, N, K = 32, 1024, 4096
    w, bias = rnd(N, K, dtype=bf, seed=2, std=0.05), rnd(N, seed=3)
    xs = [rnd(M, K, dtype=bf, seed=1), rnd(M, K, dtype=bf, seed=11)]
    ws = kn.dec_gemm_workspace(M, N, K, DEV)
    out = torch.empty(M, N, device=DEV, dtype=bf)
    firsts = []
    for x in xs:
        kn.dec_gemm(x, w, out, ws, bias=bias)
        ref = x.double() @ w.double().t() + bias.double()
        assert float((out.double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-5
        firsts.append(out.clone())
    assert not torch.equal(firsts[0], firsts[1])
    bad = torch.zeros((), device=DEV, dtype=torch.int64)
    for it in range(600):
        out.zero_()
        kn.dec_gemm(xs[it & 1], w, out, ws, bias=bias)
        bad += (out != firsts[it & 1]).any()
    assert int(bad) == 0, int(bad)
    B, H, qpb = 8, 16, 4
    D = H * 64
    shapes = [(8, 128), (1, 47), (4, 196)]
    rows = sum(B * n * s_ for n, s_ in shapes)
    for dt, tol in ((bf, 2e-2), (torch.float32, 1e-4)):
        sets = []
        for sd_ in (4, 14):
            q = rnd(B * qpb, D, dtype=dt, seed=sd_)
            kv = rnd(rows, 2 * D, dtype=dt, seed=sd_ + 1)
            mods, off = [], 0
            for n, s_ in shapes:
                sl = slice(off, off + B * n * s_)
                mods.append((kv[sl, :D], kv[sl, D:], None, None, n, s_))
                off += B * n * s_
            sets.append((q, kv, mods))
        o = torch.empty(3 * B * qpb, D, device=DEV, dtype=dt)
        xws = kn.decode_cross_attn_workspace(sum(B * n for n, _ in shapes), H, qpb, B, 3, DEV)
        firsts = []
        for q, kv, mods in sets:
            kn.decode_cross_attn(q, mods, o, xws, B, qpb, H, 0.125)
            # the first text entity of business 0, head 0, against a plain statement: the mean over the modality's entities of softmax(q k^T / 8) v
            kk, vv = kv[:B * 8 * 128, :D].double().view(B, 8, 128, H, 64), kv[:B * 8 * 128, D:].double().view(B, 8, 128, H, 64)
            qq = q.double().view(B, qpb, H, 64)
            ref = torch.einsum("bqnhs,bnshd->bqhd", torch.softmax(torch.einsum("bqhd,bnshd->bqnhs", qq, kk) * 0.125, -1), vv) / 8
            assert float((o[:B * qpb].double().view(B, qpb, H, 64) - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6
            firsts.append(o.clone())
        assert not torch.equal(firsts[0], firsts[1])
        bad = torch.zeros((), device=DEV, dtype=torch.int64)
        for it in range(300):
            o.zero_()
            kn.decode_cross_attn(sets[it & 1][0], sets[it & 1][2], o, xws, B, qpb, H, 0.125)
            bad += (o != firsts[it & 1]).any()
        assert int(bad) == 0, (str(dt), int(bad))
    torch.cuda.synchronize()


@pytest.mark.parametrize("bf", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("qpb", [4, 2])
def test_decode_cross_attn(qpb, bf):
    """mmsum_decode_cross_attn -- one workgroup per (entity, head) over the cached K / V of the three modalities, the entity mean through
    the last arriver -- against the per-entity softmax + entity mean of modeling_multimodalsum.py:819-869 in f64: trailing pads and a
    hole in the text keys (masked_fill -2^16), a null review, a business without a table, one with a null image; called twice (the
    tickets return to zero).  bf16 (the timed mode) and f32 (the parity mode's kernel, held to 1e-5)."""
    B, H = 3, 4
    D = H * 64
    mods_shape = [(3, 128), (1, 47), (2, 196)]
    g = torch.Generator().manual_seed(3)
    R = B * qpb
    q = rnd(R, D, dtype=bf, seed=1)
    rows = sum(B * N * S for N, S in mods_shape)
    kv = rnd(rows, 2 * D, dtype=bf, seed=2)
    pads, nulls, mods, off = [], [], [], 0
    for mi, (N, S) in enumerate(mods_shape):
        pad = torch.zeros(B, N, S, dtype=torch.bool)
        for b in range(B):
            for n in range(N):
                pad[b, n, int(torch.randint(S // 3, S + 1, (1,), generator=g)):] = True
        if mi == 0:
            pad[0, 1, 5] = True                    # a hole
            pad[1, 2] = True                       # a null review
        if mi == 1:
            pad[2, 0] = True                       # no table
        if mi == 2:
            pad[0, 1] = True                       # one null image
            pad[1] = False                         # whole images are attended
        pad_u8 = pad.to(torch.uint8).to(DEV).contiguous()
        nul = torch.empty(B * N, dtype=torch.uint8, device=DEV)
        kn.entity_null(pad_u8, nul, B * N, S)
        sl = slice(off, off + B * N * S)
        mods.append((kv[sl, :D], kv[sl, D:], pad_u8, nul, N, S))
        pads.append(pad)
        off += B * N * S
    out = torch.full((3 * R, D), float("nan"), device=DEV, dtype=bf)
    n_ent = sum(B * N for N, S in mods_shape)
    ws = kn.decode_cross_attn_workspace(n_ent, H, qpb, B, 3, DEV)
    for _ in range(2):
        out.fill_(float("nan"))
        kn.decode_cross_attn(q, mods, out, ws, B, qpb, H, 0.125)
        off = 0
        for mi, ((N, S), pad) in enumerate(zip(mods_shape, pads)):
            k = kv[off:off + B * N * S, :D].double().cpu().view(B, N, S, H, 64)
            v = kv[off:off + B * N * S, D:].double().cpu().view(B, N, S, H, 64)
            off += B * N * S
            qh = q.double().cpu().view(B, qpb, H, 64) * 0.125
            s = torch.einsum("bqhd,bnshd->bnhqs", qh, k)
            s = s.masked_fill(pad[:, :, None, None, :], -65536.0)
            p = torch.softmax(s, dim=-1)
            o = torch.einsum("bnhqs,bnshd->bnqhd", p, v)                    # [B,N,qpb,H,64]
            valid = (~pad.all(dim=-1)).double()                              # [B,N]
            cnt = valid.sum(1).clamp_min(1.0)
            ref = (o * valid[:, :, None, None, None]).sum(1) / cnt[:, None, None, None]
            got = out[mi * R:(mi + 1) * R].double().cpu().view(B, qpb, H, 64)
            err = float((got - ref).abs().max())
            if bf == torch.float32:
                assert err <= 1e-5 * float(ref.abs().max()) + 1e-6, (mi, err)
            else:
                assert err <= 1e-2 * float(ref.abs().max()) + 1e-3, (mi, err)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gate_add_ln_fwd(dtype):
    """mmsum_gate_add_ln_fwd == the gate (:732-744) followed by LN(res + .) (:474-477) in f64, rows of businesses without table / image included."""
    R, D, rpb = 12, 1024, 4
    pa, pb, yt, ytab, yimg, res = (rnd(R, D, dtype=dtype, seed=i) for i in range(6))
    gamma, beta = 1 + 0.1 * rnd(D, seed=7), 0.1 * rnd(D, seed=8)
    no_table = torch.tensor([0, 1, 0], dtype=torch.uint8, device=DEV)
    no_img = torch.tensor([1, 0, 0], dtype=torch.uint8, device=DEV)
    y = torch.empty(R, D, device=DEV, dtype=dtype)
    kn.gate_add_ln_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, res, gamma, beta, y, rpb, 1e-5)
    ma = (1 - no_table.double()).repeat_interleave(rpb)[:, None]
    mb = (1 - no_img.double()).repeat_interleave(rpb)[:, None]
    z = res.double() + yt.double() + ma * torch.relu(torch.tanh(pa.double())) * ytab.double() + mb * torch.relu(torch.tanh(pb.double())) * yimg.double()
    ref = torch.nn.functional.layer_norm(z, (D,), gamma.double(), beta.double(), 1e-5)
    close(y, ref, dtype, what="gate + add + LayerNorm")


@pytest.mark.parametrize("M,N,K,k1", [(32, 1024, 2048, 1024), (12, 1024, 2048, 1024), (8, 256, 512, 256), (32, 1024, 1024, 0)])
def test_gemm_pair(M, N, K, k1):
    """mmsum_gemm_pair: two independent products in one launch == the two products alone (f64 of the bf16 operands)."""
    bf = torch.bfloat16
    xs = [rnd(M, k1 or K, dtype=bf, seed=1), rnd(M, k1 or K, dtype=bf, seed=2)]
    x2s = [rnd(M, K - k1, dtype=bf, seed=3), rnd(M, K - k1, dtype=bf, seed=4)] if k1 else [None, None]
    ws = [rnd(N, K, dtype=bf, seed=5, std=0.05), rnd(N, K, dtype=bf, seed=6, std=0.05)]
    bs = [rnd(N, seed=7), rnd(N, seed=8)]
    outs = [torch.full((M, N), float("nan"), device=DEV, dtype=bf) for _ in range(2)]
    kn.gemm_pair(xs, x2s, ws, outs, bs)
    for i in range(2):
        x = torch.cat([xs[i], x2s[i]], dim=1) if k1 else xs[i]
        ref = x.double() @ ws[i].double().t() + bs[i].double()
        assert float((outs[i].double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-5, i


@pytest.mark.parametrize("n,H,C,Cout", [(5, 14, 256, 256), (3, 28, 128, 128), (2, 56, 64, 64), (37, 14, 256, 256), (1, 7, 64, 136)],
                         ids=["layer3_w4", "layer2_ring", "layer1_ring", "layer3_several_tiles", "odd"])
def test_conv3x3_implicit_gemm(n, H, C, Cout):
    """mmsum_conv3x3_gemm (the NT kernels' DMA pieces read one tap's channels of the zero-bordered padded activations) == the im2col
    matrix times the same weights through mmsum_gemm: the two take the same kernel, tile and order of summation, so the outputs and the
    BatchNorm statistics of the epilogue are compared element-wise with a bf16 bound AND against an f64 convolution; the padded layout
    comes from mmsum_bn_apply (pad_H / pad_W), whose borders must stay zero."""
    bf = torch.bfloat16
    W = H
    R = n * H * W
    x = rnd(R, C, dtype=bf, seed=1)                                 # conv1's output (compact)
    sums = torch.cat([0.1 * rnd(C, seed=2), 0.5 + rnd(C, seed=3).abs()])
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    yc = torch.empty(R, C, device=DEV, dtype=bf)
    kn.bn_apply(x, sums, gamma, beta, None, yc, None, None, 1e-5, 0.1, True, True)
    yp = torch.zeros(n * (H + 2) * (W + 2), C, device=DEV, dtype=bf)
    kn.bn_apply(x, sums, gamma, beta, None, yp, None, None, 1e-5, 0.1, True, True, pad_hw=(H, W))
    img = yp.view(n, H + 2, W + 2, C)
    assert torch.equal(img[:, 1:-1, 1:-1].reshape(R, C), yc)
    assert float(img[:, 0].abs().max()) == 0 and float(img[:, -1].abs().max()) == 0 and float(img[:, :, 0].abs().max()) == 0 and float(img[:, :, -1].abs().max()) == 0
    K = 9 * C
    wm = rnd(Cout, K, dtype=bf, seed=6, std=0.03)
    col = torch.empty(R, K, device=DEV, dtype=bf)
    kn.im2col(yc, col, n, H, W, C, 3, 3, 1, 1, H, W, K)
    ref_out, ref_raw = torch.empty(R, Cout, device=DEV, dtype=bf), torch.zeros(2 * Cout, device=DEV)
    kn.gemm(col, wm, ref_out, colsum=ref_raw, colsum_sq=True)
    out, raw = torch.full((R, Cout), float("nan"), device=DEV, dtype=bf), torch.zeros(2 * Cout, device=DEV)
    kn.conv3x3_gemm(yp, wm, out, n, H, W, C, stats=raw)
    exact = col.double() @ wm.double().t()
    tol_ = 1e-2 * float(exact.abs().max())
    assert float((out.double() - exact).abs().max()) <= tol_
    assert float((out.double() - ref_out.double()).abs().max()) <= 2.0 ** -7 * float(exact.abs().max())
    assert float((raw - ref_raw).abs().max()) <= 1e-3 * float(ref_raw.abs().max()) + 1e-3
    out2 = torch.empty(R, Cout, device=DEV, dtype=bf)
    kn.conv3x3_gemm(yp, wm, out2, n, H, W, C)                       # without the statistics
    assert float((out2.double() - exact).abs().max()) <= tol_
    # the backward kernels read the ReLU mask from the padded forward output
    dy = rnd(R, C, dtype=bf, seed=7)
    d0, d1 = torch.empty(2 * C, device=DEV), torch.empty(2 * C, device=DEV)
    kn.bn_bwd_reduce(dy, yc, x, sums, d0, 1e-5, True)
    kn.bn_bwd_reduce(dy, yp, x, sums, d1, 1e-5, True, pad_hw=(H, W))
    assert torch.equal(d0, d1)
    dx0, dx1 = torch.empty(R, C, device=DEV, dtype=bf), torch.empty(R, C, device=DEV, dtype=bf)
    kn.bn_bwd_apply(dy, yc, x, sums, d0, gamma, dx0, None, None, None, 1e-5, True)
    kn.bn_bwd_apply(dy, yp, x, sums, d0, gamma, dx1, None, None, None, 1e-5, True, pad_hw=(H, W))
    assert torch.equal(dx0, dx1)
    # ... and can write dx in the padded layout (interior only: the borders keep the caller's zeros)
    dxp = torch.zeros(n * (H + 2) * (W + 2), C, device=DEV, dtype=bf)
    kn.bn_bwd_apply(dy, yp, x, sums, d0, gamma, dxp, None, None, None, 1e-5, True, pad_hw=(H, W), dx_pad_hw=(H, W))
    dimg = dxp.view(n, H + 2, W + 2, C)
    assert torch.equal(dimg[:, 1:-1, 1:-1].reshape(R, C), dx0)
    assert float(dimg[:, 0].abs().max()) == 0 and float(dimg[:, -1].abs().max()) == 0 and float(dimg[:, :, 0].abs().max()) == 0 and float(dimg[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("n,H,C,Cout,sk", [(5, 14, 256, 256, 1), (9, 14, 256, 256, 4), (3, 7, 512, 128, 3), (212, 14, 256, 256, 28)])
def test_conv3x3_backward_without_im2col(n, H, C, Cout, sk):
    """The backward of the implicit 3x3 convolution: mmsum_conv3x3_wgrad (four-wave reduction-major kernel over the two PADDED images, the x
    rows shifted by the tile's tap) against dy^T . im2col(x) in f64, with split-K slabs; the input gradient as mmsum_conv3x3_gemm of the
    padded dy with the rotated weights (mmsum_conv_weight_permute, mode 2) against torch's conv2d input gradient in f64."""
    bf = torch.bfloat16
    W = H
    R, Rp = n * H * W, n * (H + 2) * (W + 2)
    xc = rnd(R, C, dtype=bf, seed=11)
    dyc = rnd(R, Cout, dtype=bf, seed=12)
    pad = lambda t, ch: torch.nn.functional.pad(t.view(n, H, W, ch), (0, 0, 1, 1, 1, 1)).reshape(Rp, ch).contiguous()
    xp, dyp = pad(xc, C), pad(dyc, Cout)
    # weight gradient
    col = torch.empty(R, 9 * C, device=DEV, dtype=bf)
    kn.im2col(xc, col, n, H, W, C, 3, 3, 1, 1, H, W, 9 * C)
    exact = dyc.double().t() @ col.double()
    ws = torch.full((sk * Cout, 9 * C), float("nan"), device=DEV)
    kn.conv3x3_wgrad(dyp, xp, ws, n, H, W, C, sk)
    got = ws.view(sk, Cout, 9 * C).double().sum(0)
    assert float((got - exact).abs().max()) <= 2e-5 * float(exact.abs().max()) + 1e-4 * (R ** 0.5)       # f32 accumulation of exact bf16 products
    if sk > 1:
        dwm = torch.empty(Cout, 9 * C, device=DEV)
        kn.slab_reduce(ws, sk, dwm, accumulate=False)
        assert float((dwm.double() - exact).abs().max()) <= 2e-5 * float(exact.abs().max()) + 1e-4 * (R ** 0.5)
    # input gradient
    w4 = rnd(Cout, C, 3, 3, seed=13, std=0.03)
    wr = torch.empty(C, 9 * Cout, device=DEV, dtype=bf)
    kn.conv_weight_to_dgrad_matrix(wr, w4, Cout, C, 3, 3, 9 * Cout)
    ref_wr = w4.flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * Cout).to(bf)
    assert torch.equal(wr, ref_wr)
    dx = torch.full((R, C), float("nan"), device=DEV, dtype=bf)
    kn.conv3x3_gemm(dyp, wr, dx, n, H, W, Cout)
    wq = w4.to(bf).double()                                        # the kernel multiplies bf16 weights
    ref = torch.nn.functional.conv_transpose2d(dyc.double().view(n, H, W, Cout).permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1).reshape(R, C)
    assert float((dx.double() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
