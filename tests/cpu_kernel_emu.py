"""TEST INFRASTRUCTURE ONLY -- a torch-CPU stand-in for the C-ABI kernels.

The development container has no GPU, and the host-side schedules in multimodalsum_amd/engine.py
(which kernel runs when, on which arena view, what each backward accumulates into) are ~1000 lines
of index arithmetic.  This module re-states the *contract* of every function of
multimodalsum_amd.kernels with plain torch ops so that tests/test_host_logic_cpu.py can drive the
real engine/modules/optimiser code on CPU tensors and compare it with the oracle.  It is never
imported by the product package; the product raises on CPU tensors (tests/test_abi_cpu.py).
"""
import math

import torch
import torch.nn.functional as F

EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_RELU, EPI_RELU_BWD = 0, 1, 2, 3, 4


def _keep_mask(shape, p, seed, salt=None):
    """The device's masks (multimodalsum_amd/dropout.py restates the kernels' counter hash): a function of the seed, the salt and
    row * D + column, so a live-row prefix of a matrix draws the prefix of the full mask."""
    if p <= 0:
        return torch.ones(shape)
    from multimodalsum_amd.dropout import keep_mask
    return keep_mask(seed, int(shape[0]), int(shape[1]), p, None if salt is None else int(salt)).float() / (1.0 - p)


def gemm_colsum_fusable(a, a_t=False, b_t=False, a2=None):
    return a.dtype == torch.bfloat16 and not a_t and not b_t and a2 is None and a.shape[1] % 64 == 0


def gemm_tn_colsum_ok(dy, x, ws, splitk, colsum):
    return dy.dtype == torch.bfloat16 and splitk >= 2


def _n(live, rows):
    """Live row count of a kernel call: the device scalar when given (clamped to the capacity), else every row."""
    return rows if live is None else max(0, min(rows, int(live.reshape(-1)[0])))


def gemm(a, b, out, a_t=False, b_t=False, bias=None, epi=EPI_NONE, aux=None, accumulate=False, alpha=1.0, a2=None, splitk=1, slabs=False,
         colsum=None, live=None, alpha_dev=None, colsum_sq=False):
    """`live` as in include/mmsum_hip.h: natural A -> only the first `live` rows of A / aux / out take part; a_t & b_t ->
    only the first `live` reduction rows; rows past it are neither read nor written."""
    if alpha_dev is not None:
        alpha = alpha * float(alpha_dev.reshape(-1)[0])
    if live is not None and a_t and b_t:
        n = _n(live, a.shape[0])
        a, b, live = a[:n], b[:n], None
    if colsum is not None and a_t and b_t:       # weight-gradient product: column sums of A (the bias gradient of the same Linear)
        colsum.add_(alpha * a.float().sum(0))
        colsum = None
    if live is not None:
        assert not a_t and not slabs
        n = _n(live, a.shape[0])
        return gemm(a[:n], b, out[:n], False, b_t, bias, epi, None if aux is None else aux[:n], accumulate, alpha,
                    None if a2 is None else a2[:n], splitk, False, colsum, colsum_sq=colsum_sq)
    A = a.float().t() if a_t else a.float()
    if a2 is not None:
        A = torch.cat([A, a2.float()], dim=1)
    Bm = b.float().t() if b_t else b.float()
    v = alpha * (A @ Bm.t())
    if bias is not None:
        v = v + bias
    if epi == EPI_GELU:
        if aux is not None:
            aux.copy_(v)
        v = F.gelu(v)
    elif epi == EPI_GELU_BWD:
        u = aux.float()
        v = v * (0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))
    elif epi == EPI_RELU:
        v = F.relu(v)
    elif epi == EPI_RELU_BWD:
        v = v * (aux.float() > 0)
    if slabs:        # split-K partial slabs: emulate as 'slab 0 holds everything, the rest are zero'
        out.zero_()
        out[:v.shape[0]].copy_(v)
        return out
    if accumulate:
        out.add_(v.to(out.dtype))
    else:
        out.copy_(v)
    if colsum is not None:
        N = out.shape[1]
        colsum[:N].add_(out.float().sum(0))
        if colsum_sq:
            colsum[N:2 * N].add_((out.float() ** 2).sum(0))
    return out


def slab_reduce(ws, nslabs, out, accumulate=True):
    rows = out.shape[0]
    s = ws.view(nslabs, rows, -1).sum(0)
    if accumulate:
        out.add_(s)
    else:
        out.copy_(s)


def colsum(x, out, accumulate=False, live=None):
    s = x[:_n(live, x.shape[0])].float().sum(0)
    out.copy_(out + s if accumulate else s)
    return out


def embed_ln_fwd(ids, E, P, rating_diff, rvec, gamma, beta, y, mean, rstd, nseq, T, pos_offset, eps, p_drop, seed, salt=None):
    D = E.shape[1]
    z = E.float()[ids.reshape(nseq, T)] + P.float()[torch.arange(T) + pos_offset]
    if rating_diff is not None:
        z = z + rating_diff.view(nseq, 1, 1) * rvec.float()
    z = z.view(nseq * T, D)
    mu = z.mean(-1)
    rs = (z.var(-1, unbiased=False) + eps).rsqrt()
    mean.copy_(mu)
    rstd.copy_(rs)
    o = (z - mu[:, None]) * rs[:, None] * gamma + beta
    y.copy_(o * _keep_mask(o.shape, p_drop, seed, salt))


def embed_ln_bwd(dy, ids, E, P, rating_diff, rvec, gamma, mean, rstd, dE, dP, drvec, dgamma, dbeta, nseq, T, pos_offset, pad_id,
                 p_drop, seed, salt=None):
    D = E.shape[1]
    z = E.float()[ids.reshape(nseq, T)] + P.float()[torch.arange(T) + pos_offset]
    if rating_diff is not None:
        z = z + rating_diff.view(nseq, 1, 1) * rvec.float()
    z = z.view(nseq * T, D)
    g = dy.float() * _keep_mask(dy.shape, p_drop, seed, salt)
    xh = (z - mean[:, None]) * rstd[:, None]
    gdy = g * gamma
    dz = rstd[:, None] * (gdy - gdy.mean(-1, keepdim=True) - xh * (gdy * xh).mean(-1, keepdim=True))
    dgamma.add_((g * xh).sum(0))
    dbeta.add_(g.sum(0))
    flat = ids.reshape(-1)
    keep = flat != pad_id
    dE.index_add_(0, flat[keep], dz[keep])
    dP.index_add_(0, (torch.arange(T) + pos_offset).repeat(nseq), dz)
    if rating_diff is not None:
        drvec.add_((dz.view(nseq, T, D) * rating_diff.view(nseq, 1, 1)).sum((0, 1)))


def add_ln_fwd(x, res, gamma, beta, y, mean, rstd, eps, p_drop, seed, salt=None, live=None, y_f32=None):
    if live is not None:
        n = _n(live, x.shape[0])
        return add_ln_fwd(x[:n], res[:n], gamma, beta, y[:n], mean[:n], rstd[:n], eps, p_drop, seed, salt)
    z = x.float() * _keep_mask(x.shape, p_drop, seed, salt) + res.float()
    mu = z.mean(-1)
    rs = (z.var(-1, unbiased=False) + eps).rsqrt()
    mean.copy_(mu)
    rstd.copy_(rs)
    y.copy_((z - mu[:, None]) * rs[:, None] * gamma + beta)
    if y_f32 is not None:
        y_f32.copy_((z - mu[:, None]) * rs[:, None] * gamma + beta)


def add_ln_bwd(dy, x, res, gamma, mean, rstd, dx, dres, accumulate_dres, dgamma, dbeta, p_drop, seed, dxsum=None, salt=None, live=None):
    if live is not None:
        n = _n(live, x.shape[0])
        return add_ln_bwd(dy[:n], x[:n], res[:n], gamma, mean[:n], rstd[:n], dx[:n], dres[:n], accumulate_dres, dgamma, dbeta, p_drop,
                          seed, dxsum, salt)
    km = _keep_mask(x.shape, p_drop, seed, salt)
    z = x.float() * km + res.float()
    xh = (z - mean[:, None]) * rstd[:, None]
    g = dy.float()
    gdy = g * gamma
    dz = rstd[:, None] * (gdy - gdy.mean(-1, keepdim=True) - xh * (gdy * xh).mean(-1, keepdim=True))
    dgamma.add_((g * xh).sum(0))
    dbeta.add_(g.sum(0))
    dx.copy_(dz * km)
    if dxsum is not None:
        dxsum.add_((dz * km).sum(0))
    if accumulate_dres:
        dres.add_(dz)
    else:
        dres.copy_(dz)


class _Desc:
    pass


def make_attn_desc(q, k, v, out, pad, null_entity, n_qblocks, T, qpb, N, S, H, exclude_self, causal, scale, q_rows=None, kv_rows=None, causal_q0=0):
    d = _Desc()
    d.__dict__.update(q=q, k=k, v=v, out=out, pad=pad, null=null_entity, nq=n_qblocks, T=T, qpb=qpb, N=N, S=S, H=H,
                      excl=bool(exclude_self), causal=bool(causal), scale=scale, q_rows=q_rows, kv_rows=kv_rows, q0=int(causal_q0))
    return d


def _expand(t, rows):
    """Logical (padded) view of a compact matrix through a row map: absent rows read as zeros."""
    if rows is None:
        return t
    sel = rows >= 0
    out = torch.zeros(rows.numel(), t.shape[1], dtype=t.dtype)
    out[sel] = t[rows[sel].long()]
    return out


def _scatter(dst, src, rows, accumulate=False):
    """Write the existing rows of the logical matrix `src` to their physical rows of `dst`; nothing else is touched."""
    if rows is None:
        dst.add_(src) if accumulate else dst.copy_(src)
        return
    sel = rows >= 0
    idx = rows[sel].long()
    if accumulate:
        dst[idx] = (dst[idx].float() + src[sel].float()).to(dst.dtype)
    else:
        dst[idx] = src[sel].to(dst.dtype)


def entity_null(pad, null_entity, n_entities, S):
    null_entity.copy_(pad.reshape(n_entities, S).bool().all(-1).to(torch.uint8))


def _attn_ref(d, q, k, v):
    B = d.nq // d.qpb
    qh = q.reshape(d.nq, d.T, d.H, 64).permute(0, 2, 1, 3)
    kh = k.reshape(B, d.N, d.S, d.H, 64).permute(0, 1, 3, 2, 4)
    vh = v.reshape(B, d.N, d.S, d.H, 64).permute(0, 1, 3, 2, 4)
    pad = d.pad.reshape(B, d.N, d.S).bool() if d.pad is not None else None
    null = d.null.reshape(B, d.N).bool() if d.null is not None else None
    outs = []
    for qb in range(d.nq):
        b, i = qb // d.qpb, qb % d.qpb
        acc, cnt = torch.zeros(d.H, d.T, 64), 0
        for n in range(d.N):
            if (d.excl and n == i) or (null is not None and bool(null[b, n])):
                continue
            s = torch.einsum("htd,hsd->hts", qh[qb], kh[b, n]) * d.scale
            if pad is not None:
                s = s.masked_fill(pad[b, n][None, None, :], float("-inf"))
            if d.causal:
                s = s + torch.triu(torch.full((d.T, d.S), float("-inf")), 1 + d.q0)          # key s masked when s > causal_q0 + t
            acc = acc + torch.einsum("hts,hsd->htd", torch.softmax(s, -1), vh[b, n])
            cnt += 1
        outs.append(acc / max(cnt, 1) + 0 * qh[qb])
    return torch.stack(outs).permute(0, 2, 1, 3).reshape(d.nq * d.T, d.H * 64)


def attn_fwd(d, dtype_tensor):
    if (d.q_rows is not None or d.kv_rows is not None) and d.q.dtype != torch.bfloat16:
        raise RuntimeError("mmsum_attn_fwd: row maps are a bf16 feature")
    with torch.no_grad():
        o = _attn_ref(d, _expand(d.q, d.q_rows).float(), _expand(d.k, d.kv_rows).float(), _expand(d.v, d.kv_rows).float())
        _scatter(d.out, o, d.q_rows)


def attn_bwd_workspace(d):
    return 16


def attn_bwd(d, dout, dq, accumulate_dq, dk, dv, stats):
    q, k, v = (t.float().clone().requires_grad_(True) for t in (_expand(d.q, d.q_rows), _expand(d.k, d.kv_rows), _expand(d.v, d.kv_rows)))
    with torch.enable_grad():
        o = _attn_ref(d, q, k, v)
        gq, gk, gv = torch.autograd.grad(o, (q, k, v), _expand(dout, d.q_rows).float(), allow_unused=True)
    z = lambda g, t: torch.zeros_like(t) if g is None else g  # noqa: E731
    _scatter(dq, z(gq, q), d.q_rows, accumulate=bool(accumulate_dq))
    _scatter(dk, z(gk, k), d.kv_rows)
    _scatter(dv, z(gv, v), d.kv_rows)


def gate_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, out, rows_per_b):
    ma = (1 - no_table.float()).repeat_interleave(rows_per_b)[:, None]
    mb = (1 - no_img.float()).repeat_interleave(rows_per_b)[:, None]
    out.copy_(yt.float() + ma * F.relu(torch.tanh(pa.float())) * ytab.float() + mb * F.relu(torch.tanh(pb.float())) * yimg.float())


def gate_bwd(dout, pa, pb, ytab, yimg, no_table, no_img, dpa, dpb, dyt, dytab, dyimg, rows_per_b, sums=None):
    ma = (1 - no_table.float()).repeat_interleave(rows_per_b)[:, None]
    mb = (1 - no_img.float()).repeat_interleave(rows_per_b)[:, None]
    g = dout.float()
    ta, tb = torch.tanh(pa.float()), torch.tanh(pb.float())
    dytab.copy_(g * ma * F.relu(ta))
    dyimg.copy_(g * mb * F.relu(tb))
    dpa.copy_(torch.where(ta > 0, ma * g * ytab.float() * (1 - ta * ta), torch.zeros_like(g)))
    dpb.copy_(torch.where(tb > 0, mb * g * yimg.float() * (1 - tb * tb), torch.zeros_like(g)))
    dyt.copy_(g)
    if sums is not None:
        sums[0].add_(dpa.float().sum(0))
        sums[1].add_(dpb.float().sum(0))


def ls_loss(logits, target, row_loss, V, smoothing, gscale, write_grad=True):
    x = logits[:, :V].float()
    logp = x.log_softmax(-1)
    eps_p = smoothing / (V - 1) if V > 1 else 0.0
    td = torch.full_like(logp, eps_p)
    td.scatter_(1, target.reshape(-1, 1), 1.0 - smoothing)
    row_loss.copy_(-(td * logp).sum(-1))
    if write_grad:
        logits.zero_()
        logits[:, :V] = gscale * (logp.exp() - td)


def segment_sum(x, out, nseg, seg, scale):
    out.copy_(x[:nseg * seg].double().view(nseg, seg).sum(-1).float() * scale)


def l2norm_sq(g, out, accumulate=False):
    s = (g.double() ** 2).sum().float()
    out.copy_(out + s if accumulate else s.reshape(1))


def _clip(norm_sq, max_norm):
    if norm_sq is None or max_norm <= 0:
        return 1.0
    return min(1.0, max_norm / (float(norm_sq.sqrt()) + 1e-6))


def adamw(p, g, m, v, shadow, hyper, norm_sq, beta1, beta2, eps):
    step_size, lr_wd, max_norm = float(hyper[0]), float(hyper[1]), float(hyper[2])
    gg = g * _clip(norm_sq, max_norm)
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    p.sub_(step_size * m / (v.sqrt() + eps))
    p.sub_(lr_wd * p)
    if shadow is not None:
        shadow.copy_(p)


def cast(dst, src):
    dst.copy_(src)
    return dst


def transpose(src, dst, rows_pad=None, colsum=None):
    dst.zero_()
    dst[:, :src.shape[0]] = src.t()
    if colsum is not None:
        colsum.add_(src.float().sum(0))
    return dst


def transpose_batched(src_base, dst_base, desc, n, max_tiles):
    for i in range(n):
        so, do, r, c, ls, ld = [int(v) for v in desc[i]]
        src = torch.as_strided(src_base, (r, c), (ls, 1), so)
        torch.as_strided(dst_base, (c, r), (ld, 1), do).copy_(src.t())


def scale_by_clip(g, norm_sq, max_norm):
    g.mul_(_clip(norm_sq, max_norm))


class ImagePlan:
    """CPU twin of kernels.ImagePlan (mmsum_image_plan's outputs)."""

    def __init__(self, n, positions, row_kinds, device):
        self.n, self.positions, self.row_kinds = n, positions, list(row_kinds)
        self.plan = torch.zeros(4 + len(self.row_kinds), dtype=torch.int32)
        self.src = torch.zeros(n, dtype=torch.int32)
        self.slot_rows = torch.zeros(n * positions, dtype=torch.int64)
        self.run_rows = torch.zeros(n * positions, dtype=torch.int64)

    def rows(self, rpi, adjust=0):
        k = self.row_kinds.index((rpi, adjust))
        return self.plan[4 + k:5 + k]


def image_plan(img, mask, ip):
    """Contract of mmsum_image_plan: empty = masked AND all zero; run order = the non-empty slots in batch order, then the first empty
    slot as the representative of all of them."""
    n, P = ip.n, ip.positions
    empty = [bool(mask.reshape(-1)[i] == 0) and not bool(img[i].ne(0).any()) for i in range(n)]
    live = [i for i in range(n) if not empty[i]]
    mult = sum(empty)
    rep = len(live) if mult else -1
    nrun = len(live) + (1 if mult else 0)
    ip.src.zero_()
    ip.src[:len(live)] = torch.tensor(live, dtype=torch.int32)
    if mult:
        ip.src[rep] = empty.index(True)
    ip.plan[0], ip.plan[1], ip.plan[2], ip.plan[3] = nrun, rep, max(mult, 1), len(live)
    for k, (r, adj) in enumerate(ip.row_kinds):
        ip.plan[4 + k] = max(0, nrun * r + adj)
    run_of = {s_: r for r, s_ in enumerate(live)}
    ar = torch.arange(P)
    for a in range(n):
        ip.slot_rows[a * P:(a + 1) * P] = (rep if empty[a] else run_of[a]) * P + ar
        ip.run_rows[a * P:(a + 1) * P] = (live[a] * P + ar) if a < len(live) else -1
    return ip


def _win(images, R):
    """(rows that run, first row of the representative, multiplicity) of an [R, .] matrix under the live-image window."""
    if images is None:
        return R, R, 1.0
    rpi = R // images.n
    nrun, rep, mult = (int(v) for v in images.plan[:3])
    return min(R, nrun * rpi), (rep * rpi if rep >= 0 else R), (float(mult) if rep >= 0 else 1.0)


def _nimg(images, N):
    return N if images is None else min(N, int(images.plan[0]))


def im2col(x, col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images=None):
    if images is not None:
        n = _nimg(images, N)
        return im2col(x[:n * H * W], col[:n * Ho * Wo], n, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad) if n else None
    xi = x.float().view(N, H, W, C).permute(0, 3, 1, 2)
    u = F.unfold(xi, (KH, KW), padding=pad, stride=stride)              # [N, C*KH*KW, L], rows (c, kh, kw)
    u = u.view(N, C, KH * KW, Ho * Wo).permute(0, 3, 2, 1).reshape(N * Ho * Wo, KH * KW * C)
    col.zero_()
    col[:, :KH * KW * C] = u


def col2im(dcol, dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images=None):
    if images is not None:
        n = _nimg(images, N)
        return col2im(dcol[:n * Ho * Wo], dx[:n * H * W], n, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad) if n else None
    u = dcol[:, :KH * KW * C].float().view(N, Ho * Wo, KH * KW, C).permute(0, 3, 2, 1).reshape(N, C * KH * KW, Ho * Wo)
    xi = F.fold(u, (H, W), (KH, KW), padding=pad, stride=stride)
    dx.copy_(xi.permute(0, 2, 3, 1).reshape(N * H * W, C))


def conv_weight_to_matrix(matrix, weight, Cout, Cin, KH, KW, Kpad):
    matrix.zero_()
    matrix[:, :KH * KW * Cin] = weight.view(Cout, Cin, KH, KW).permute(0, 2, 3, 1).reshape(Cout, -1)


def conv_matrix_grad_to_weight(matrix_f32, dweight, Cout, Cin, KH, KW, Kpad, accumulate):
    g = matrix_f32[:, :KH * KW * Cin].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
    if accumulate:
        dweight.add_(g)
    else:
        dweight.copy_(g)


def bn_reduce(x, sums, images=None):
    R, C = x.shape
    rows, rep0, mult = _win(images, R)
    w = torch.ones(rows, 1)
    w[rep0:] = mult                                  # the representative's rows count `mult` times; the count stays all R rows
    xf = x[:rows].float()
    mean = (w * xf).sum(0) / R
    sums[:C] = mean
    sums[C:] = (w * (xf - mean) ** 2).sum(0) / R


def bn_rep_fix(y, raw, images):
    R, C = y.shape
    rows, rep0, mult = _win(images, R)
    if rep0 >= rows or mult <= 1:
        return
    yr = y[rep0:min(rows, rep0 + R // images.n)].float()
    raw[:C] += (mult - 1) * yr.sum(0)
    raw[C:2 * C] += (mult - 1) * (yr ** 2).sum(0)


def bn_stats_from_sums(raw, R, sums, running_mean, running_var, momentum):
    C = sums.numel() // 2
    mean = raw[:C] / R
    var = (raw[C:2 * C] / R - mean * mean).clamp_min(0)
    sums[:C], sums[C:] = mean, var
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * (R / (R - 1) if R > 1 else 1.0))


def _bn_stats(sums, R, C, eps):
    mean, var = sums[:C], sums[C:]
    return mean, var, (var + eps).rsqrt()


def _pad_rows(R, pad_hw):
    """Rows of the pixels of [n, H, W] in the zero-bordered padded layout [n, H+2, W+2] (mmsum_bn_apply's pad_H / pad_W)."""
    H, W = pad_hw
    r = torch.arange(R)
    n, rem = r // (H * W), r % (H * W)
    return n * (H + 2) * (W + 2) + (rem // W + 1) * (W + 2) + rem % W + 1


def bn_apply(x, sums, gamma, beta, residual, y, running_mean, running_var, eps, momentum, relu, training, pad_hw=None, raw=None, images=None):
    R, C = x.shape
    rows = _win(images, R)[0]
    if raw is not None:                            # the statistics from the GEMM epilogue's plain sums; `sums` is written here
        assert training
        sums[:C] = raw[:C] / R
        sums[C:] = (raw[C:] / R - sums[:C] ** 2).clamp_min(0.0)
    if pad_hw is not None:
        yc = torch.empty(R, C, dtype=y.dtype)
        bn_apply(x, sums, gamma, beta, residual, yc, running_mean, running_var, eps, momentum, relu, training, images=images)
        y[_pad_rows(R, pad_hw)[:rows]] = yc[:rows]               # the borders keep what the caller put there (zeros)
        return
    if training:
        mean, var, rstd = _bn_stats(sums, R, C, eps)
    else:
        mean, var = running_mean, running_var
        rstd = (var + eps).rsqrt()
    o = (x[:rows].float() - mean) * rstd * gamma + beta
    if residual is not None:
        o = o + residual[:rows].float()
    y[:rows].copy_(F.relu(o) if relu else o)       # rows of images that do not run are neither read nor written
    if training and running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * (R / (R - 1) if R > 1 else 1.0))


def bn_bwd_reduce(dy, y, x, sums, dsums, eps, relu, pad_hw=None, images=None):
    R, C = x.shape
    rows = _win(images, R)[0]
    if pad_hw is not None:
        y = y[_pad_rows(R, pad_hw)]
    mean, var, rstd = _bn_stats(sums, R, C, eps)
    g = dy[:rows].float() * ((y[:rows].float() > 0) if relu else 1.0)     # (the representative's rows arrive multiplied: no weights here)
    dsums[:C] = g.sum(0)
    dsums[C:] = (g * (x[:rows].float() - mean) * rstd).sum(0)


def bn_bwd_apply(dy, y, x, sums, dsums, gamma, dx, dresidual, dgamma, dbeta, eps, relu, pad_hw=None, dx_pad_hw=None, images=None):
    R, C = x.shape
    rows, rep0, mult = _win(images, R)
    if pad_hw is not None:
        y = y[_pad_rows(R, pad_hw)]
    mean, var, rstd = _bn_stats(sums, R, C, eps)
    g = dy[:rows].float() * ((y[:rows].float() > 0) if relu else 1.0)
    xh = (x[:rows].float() - mean) * rstd
    w = torch.ones(rows, 1)
    w[rep0:] = mult                 # the representative's gradient rows travel multiplied by its multiplicity: so do the batch-mean terms
    val = gamma * rstd * (g - w * (dsums[:C] / R + xh * dsums[C:] / R))
    if dx_pad_hw is not None:
        dx[_pad_rows(R, dx_pad_hw)[:rows]] = val.to(dx.dtype)          # interior only: the borders are the caller's zeros
    else:
        dx[:rows].copy_(val)
    if dresidual is not None:
        dresidual[:rows].copy_(g)
    if dgamma is not None:
        dbeta.add_(dsums[:C])
        dgamma.add_(dsums[C:])


def conv3x3_gemm(xp, w, y, n, H, W, C, stats=None, live=None):
    """Contract of mmsum_conv3x3_gemm: the im2col matrix of the padded image (an (H+2) x (W+2) image, padding 0) times w^T."""
    Kpad = w.shape[1]
    if live is not None:
        rows = _n(live, n * H * W)
        assert rows % (H * W) == 0
        n = rows // (H * W)
        if n == 0:
            return y
        xp, y = xp[:n * (H + 2) * (W + 2)], y[:rows]
    col = torch.zeros(n * H * W, Kpad, dtype=xp.dtype)
    im2col(xp, col, n, H + 2, W + 2, C, 3, 3, 1, 0, H, W, Kpad)
    gemm(col, w, y, colsum=stats, colsum_sq=stats is not None)
    return y


def conv3x3_wgrad(dyp, xp, out, n, H, W, C, splitk=1, live=None):
    """Contract of mmsum_conv3x3_wgrad, computed the way the kernel does: over ALL padded positions but the first / last W + 3, the x rows
    shifted by the tap's offset; the k range cut into `splitk` slabs."""
    Cout = dyp.shape[1]
    Wp, skip = W + 2, W + 3
    K = _n(live, dyp.shape[0] - 2 * skip)
    nst = (K + 31) // 32
    per = (nst + splitk - 1) // splitk
    for s_ in range(splitk):
        k0, k1 = min(K, s_ * per * 32), min(K, (s_ + 1) * per * 32)
        a = dyp[skip + k0:skip + k1].float()
        for tap in range(9):
            sh = (tap // 3 - 1) * Wp + (tap % 3 - 1)
            out[s_ * Cout:(s_ + 1) * Cout, tap * C:(tap + 1) * C] = a.t() @ xp[skip + sh + k0:skip + sh + k1].float()
    return out


def conv_weight_to_dgrad_matrix(matrix, weight, Cout, Cin, KH, KW, Kpad):
    matrix.zero_()
    matrix[:, :KH * KW * Cout] = weight.view(Cout, Cin, KH, KW).flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, -1)


def maxpool3x3s2(x, y, N, H, W, C, Ho, Wo, images=None):
    n = _nimg(images, N)
    if n:
        y[:n * Ho * Wo].copy_(F.max_pool2d(x[:n * H * W].float().view(n, H, W, C).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, C))


def nchw_to_nhwc(x, y, N, C, H, W, images=None):
    if images is None:
        y.copy_(x.permute(0, 2, 3, 1).reshape(-1, C))
        return
    n = _nimg(images, N)
    y[:n * H * W].copy_(x[images.src[:n].long()].permute(0, 2, 3, 1).reshape(-1, C))


def table_gather(E, field, fv, w_rating, w_hours, out, mask, B, pad_id):
    name, category, str_cat, str_bool, rating, hours = fv
    Ef = E.float()

    def msum(ids, dim):
        return (Ef[ids] * ids.ne(pad_id).unsqueeze(-1).float()).sum(dim=dim)

    names = msum(field, 1).unsqueeze(0).expand(B, -1, -1)
    cat_valid = category.ne(pad_id).any(-1).unsqueeze(-1).float()
    vals = torch.cat([msum(name, 1).unsqueeze(1),
                      (msum(category, 2) * cat_valid).sum(1, keepdim=True) / (cat_valid.sum(1, keepdim=True) + 1e-6),
                      msum(str_cat, 2), Ef[str_bool.squeeze(-1)] * str_bool.ne(pad_id).float(),
                      F.linear(rating.float(), w_rating.float()).unsqueeze(1), F.linear(hours.float(), w_hours.float())], 1)
    out.copy_(torch.cat([names, vals], -1).reshape(B * 47, -1))
    ones = torch.ones(B, 1, dtype=torch.bool)
    m = torch.cat([ones, category[:, :1, 0].ne(pad_id), str_cat[:, :, 0].ne(pad_id), str_bool[:, :, 0].ne(pad_id), ones,
                   hours.sum(-1) != 0], 1)
    mask.copy_(m.to(torch.uint8))


def table_gather_bwd(dall, rating, hours, dw_rating, dw_hours, B, D):
    dv = dall.float().view(B, 47, 2 * D)[:, :, D:]
    dw_rating.add_(torch.einsum("bk,bd->dk", rating.float(), dv[:, 39]))
    dw_hours.add_(torch.einsum("bjk,bjd->dk", hours.float(), dv[:, 40:47]))


def amazon_table_gather(E, field, fv, w_price, w_rating, out, mask, B, pad_id):
    price, rating, brand, name, category, description = fv
    Ef = E.float()

    def msum(ids, dim):
        return (Ef[ids] * ids.ne(pad_id).unsqueeze(-1).float()).sum(dim=dim)

    fn = Ef[field.reshape(-1)]
    names = torch.cat([fn[:5], fn[5:6].expand(128, -1)], 0).unsqueeze(0).expand(B, -1, -1)
    rows = msum(category, 3)                                             # [B,3,8,D]
    rv = category.ne(pad_id).any(-1)                                     # [B,3,8]
    grp = (rows * rv.unsqueeze(-1).float()).sum(2) / (rv.float().sum(2, keepdim=True) + 1e-6)
    gv = rv.any(-1)                                                      # [B,3]
    cat = (grp * gv.unsqueeze(-1).float()).sum(1, keepdim=True) / (gv.float().sum(1, keepdim=True).unsqueeze(-1) + 1e-6)
    vals = torch.cat([F.linear(price.float(), w_price.float()).unsqueeze(1), F.linear(rating.float(), w_rating.float()).unsqueeze(1),
                      msum(brand, 1).unsqueeze(1), msum(name, 1).unsqueeze(1), cat, Ef[description]], 1)
    out.copy_(torch.cat([names, vals], -1).reshape(B * 133, -1))
    ones = torch.ones(B, 1, dtype=torch.bool)
    m = torch.cat([price.sum(1, keepdim=True) != 0, ones, brand[:, :1].ne(pad_id), name[:, :1].ne(pad_id), ones, description.ne(pad_id)], 1)
    mask.copy_(m.to(torch.uint8))


def amazon_table_gather_bwd(dall, price, rating, dw_price, dw_rating, B, D):
    dv = dall.float().view(B, 133, 2 * D)[:, :, D:]
    dw_price.add_(torch.einsum("bk,bd->dk", price.float(), dv[:, 0]))
    dw_rating.add_(torch.einsum("bk,bd->dk", rating.float(), dv[:, 1]))


def rows_gather(src, dst, row_map, live=None):
    n = _n(live, dst.shape[0])              # rows of dst past the live count are left as they are
    ok = row_map[:n] >= 0
    d = dst[:n]
    d.zero_()
    d[ok] = src[row_map[:n][ok]]
    return dst


def beam_topk(logits, V, beam_scores, banned, force_token, ban_token, num_beams, out_scores, out_ids, penalized=None, penalty=1.0,
              penalty_on_logits=False, ncand=0):
    """Contract of mmsum_beam_topk restated with torch ops in the reference's order (adjust_logits, log_softmax, repetition penalty,
    bans, + beam score, topk over [B, beams * V]); ties by lower flat index.  penalty_on_logits: the penalty applies to the raw logits
    (greedy decoding), and the candidates' scores are those penalised logits minus the ORIGINAL row's log-sum-exp (any per-row
    constant: only the order within a row matters there)."""
    x = logits[:, :V].float().clone()
    if force_token >= 0:
        keep = x[:, force_token].clone()
        x.fill_(float("-inf"))
        x[:, force_token] = keep

    def penalise(t):
        if penalized is None or penalty == 1.0 or force_token >= 0:
            return
        for r in range(penalized.shape[0]):
            for tok in penalized[r].tolist():
                if tok < 0:
                    break
                t[r, tok] = t[r, tok] * penalty if t[r, tok] < 0 else t[r, tok] / penalty
    if penalty_on_logits:
        lse = torch.logsumexp(x, dim=-1, keepdim=True)
        penalise(x)
        sc = x - lse
    else:
        sc = torch.log_softmax(x, dim=-1)
        penalise(sc)
    if ban_token >= 0:
        sc[:, ban_token] = float("-inf")
    if banned is not None:
        for r in range(banned.shape[0]):
            for t in banned[r].tolist():
                if t < 0:
                    break                      # the contract: a row's list is filled from the front, its first -1 ends it
                sc[r, t] = float("-inf")
    B = logits.shape[0] // num_beams
    cand = (sc + beam_scores[:, None]).view(B, num_beams * V)
    K = ncand or 2 * num_beams
    order = torch.sort(cand, dim=1, descending=True, stable=True)
    out_scores.copy_(order.values[:, :K])
    out_ids.copy_(order.indices[:, :K])


def decode_self_attn(q, k_cache, v_cache, ancestors, out, H, length, Tmax, scale, k_new=None, v_new=None):
    R, D = q.shape[0], q.shape[1]
    if k_new is not None:            # the kernel appends this step's projections at position length - 1 of every row
        assert bool((ancestors[:, length - 1] == torch.arange(R, dtype=ancestors.dtype)).all())
        k_cache.view(R, Tmax, D)[:, length - 1].copy_(k_new)
        v_cache.view(R, Tmax, D)[:, length - 1].copy_(v_new)
    hd = D // H
    s = torch.arange(length)
    for r in range(R):
        phys = ancestors[r, :length].long() * Tmax + s
        k = k_cache.float()[phys].view(length, H, hd)
        v = v_cache.float()[phys].view(length, H, hd)
        qq = q[r].float().view(H, hd) * scale
        p = torch.softmax(torch.einsum("hd,shd->hs", qq, k), dim=-1)
        out[r] = torch.einsum("hs,shd->hd", p, v).reshape(D).to(out.dtype)


def install(monkeypatch):
    """Route multimodalsum_amd.{engine,modules,optim,generation}.kn to this module for the duration of a test."""
    import sys
    import multimodalsum_amd.engine as eng
    import multimodalsum_amd.modules as mods
    import multimodalsum_amd.optim as opt
    import multimodalsum_amd.generation as gen
    me = sys.modules[__name__]
    for m in (eng, mods, opt, gen):
        monkeypatch.setattr(m, "kn", me)
    # scratch tensors come back poisoned: a schedule that reads a row no kernel wrote (e.g. a compact row past the live
    # count) turns its results into NaN instead of passing on whatever the allocator happened to hand out
    real_empty = eng.Engine.empty

    def poisoned_empty(self, *shape, dtype=None):
        t = real_empty(self, *shape, dtype=dtype)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t

    monkeypatch.setattr(eng.Engine, "empty", poisoned_empty)
