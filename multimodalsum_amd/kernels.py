"""Tensor-level wrappers over the C ABI: raw device pointers + the current HIP stream.

PyTorch is plumbing here (device memory, streams); every function below enqueues hand-written HIP
kernels from libmmsum_hip.so and nothing else.  There is no fallback path: CPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from ._lib import lib, check, AttnDesc, F32, BF16

EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_RELU, EPI_RELU_BWD = 0, 1, 2, 3, 4


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError("unsupported dtype %s" % t.dtype)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmmsum_hip kernels need device tensors (no CPU fallback exists)")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "2-D tensor with unit inner stride expected"
    return t.stride(0)


def gemm_colsum_fusable(a, a_t=False, b_t=False, a2=None):
    """True when mmsum_gemm can add the output's column sums in its epilogue (bf16 LDS-DMA NT path)."""
    return a.dtype == torch.bfloat16 and not a_t and not b_t and a2 is None and a.shape[1] % 64 == 0


def gemm_tn_colsum_ok(dy, x, ws, splitk, colsum):
    """True when the weight-gradient product gemm(dy, x, ws, a_t=True, b_t=True, splitk, slabs=True) can also leave the column sums
    of dy (the bias gradient) in `colsum`: the four-wave 256x256 TN kernel (asks mmsum_gemm_plan, which validates the call)."""
    if dy.dtype != torch.bfloat16 or splitk < 2:
        return False
    try:
        plan = gemm_plan(dy, x, ws, a_t=True, b_t=True, splitk=splitk, slabs=True, colsum=colsum)
    except RuntimeError:
        return False
    return plan[0] == _lib.PLAN_TN_RING and (plan[1], plan[2]) == (256, 256)


def _live(t):
    """Device int32 scalar holding a live row count (or None)."""
    if t is None:
        return None
    assert t.dtype == torch.int32 and t.numel() == 1
    return _p(t)


def _gemm_args(a, b, out_rows, out_cols, a_t, b_t, bias, epi, aux, accumulate, a2, splitk, slabs, colsum, out_f32, colsum_sq=False):
    dt = _dt(b)
    a_f32 = a.dtype == torch.float32 and dt == BF16       # f32 x beside bf16 weights: the decode LM head (MMSUM_GEMM_A_F32)
    assert a_f32 or _dt(a) == dt
    M, K = (a.shape[1], a.shape[0]) if a_t else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if b_t else (b.shape[0], b.shape[1])
    ksplit = 0
    if a2 is not None:
        ksplit = K
        K = K + a2.shape[1]
    assert K == Kb, (K, Kb)
    assert out_rows == (M * splitk if slabs else M) and out_cols == N, (out_rows, out_cols, M, N)
    flags = (_lib.GEMM_A_T if a_t else 0) | (_lib.GEMM_B_T if b_t else 0) | (_lib.GEMM_BIAS if bias is not None else 0)
    flags |= _lib.gemm_epi(epi) | (_lib.GEMM_ACCUM if accumulate else 0) | (_lib.GEMM_SLABS if slabs else 0)
    if colsum is not None:            # NT: f32 [N] += column sums of the stored result; weight-gradient product: f32 [M] += column sums of A
        assert bias is None and colsum.dtype == torch.float32 and ((a_t and b_t) or gemm_colsum_fusable(a, a_t, b_t, a2))
        flags |= _lib.GEMM_COLSUM
        if colsum_sq:                 # colsum holds 2 N floats: [N:] += column sums of the squares (BatchNorm statistics)
            assert not (a_t and b_t) and colsum.numel() == 2 * N
            flags |= _lib.GEMM_COLSUM2
        bias = colsum
    if out_f32:
        flags |= _lib.GEMM_OUT_F32
    if a_f32:
        assert out_f32 and not a_t and a2 is None
        flags |= _lib.GEMM_A_F32
    if bias is not None:
        assert bias.dtype == torch.float32
    return dt, M, N, K, ksplit, flags, bias


def gemm_plan(a, b, out, a_t=False, b_t=False, bias=None, epi=EPI_NONE, aux=None, accumulate=False, a2=None, splitk=1,
              slabs=False, colsum=None, live=None, alpha_dev=None):
    """What gemm(...) with the same arguments would launch: (kernel family, BM, BN, workgroups) -- see mmsum_gemm_plan.
    live / alpha_dev take part in the kernel selection (whether they are given, not their values)."""
    dt, M, N, K, ksplit, flags, bias = _gemm_args(a, b, out.shape[0], out.shape[1], a_t, b_t, bias, epi, aux, accumulate, a2, splitk,
                                                   slabs, colsum, out.dtype == torch.float32)
    plan = (ctypes.c_int * 4)()
    lend = _lends_split_workspace(dt, M, N, a_t, b_t, splitk)
    check(lib.mmsum_gemm_plan(dt, _p(a), _ld(a), _p(a2), _ld(a2) if a2 is not None else 0, ksplit, _p(b), _ld(b), _p(out), _ld(out),
                              _p(bias), _p(aux), _ld(aux) if aux is not None else 0, M, N, K, flags, splitk, _live(live), _p(alpha_dev),
                              GEMM_WORKSPACE_BYTES if lend else 0, plan),
          "mmsum_gemm_plan")
    return tuple(plan)


GEMM_WORKSPACE_BYTES = 4096 + 256 * 128 * 128 * 4          # MMSUM_GEMM_WORKSPACE_BYTES
_split_ws_cache = {}


def _lends_split_workspace(dt, M, N, a_t, b_t, splitk):
    """Products that may take mmsum_gemm's fused split (bf16 NT, a 128x128 tile list of at most half the CUs): only they are lent the
    workspace.  The library decides whether it uses it."""
    return dt == BF16 and not a_t and not b_t and splitk == 1 and ((M + 127) // 128) * ((N + 127) // 128) <= 128


def _split_workspace(device):
    """The zeroed scratch of mmsum_gemm's fused split, one per stream (two streams' products may be in flight at once)."""
    k = (str(device), _stream())
    w = _split_ws_cache.get(k)
    if w is None:
        w = torch.zeros(GEMM_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _split_ws_cache[k] = w
    return w


def gemm(a, b, out, a_t=False, b_t=False, bias=None, epi=EPI_NONE, aux=None, accumulate=False, alpha=1.0, a2=None,
         splitk=1, slabs=False, colsum=None, live=None, alpha_dev=None, colsum_sq=False):
    """out[M,N] = epi(alpha * A.B^T + bias) (+ out).  a: [M,K] (or [K,M] when a_t); b: [N,K] (or [K,N] when b_t);
    a2: optional second half of the K range ([M,K2], natural layout).  out may be f32 while a/b are bf16.
    live: device int32 scalar, the live rows of the row-streamed operand (M, or K of the a_t & b_t product);
    alpha_dev: device f32 scalar multiplied into alpha on the device."""
    if out.dtype != torch.float32:
        assert _dt(out) == _dt(a)
    dt, M, N, K, ksplit, flags, bias = _gemm_args(a, b, out.shape[0], out.shape[1], a_t, b_t, bias, epi, aux, accumulate, a2, splitk,
                                                   slabs, colsum, out.dtype == torch.float32, colsum_sq)
    ws = _split_workspace(out.device) if _lends_split_workspace(dt, M, N, a_t, b_t, splitk) else None
    check(lib.mmsum_gemm(dt, _p(a), _ld(a), _p(a2), _ld(a2) if a2 is not None else 0, ksplit, _p(b), _ld(b), _p(out), _ld(out),
                         _p(bias), _p(aux), _ld(aux) if aux is not None else 0, M, N, K, float(alpha), _p(alpha_dev), flags, splitk, _live(live),
                         _p(ws), GEMM_WORKSPACE_BYTES if ws is not None else 0, _stream()),
          "mmsum_gemm")
    return out


def dec_gemm_workspace(M, N, K, device):
    """A zeroed workspace for dec_gemm products up to this size (tickets + f32 slabs); one serves every product of a stream in turn."""
    return torch.zeros(max(16, lib.mmsum_dec_gemm_workspace(M, N, K)), dtype=torch.uint8, device=device)


def dec_gemm(x, w, out, ws, bias=None, epi=EPI_NONE, x2=None, residual=None):
    """out[M, N] = gelu?(x . w^T + bias) (+ residual): the decode step's weight-streaming product with the reduction split over
    workgroups (mmsum_dec_gemm).  x bf16 [M, K] (or f32: the LM head on the un-rounded final LayerNorm output, out f32); x2: the second
    half of the K range ([M, K2]); w bf16 [N, K]; out bf16 or f32."""
    M, K = x.shape
    N = w.shape[0]
    ksplit = 0
    if x2 is not None:
        ksplit, K = K, K + x2.shape[1]
    assert w.shape[1] == K and out.shape == (M, N) and w.dtype == torch.bfloat16
    flags = _lib.gemm_epi(epi)
    if out.dtype == torch.float32:
        flags |= _lib.GEMM_OUT_F32
    if x.dtype == torch.float32:
        flags |= _lib.GEMM_A_F32
    assert ws.numel() >= lib.mmsum_dec_gemm_workspace(M, N, K), "dec_gemm workspace too small"
    check(lib.mmsum_dec_gemm(_p(x), _ld(x), _p(x2), _ld(x2) if x2 is not None else 0, ksplit, _p(w), _ld(w), _p(bias), _p(residual),
                             _ld(residual) if residual is not None else 0, _p(out), _ld(out), M, N, K, flags, _p(ws), _stream()), "mmsum_dec_gemm")
    return out


def slab_reduce(ws, nslabs, out, accumulate=True):
    """out[rows, cols] (+)= sum of the nslabs f32 slabs stacked in ws [nslabs*rows, cols]."""
    rows, cols = out.shape
    check(lib.mmsum_slab_reduce(_p(ws), nslabs, rows, cols, _p(out), _ld(out), int(accumulate), _stream()), "mmsum_slab_reduce")


_ws_cache = {}


def _workspace(nbytes, device, key):
    k = (key, str(device), _stream())        # per stream: the image branch of the step runs beside the text encoder
    w = _ws_cache.get(k)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws_cache[k] = w
    return w


def colsum(x, out, accumulate=False, live=None):
    R, C = x.shape
    ws = _workspace(lib.mmsum_colsum_workspace(C), x.device, "colsum")
    check(lib.mmsum_colsum(_dt(x), _p(x), _ld(x), R, C, _p(out), int(accumulate), _p(ws), _live(live), _stream()), "mmsum_colsum")
    return out


def embed_ln_fwd(ids, E, P, rating_diff, rvec, gamma, beta, y, mean, rstd, nseq, T, pos_offset, eps, p_drop, seed, salt=None):
    D = E.shape[1]
    check(lib.mmsum_embed_ln_fwd(_dt(E), _p(ids), _p(E), _p(P), _p(rating_diff), _p(rvec), _p(gamma), _p(beta), _p(y),
                                 _p(mean), _p(rstd), nseq, T, D, pos_offset, eps, p_drop, seed, _p(salt), _stream()), "mmsum_embed_ln_fwd")


def embed_ln_bwd(dy, ids, E, P, rating_diff, rvec, gamma, mean, rstd, dE, dP, drvec, dgamma, dbeta, nseq, T, pos_offset,
                 pad_id, p_drop, seed, salt=None):
    D = E.shape[1]
    check(lib.mmsum_embed_ln_bwd(_dt(E), _p(dy), _p(ids), _p(E), _p(P), _p(rating_diff), _p(rvec), _p(gamma), _p(mean),
                                 _p(rstd), _p(dE), _p(dP), _p(drvec), _p(dgamma), _p(dbeta), nseq, T, D, pos_offset, pad_id,
                                 p_drop, seed, _p(salt), _stream()), "mmsum_embed_ln_bwd")


def add_ln_fwd(x, res, gamma, beta, y, mean, rstd, eps, p_drop, seed, salt=None, live=None, y_f32=None):
    """y_f32 (f32 [R, D], optional): the same result un-rounded, beside y (the decode step's last LayerNorm in bf16 mode)."""
    R, D = x.shape
    if y_f32 is not None:
        assert y_f32.dtype == torch.float32 and y_f32.shape == (R, D) and y_f32.is_contiguous()
    check(lib.mmsum_add_ln_fwd(_dt(x), _p(x), _p(res), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), R, D, eps, p_drop, seed,
                               _p(salt), _live(live), _p(y_f32), _stream()), "mmsum_add_ln_fwd")


def add_ln_bwd(dy, x, res, gamma, mean, rstd, dx, dres, accumulate_dres, dgamma, dbeta, p_drop, seed, dxsum=None, salt=None, live=None):
    """dxsum (f32 [D], optional) += column sums of dx: the bias gradient of the Linear that produced x."""
    R, D = x.shape
    check(lib.mmsum_add_ln_bwd(_dt(x), _p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dres),
                               int(accumulate_dres), _p(dgamma), _p(dbeta), R, D, p_drop, seed, _p(salt), _p(dxsum), _live(live),
                               _stream()), "mmsum_add_ln_bwd")


def make_attn_desc(q, k, v, out, pad, null_entity, n_qblocks, T, qpb, N, S, H, exclude_self, causal, scale, q_rows=None, kv_rows=None,
                   causal_q0=0):
    """q_rows / kv_rows: optional int32 row maps (logical padded row -> physical row of the compact matrices, -1 = absent).
    causal_q0 (causal only, a multiple of 32): key position of the query block's first row."""
    d = AttnDesc()
    d.causal_q0 = int(causal_q0)
    d._keep = (q_rows, kv_rows)              # the maps must outlive the descriptor's launches
    d.q_rows, d.kv_rows = _p(q_rows), _p(kv_rows)
    d.q, d.k, d.v, d.out = _p(q), _p(k), _p(v), _p(out)
    d.ldq, d.ldk, d.ldv = q.stride(0), k.stride(0), v.stride(0)
    d.ldo = out.stride(0) if out is not None else 0
    d.pad, d.null_entity = _p(pad), _p(null_entity)
    d.n_qblocks, d.T, d.qpb, d.N, d.S, d.H = n_qblocks, T, qpb, N, S, H
    d.exclude_self, d.causal, d.scale = int(exclude_self), int(causal), float(scale)
    return d


def entity_null(pad, null_entity, n_entities, S):
    check(lib.mmsum_entity_null(_p(pad), _p(null_entity), n_entities, S, _stream()), "mmsum_entity_null")


def attn_fwd(desc, dtype_tensor):
    check(lib.mmsum_attn_fwd(_dt(dtype_tensor), ctypes.byref(desc), _stream()), "mmsum_attn_fwd")


def attn_bwd_workspace(desc):
    return lib.mmsum_attn_bwd_workspace(ctypes.byref(desc))


def attn_bwd(desc, dout, dq, accumulate_dq, dk, dv, stats):
    check(lib.mmsum_attn_bwd(_dt(dout), ctypes.byref(desc), _p(dout), dout.stride(0), _p(dq), dq.stride(0), int(accumulate_dq),
                             _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(stats), _stream()), "mmsum_attn_bwd")


def gate_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, out, rows_per_b):
    R, D = yt.shape
    check(lib.mmsum_gate_fwd(_dt(yt), _p(pa), _p(pb), _p(yt), _p(ytab), _p(yimg), _p(no_table), _p(no_img), _p(out), R, D,
                             rows_per_b, _stream()), "mmsum_gate_fwd")


def gate_add_ln_fwd(pa, pb, yt, ytab, yimg, no_table, no_img, res, gamma, beta, y, rows_per_b, eps):
    """y = LN(res + yt + relu(tanh(pa)) ytab + relu(tanh(pb)) yimg) in one launch (the decode step: nothing saved for a backward)."""
    R, D = yt.shape
    check(lib.mmsum_gate_add_ln_fwd(_dt(yt), _p(pa), _p(pb), _p(yt), _p(ytab), _p(yimg), _p(no_table), _p(no_img), _p(res), _p(gamma), _p(beta),
                                    _p(y), R, D, rows_per_b, eps, _stream()), "mmsum_gate_add_ln_fwd")


def gate_bwd(dout, pa, pb, ytab, yimg, no_table, no_img, dpa, dpb, dyt, dytab, dyimg, rows_per_b, sums=None):
    """sums = (sum_dpa, sum_dpb), f32 [D] each: += column sums of dpa and dpb (the alpha_proj / beta_proj bias gradients)."""
    R, D = dout.shape
    s0, s1 = sums if sums is not None else (None, None)
    check(lib.mmsum_gate_bwd(_dt(dout), _p(dout), _p(pa), _p(pb), _p(ytab), _p(yimg), _p(no_table), _p(no_img), _p(dpa), _p(dpb),
                             _p(dyt), _p(dytab), _p(dyimg), R, D, rows_per_b, _p(s0), _p(s1), _stream()), "mmsum_gate_bwd")


def ls_loss(logits, target, row_loss, V, smoothing, gscale, write_grad=True):
    R = logits.shape[0]
    check(lib.mmsum_ls_loss(_dt(logits), _p(logits), logits.stride(0), _p(target), _p(row_loss), R, V, smoothing, gscale,
                            int(write_grad), _stream()), "mmsum_ls_loss")


def segment_sum(x, out, nseg, seg, scale):
    check(lib.mmsum_segment_sum(_p(x), _p(out), nseg, seg, scale, _stream()), "mmsum_segment_sum")


def l2norm_sq(g, out, accumulate=False):
    ws = _workspace(lib.mmsum_l2_workspace(), g.device, "l2")
    check(lib.mmsum_l2norm_sq(_p(g), g.numel(), _p(out), int(accumulate), _p(ws), _stream()), "mmsum_l2norm_sq")


def adamw(p, g, m, v, shadow, hyper, norm_sq, beta1, beta2, eps):
    check(lib.mmsum_adamw(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(hyper), _p(norm_sq), beta1, beta2, eps, _stream()),
          "mmsum_adamw")


def cast(dst, src):
    assert dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    check(lib.mmsum_cast(_dt(dst), _p(dst), _dt(src), _p(src), src.numel(), _stream()), "mmsum_cast")
    return dst


def transpose(src, dst, rows_pad=None, colsum=None):
    """dst[c, r] = src[r, c] (bf16); dst has >= rows_pad columns, [rows, rows_pad) zero filled;
    colsum (f32 [cols], optional) += column sums of src."""
    rows, cols = src.shape
    rp = rows if rows_pad is None else rows_pad
    check(lib.mmsum_transpose_bf16(_p(src), _ld(src), _p(dst), _ld(dst), rows, cols, rp, _p(colsum), _stream()), "mmsum_transpose_bf16")
    return dst


def transpose_batched(src_base, dst_base, desc, n, max_tiles):
    check(lib.mmsum_transpose_bf16_batched(_p(src_base), _p(dst_base), _p(desc), n, max_tiles, _stream()), "mmsum_transpose_bf16_batched")


def scale_by_clip(g, norm_sq, max_norm):
    check(lib.mmsum_scale_by_clip(_p(g), g.numel(), _p(norm_sq), max_norm, _stream()), "mmsum_scale_by_clip")


class ImagePlan:
    """Which slots of a batch of `n` images the image branch runs (mmsum_image_plan): the non-empty slots first, then ONE representative
    of the empty (masked, all-zero) ones with a multiplicity.  Everything is device-resident (the counts differ from batch to batch inside
    one captured graph): plan int32 [4 + len(rows)] = {images that run, representative's index or -1, multiplicity, non-empty slots,
    row counts ...}; rows(rpi, adjust) = the device row count `images that run * rpi + adjust` for the GEMM entry points' live_rows."""

    def __init__(self, n, positions, row_kinds, device):
        assert len(row_kinds) <= 8
        self.n, self.positions, self.row_kinds = n, positions, list(row_kinds)
        self.plan = torch.empty(4 + len(self.row_kinds), dtype=torch.int32, device=device)       # all four are written whole by the plan kernel
        self.src = torch.empty(n, dtype=torch.int32, device=device)
        self.slot_rows = torch.empty(n * positions, dtype=torch.int64, device=device)
        self.run_rows = torch.empty(n * positions, dtype=torch.int64, device=device)

    def rows(self, rpi, adjust=0):
        k = self.row_kinds.index((rpi, adjust))
        return self.plan[4 + k:5 + k]


def image_plan(img, mask, ip):
    """Fills ip (ImagePlan) for the images img f32 [n, ...] with mask [n] (non-zero = a real image)."""
    n = ip.n
    assert img.dtype == torch.float32 and img.is_contiguous() and img.shape[0] == n and mask.numel() == n
    m8 = mask.reshape(-1).to(torch.uint8).contiguous()
    nk = len(ip.row_kinds)
    rpi = (ctypes.c_int * max(nk, 1))(*[r for r, _ in ip.row_kinds])
    adj = (ctypes.c_int * max(nk, 1))(*[a for _, a in ip.row_kinds])
    ws = _workspace(lib.mmsum_image_plan_workspace(n), img.device, "imgplan")
    check(lib.mmsum_image_plan(_p(img), img.numel() // n, _p(m8), n, ip.positions, rpi, adj, nk, _p(ip.plan), _p(ip.src), _p(ip.slot_rows),
                               _p(ip.run_rows), _p(ws), _stream()), "mmsum_image_plan")
    return ip


def _img(images, R=None):
    """(device pointer of the live-image window, rows per image) for an ImagePlan or None."""
    if images is None:
        return None, 0
    if R is None:
        return _p(images.plan), 0
    assert R % images.n == 0
    return _p(images.plan), R // images.n


def im2col(x, col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images=None):
    check(lib.mmsum_im2col(_dt(x), _p(x), _p(col), N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, _img(images)[0], _stream()), "mmsum_im2col")


def col2im(dcol, dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images=None):
    check(lib.mmsum_col2im(_dt(dcol), _p(dcol), _p(dx), N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, _img(images)[0], _stream()), "mmsum_col2im")


def conv_weight_to_matrix(matrix, weight, Cout, Cin, KH, KW, Kpad):
    check(lib.mmsum_conv_weight_permute(_dt(matrix), _p(matrix), _p(weight), Cout, Cin, KH, KW, Kpad, 1, 0, _stream()),
          "mmsum_conv_weight_permute")


def conv_matrix_grad_to_weight(matrix_f32, dweight, Cout, Cin, KH, KW, Kpad, accumulate):
    check(lib.mmsum_conv_weight_permute(F32, _p(matrix_f32), _p(dweight), Cout, Cin, KH, KW, Kpad, 0, int(accumulate), _stream()),
          "mmsum_conv_weight_permute")


def bn_reduce(x, sums, images=None):
    R, C = x.shape
    ws = _workspace(lib.mmsum_bn_workspace(C), x.device, "bn")
    ip, rpi = _img(images, R)
    check(lib.mmsum_bn_reduce(_dt(x), _p(x), R, C, _p(sums), _p(ws), ip, rpi, _stream()), "mmsum_bn_reduce")


def bn_rep_fix(y, raw, images):
    """raw (the GEMM epilogue's plain column sums over the rows that ran) += (multiplicity - 1) * the representative's rows' share."""
    R, C = y.shape
    ip, rpi = _img(images, R)
    check(lib.mmsum_bn_rep_fix(_dt(y), _p(y), _p(raw), R, C, ip, rpi, _stream()), "mmsum_bn_rep_fix")


def bn_stats_from_sums(raw, R, sums, running_mean, running_var, momentum):
    """raw f32 [2C] = {sum x, sum x^2} (the conv GEMM's epilogue: gemm(..., colsum=raw, colsum_sq=True)) -> sums = {mean, var}, and the
    running statistics' momentum update."""
    C = sums.numel() // 2
    check(lib.mmsum_bn_stats_from_sums(_p(raw), R, C, _p(sums), _p(running_mean), _p(running_var), momentum, _stream()), "mmsum_bn_stats_from_sums")


def bn_apply(x, sums, gamma, beta, residual, y, running_mean, running_var, eps, momentum, relu, training, pad_hw=None, raw=None, images=None):
    """pad_hw = (H, W): y is the zero-bordered padded layout [n, H+2, W+2, C] (the operand of conv3x3_gemm); the caller zeroed it.
    raw (f32 [2C], training): plain column sums {sum x, sum x^2} from the convolution's GEMM epilogue -- the kernel derives the statistics from
    them, writes {mean, var} to `sums` and updates the running statistics itself (no bn_stats_from_sums launch)."""
    R, C = x.shape
    pH, pW = pad_hw or (0, 0)
    ip, rpi = _img(images, R)
    check(lib.mmsum_bn_apply(_dt(x), _p(x), _p(sums), _p(raw), _p(gamma), _p(beta), _p(residual), _p(y), _p(running_mean), _p(running_var),
                             R, C, eps, momentum, int(relu), int(training), pH, pW, ip, rpi, _stream()), "mmsum_bn_apply")


def bn_bwd_reduce(dy, y, x, sums, dsums, eps, relu, pad_hw=None, images=None):
    R, C = x.shape
    pH, pW = pad_hw or (0, 0)
    ws = _workspace(lib.mmsum_bn_workspace(C), x.device, "bn")
    ip, rpi = _img(images, R)
    check(lib.mmsum_bn_bwd_reduce(_dt(x), _p(dy), _p(y), _p(x), _p(sums), R, C, eps, int(relu), _p(dsums), _p(ws), pH, pW, ip, rpi, _stream()),
          "mmsum_bn_bwd_reduce")


def bn_bwd_apply(dy, y, x, sums, dsums, gamma, dx, dresidual, dgamma, dbeta, eps, relu, pad_hw=None, dx_pad_hw=None, images=None):
    """pad_hw: y (the ReLU mask) is in the padded layout; dx_pad_hw = (H, W): dx is WRITTEN in the padded layout (interior only)."""
    R, C = x.shape
    pH, pW = pad_hw or (0, 0)
    dH, dW = dx_pad_hw or (0, 0)
    if dx_pad_hw is not None:
        assert dx.shape == (R // (dH * dW) * (dH + 2) * (dW + 2), C) and dx.is_contiguous()
    ip, rpi = _img(images, R)
    check(lib.mmsum_bn_bwd_apply(_dt(x), _p(dy), _p(y), _p(x), _p(sums), _p(dsums), _p(gamma), _p(dx), _p(dresidual), _p(dgamma),
                                 _p(dbeta), R, C, eps, int(relu), pH, pW, dH, dW, ip, rpi, _stream()), "mmsum_bn_bwd_apply")


def conv3x3_gemm(xp, w, y, n, H, W, C, stats=None, live=None):
    """y [n*H*W, Cout] = 3x3 convolution (stride 1, padding 1) of xp, the zero-bordered padded NHWC activations [n*(H+2)*(W+2), C] (bf16),
    with the weight matrix w [Cout, >= 9 C] in (ky, kx, c) column order -- an implicit GEMM: no im2col matrix (mmsum_conv3x3_gemm).
    stats (f32 [2 Cout], optional) += column sums of y and y^2 (the BatchNorm statistics)."""
    assert xp.dtype == torch.bfloat16 and xp.shape == (n * (H + 2) * (W + 2), C) and xp.is_contiguous() and y.shape[0] == n * H * W
    check(lib.mmsum_conv3x3_gemm(_p(xp), _p(w), _ld(w), _p(y), _ld(y), _p(stats), n, H, W, C, w.shape[0], _live(live), _stream()), "mmsum_conv3x3_gemm")
    return y


def conv3x3_wgrad(dyp, xp, out, n, H, W, C, splitk=1, live=None):
    """out f32 [splitk * Cout, 9 C] (splitk slabs; slab_reduce adds them) = weight gradient of the 3x3 / stride 1 / padding 1 convolution in the
    (ky, kx, c) matrix layout, from the PADDED output gradient dyp [n*(H+2)*(W+2), Cout] and the PADDED input xp [n*(H+2)*(W+2), C], both
    with zero borders (mmsum_conv3x3_wgrad: the reduction-major kernel with a per-tile row shift; no im2col matrix)."""
    Cout = dyp.shape[1]
    rows = n * (H + 2) * (W + 2)
    assert dyp.dtype == xp.dtype == torch.bfloat16 and out.dtype == torch.float32 and dyp.is_contiguous() and xp.is_contiguous()
    assert dyp.shape == (rows, Cout) and xp.shape == (rows, C) and out.shape == (splitk * Cout, 9 * C) and out.is_contiguous()
    check(lib.mmsum_conv3x3_wgrad(_p(dyp), _p(xp), _p(out), _ld(out), n, H, W, C, Cout, splitk, _live(live), _stream()), "mmsum_conv3x3_wgrad")
    return out


def conv_weight_to_dgrad_matrix(matrix, weight, Cout, Cin, KH, KW, Kpad):
    """matrix [Cin, Kpad >= KH*KW*Cout] = the weights rotated by 180 degrees with the channel roles exchanged (mmsum_conv_weight_permute, mode 2):
    the input gradient of a stride-1 convolution = conv3x3_gemm(padded dy, matrix)."""
    check(lib.mmsum_conv_weight_permute(_dt(matrix), _p(matrix), _p(weight), Cout, Cin, KH, KW, Kpad, 2, 0, _stream()),
          "mmsum_conv_weight_permute")


def maxpool3x3s2(x, y, N, H, W, C, Ho, Wo, images=None):
    check(lib.mmsum_maxpool3x3s2(_dt(x), _p(x), _p(y), N, H, W, C, Ho, Wo, _img(images)[0], _stream()), "mmsum_maxpool3x3s2")


def nchw_to_nhwc(x, y, N, C, H, W, images=None):
    """images (ImagePlan): image r of y is slot images.src[r] of x, and only the images that run are converted."""
    check(lib.mmsum_nchw_to_nhwc(_dt(y), _p(x), _p(y), N, C, H, W, _img(images)[0], _p(images.src) if images is not None else None, _stream()),
          "mmsum_nchw_to_nhwc")


def table_gather(E, field, fv, w_rating, w_hours, out, mask, B, pad_id):
    name, category, str_cat, str_bool, rating, hours = fv
    D = E.shape[1]
    check(lib.mmsum_table_gather(_dt(E), _p(E), _p(field), _p(name), _p(category), _p(str_cat), _p(str_bool), _p(rating), _p(hours),
                                 _p(w_rating), _p(w_hours), _p(out), _p(mask), B, D, pad_id, _stream()), "mmsum_table_gather")


def table_gather_bwd(dall, rating, hours, dw_rating, dw_hours, B, D):
    check(lib.mmsum_table_gather_bwd(_dt(dall), _p(dall), _p(rating), _p(hours), _p(dw_rating), _p(dw_hours), B, D, _stream()),
          "mmsum_table_gather_bwd")


def amazon_table_gather(E, field, fv, w_price, w_rating, out, mask, B, pad_id):
    price, rating, brand, name, category, description = fv
    D = E.shape[1]
    check(lib.mmsum_amazon_table_gather(_dt(E), _p(E), _p(field), _p(price), _p(rating), _p(brand), _p(name), _p(category), _p(description),
                                        _p(w_price), _p(w_rating), _p(out), _p(mask), B, D, pad_id, _stream()), "mmsum_amazon_table_gather")


def amazon_table_gather_bwd(dall, price, rating, dw_price, dw_rating, B, D):
    check(lib.mmsum_amazon_table_gather_bwd(_dt(dall), _p(dall), _p(price), _p(rating), _p(dw_price), _p(dw_rating), B, D, _stream()),
          "mmsum_amazon_table_gather_bwd")


def rows_gather(src, dst, row_map, live=None):
    """dst[i] = src[row_map[i]] (zeros where row_map[i] < 0).  src [Rs, C], dst [Rd, C] (unit inner stride), row_map int64 [Rd];
    live: device int32 scalar, only rows i < live of dst are written."""
    assert src.dtype == dst.dtype and src.shape[1] == dst.shape[1] and row_map.dtype == torch.int64 and row_map.numel() == dst.shape[0]
    es = src.element_size()
    check(lib.mmsum_rows_gather(_p(src), _ld(src) * es, src.shape[0], _p(dst), _ld(dst) * es, _p(row_map), dst.shape[0], src.shape[1] * es,
                                _live(live), _stream()), "mmsum_rows_gather")
    return dst


def beam_topk(logits, V, beam_scores, banned, force_token, ban_token, num_beams, out_scores, out_ids, penalized=None, penalty=1.0,
              penalty_on_logits=False, ncand=0):
    """Tail of one beam-search step (see mmsum_beam_topk): logits [rows, >=V] (banned entries are overwritten with -inf),
    beam_scores [rows] f32, banned [rows, nban] int32 (filled from the front, the first -1 ends a row's list) or None -> out_scores f32 / out_ids int64 [B, 2*num_beams].
    penalized [rows, npen] int32 (a row's distinct previous tokens, -1 terminated) + penalty: the repetition penalty, on the
    log-probabilities (beam search) or on the raw logits (penalty_on_logits: greedy decoding).
    ncand > 0: that many candidates per business instead of 2 * num_beams (sampling: the top_k best of a row; out_* [B, ncand])."""
    rows = logits.shape[0]
    assert beam_scores.dtype == torch.float32 and out_scores.dtype == torch.float32 and out_ids.dtype == torch.int64
    nban = 0 if banned is None else banned.shape[1]
    if banned is not None:
        assert banned.dtype == torch.int32 and banned.is_contiguous() and banned.shape[0] == rows
    npen = 0 if penalized is None else penalized.shape[1]
    if penalized is not None:
        assert penalized.dtype == torch.int32 and penalized.is_contiguous() and penalized.shape[0] == rows
    assert out_scores.shape[1] == (ncand or 2 * num_beams) and out_ids.shape == out_scores.shape
    ws = _workspace(lib.mmsum_beam_topk_workspace(rows, num_beams, int(ncand)), logits.device, "topk")
    check(lib.mmsum_beam_topk(_dt(logits), _p(logits), _ld(logits), V, _p(beam_scores), _p(banned), nban, int(force_token), int(ban_token), rows,
                              num_beams, _p(ws), _p(out_scores), _p(out_ids), _p(penalized), npen, float(penalty), int(bool(penalty_on_logits)),
                              int(ncand), _stream()), "mmsum_beam_topk")


def decode_self_attn(q, k_cache, v_cache, ancestors, out, H, length, Tmax, scale, k_new=None, v_new=None):
    """out[r] = softmax(q[r] . K_r^T) V_r per head, K_r / V_r = the first `length` cache positions of hypothesis r reached through
    cache row ancestors[r, s] * Tmax + s (see mmsum_decode_self_attn).  k_new / v_new [rows, D]: this step's projections; the kernel
    appends them to the caches at position length - 1 (ancestors[r, length - 1] must be r)."""
    assert ancestors.dtype == torch.int32 and ancestors.is_contiguous() and ancestors.shape == (q.shape[0], Tmax)
    assert k_cache.shape == v_cache.shape and (k_new is None) == (v_new is None)
    if k_new is not None:
        assert k_new.stride(0) == v_new.stride(0) and k_new.stride(1) == 1 and v_new.stride(1) == 1
    check(lib.mmsum_decode_self_attn(_dt(q), _p(q), q.stride(0), _p(k_cache), _p(v_cache), _ld(k_cache), _p(ancestors), _p(out), out.stride(0),
                                     q.shape[0], H, int(length), Tmax, float(scale), _p(k_new), _p(v_new),
                                     k_new.stride(0) if k_new is not None else 0, _stream()), "mmsum_decode_self_attn")


def decode_cross_attn_workspace(n_entities, H, qpb, B, nmod, device):
    return torch.zeros(max(16, lib.mmsum_decode_cross_attn_workspace(n_entities, H, qpb, B, nmod)), dtype=torch.uint8, device=device)


def decode_cross_attn(q, mods, out, ws, B, qpb, H, scale):
    """The decode step's cross-attention + entity mean over the cached K / V of every modality in one launch (mmsum_decode_cross_attn).
    q bf16 (or f32: the parity mode) [B*qpb, H*64]; mods: list of (k, v, pad uint8 [B,N,S] or None, null_entity uint8 [B*N] or None, N, S)
    with k / v [B*N*S, H*64] views of one pitch in q's dtype; out [len(mods) * B*qpb, H*64] in q's dtype."""
    arr = (_lib.XattnMemory * len(mods))()
    ldkv = mods[0][0].stride(0)
    for i, (k, v, pad, nul, N, S) in enumerate(mods):
        assert k.dtype == q.dtype and v.dtype == q.dtype and k.stride(0) == ldkv and v.stride(0) == ldkv and k.shape[0] == B * N * S
        arr[i].k, arr[i].v, arr[i].pad, arr[i].null_entity, arr[i].N, arr[i].S = _p(k), _p(v), _p(pad), _p(nul), N, S
    assert out.dtype == q.dtype and out.shape[0] == len(mods) * B * qpb
    check(lib.mmsum_decode_cross_attn(_dt(q), _p(q), q.stride(0), arr, len(mods), ldkv, _p(out), out.stride(0), B, qpb, H, float(scale), _p(ws), _stream()),
          "mmsum_decode_cross_attn")
    return out


def gemm_pair(xs, x2s, ws, outs, biases):
    """Two independent bf16 products of one shape in one launch (mmsum_gemm_pair): outs[i] = [xs[i] | x2s[i]] . ws[i]^T + biases[i]."""
    arr = (_lib.GemmOperands * 2)()
    M, k1 = xs[0].shape
    k2 = x2s[0].shape[1] if x2s[0] is not None else 0
    N, K = ws[0].shape
    assert K == k1 + k2
    for i in range(2):
        assert xs[i].shape == (M, k1) and ws[i].shape == (N, K) and outs[i].shape == (M, N) and xs[i].dtype == torch.bfloat16
        o = arr[i]
        o.A, o.A2, o.B, o.C, o.bias = _p(xs[i]), _p(x2s[i]), _p(ws[i]), _p(outs[i]), _p(biases[i])
        o.lda, o.lda2, o.ldb, o.ldc = _ld(xs[i]), (_ld(x2s[i]) if x2s[i] is not None else 0), _ld(ws[i]), _ld(outs[i])
    check(lib.mmsum_gemm_pair(arr, M, N, K, k1 if k2 else 0, _stream()), "mmsum_gemm_pair")
