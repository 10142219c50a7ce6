"""GPU: the products of the SMALL-batch step (the reference's own per-GPU batch is 1, multimodal_train.py:420: 1,152 decoder rows, a few
hundred encoder rows) on the kernels mmsum_gemm sends them to:

  * a 128x128 tile list of one round of the CUs or less runs on gemm_nt_ring_kernel<128, 128, 4, 2> (eight waves per tile);
  * of those, the products with at most half a round of tiles and K >= 2,304 run the FUSED SPLIT (gemm_nt_ring_kernel<..., FS>):
    2 .. 4 reduction slices per tile on as many CUs, met inside the launch through the workspace the caller lends (write-through
    slabs, arrival ticket, the last arriver adds the slices in slice order and runs the epilogue);
  * the weight gradients of few rows run unsplit (engine.splitk_rule).

Every epilogue / output form the step uses, element-wise against an fp32 matmul of the same bf16 operands; the plan asserts that each
case reaches the kernel it means to cover; the split must not depend on the order the slices arrive in (bit-equal repeats), must
leave the workspace reusable (products back to back, no host synchronisation) and must hold for two streams at once.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multimodalsum_amd import kernels as kn
    from multimodalsum_amd import _lib

from multimodalsum_amd.engine import splitk_rule
from tests.test_bench_shapes_gpu import rnd, check, DEV, BF


def tiles128(M, N):
    return ((M + 127) // 128) * ((N + 127) // 128)


# (M, N, K, reduction slices the library is expected to take)
CASES = [(1152, 1024, 1024, 1),        # decoder rows at B = 1: 72 tiles, eight waves, K too short to split
         (1152, 1024, 4096, 3),        # fc2 / fc1's input gradient at B = 1: 72 tiles x 3 slices
         (640, 1024, 4096, 4),         # encoder rows at B = 1: 40 tiles x 4
         (640, 3072, 1024, 1),         # qkv: 120 tiles
         (600, 1000, 2304, 3),         # ragged rows and columns: 40 tiles, 72 slabs -> 3 slices of 24 (4 would hold 18 < 24 each)
         (1000, 520, 128, 1),          # short reduction (table encoder width)
         (200, 136, 4608, 4)]          # four tiles, ragged, a long reduction


def expected_slices(M, N, K):
    t = tiles128(M, N)
    s = min(4, 256 // t)
    ns = K // 32
    while s >= 2:
        per = ((ns + s - 1) // s + 1) & ~1
        if per >= 24 and (s - 1) * per < ns:
            return s
        s -= 1
    return 1


@pytest.mark.parametrize("M,N,K,slices", CASES)
def test_nt_small_tile_lists_all_epilogues(M, N, K, slices):
    assert expected_slices(M, N, K) == slices            # the host statement of nt_fused_split (gemm_fast.hip)
    a, w = rnd(M, K, seed=1, std=0.5), rnd(N, K, seed=2, std=0.5)
    bias = rnd(N, dtype=torch.float32, seed=3)
    ref = a.float() @ w.float().t()
    sc = math.sqrt(K / 1024.0)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    plan = kn.gemm_plan(a, w, out, bias=bias)
    assert plan[:3] == (_lib.PLAN_NT_RING, 128, 128) and plan[3] == tiles128(M, N) * slices, plan
    kn.gemm(a, w, out, bias=bias)
    check(out, ref + bias, "bias", atol=2e-2 * sc)
    again = torch.empty_like(out)
    for _ in range(3):                                     # the sum does not depend on which slice arrives last
        kn.gemm(a, w, again, bias=bias)
        assert torch.equal(again, out)
    aux = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    kn.gemm(a, w, out, bias=bias, alpha=0.25, epi=kn.EPI_GELU, aux=aux)
    pre = ref * 0.25 + bias
    check(aux, pre, "gelu aux", atol=2e-2 * sc)
    check(out, F.gelu(pre), "gelu out", atol=2e-2 * sc)
    u = rnd(M, N, seed=4)
    uf = u.float()
    gp = 0.5 * (1 + torch.erf(uf / math.sqrt(2))) + uf * torch.exp(-0.5 * uf * uf) / math.sqrt(2 * math.pi)
    if N % 8 == 0:
        if kn.gemm_colsum_fusable(a):
            cs = torch.ones(N, device=DEV)
            kn.gemm(a, w, out, epi=kn.EPI_GELU_BWD, aux=u, colsum=cs)          # + the bias gradient's column sums, once per column
            want_cs = 1.0 + out.double().sum(0)
            assert ((cs.double() - want_cs).abs() <= 1e-3 * out.double().abs().sum(0) + 1e-2).all(), "column sums"
        else:
            kn.gemm(a, w, out, epi=kn.EPI_GELU_BWD, aux=u)
        check(out, ref * gp, "gelu' out", rel=2.0 ** -6, atol=4e-2 * sc)
        r = rnd(M, N, seed=5)
        kn.gemm(a, w, out, epi=kn.EPI_RELU_BWD, aux=r)
        check(out, ref * (r.float() > 0), "relu'", atol=2e-2 * sc)
    kn.gemm(a, w, out, bias=bias, epi=kn.EPI_RELU)
    check(out, torch.relu(ref + bias), "relu", atol=2e-2 * sc)
    prev = rnd(M, N, seed=6)
    acc = prev.clone()
    kn.gemm(a, w, acc, accumulate=True)
    check(acc, prev.float() + ref, "+= bf16", rel=2.0 ** -6, atol=3e-2 * sc)
    accf = rnd(M, N, dtype=torch.float32, seed=7)
    want = accf + ref
    kn.gemm(a, w, accf, accumulate=True)
    check(accf, want, "+= f32", rel=1e-4, atol=2e-3 * sc)
    kn.gemm(a, w, accf)
    check(accf, ref, "f32 out", rel=1e-4, atol=2e-3 * sc)


def test_nt_fused_split_two_operands_and_live_rows():
    """K split over two A tensors (the alpha / beta projections' cat([text, table]) without the concat) through the fused split, and a
    device-side live row count below the capacity: rows past it are neither read nor written, the slices of the live tiles still
    meet (the ticket of a tile is taken by exactly its slices)."""
    M, N, K1, K2 = 1152, 1024, 2048, 2048
    a1, a2, w = rnd(M, K1, seed=1, std=0.5), rnd(M, K2, seed=2, std=0.5), rnd(N, K1 + K2, seed=3, std=0.5)
    bias = rnd(N, dtype=torch.float32, seed=4)
    out = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
    assert kn.gemm_plan(a1, w, out, a2=a2, bias=bias)[3] == tiles128(M, N) * 3
    kn.gemm(a1, w, out, a2=a2, bias=bias)
    ref = torch.cat([a1, a2], 1).float() @ w.float().t() + bias
    check(out, ref, "two operands", atol=4e-2)
    a = torch.cat([a1, a2], 1).contiguous()
    for live_rows in (1152, 1000, 129, 1, 0):
        live = torch.tensor([live_rows], device=DEV, dtype=torch.int32)
        out.fill_(7.0)
        kn.gemm(a, w, out, bias=bias, live=live)
        check(out[:live_rows], ref[:live_rows], "live %d" % live_rows, atol=4e-2)
        assert bool((out[live_rows:] == 7.0).all()), "rows past the live count were written"
    # the workspace is left reusable: a full product right after the partial ones
    kn.gemm(a, w, out, bias=bias)
    check(out, ref, "after live-row products", atol=4e-2)


def test_nt_fused_split_back_to_back_and_two_streams():
    """Products of different shapes queue on one stream without host synchronisation (one workspace serves them in turn: the kernel
    leaves the ticket words zero); two streams hold their own workspaces and run at once."""
    shapes = [(1152, 1024, 4096), (640, 1024, 4096), (1152, 1024, 2304), (200, 136, 4608), (1152, 1024, 4096)]
    ops = [(rnd(M, K, seed=10 + i, std=0.5), rnd(N, K, seed=20 + i, std=0.5)) for i, (M, N, K) in enumerate(shapes)]
    refs = [a.float() @ w.float().t() for a, w in ops]
    outs = [torch.empty(a.shape[0], w.shape[0], device=DEV, dtype=BF) for a, w in ops]
    for rep in range(4):
        for (a, w), o in zip(ops, outs):
            kn.gemm(a, w, o)
    torch.cuda.synchronize()
    for (M, N, K), o, r in zip(shapes, outs, refs):
        check(o, r, "back to back %s" % ((M, N, K),), atol=2e-2 * math.sqrt(K / 1024.0))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs2 = [torch.empty_like(o) for o in outs]
    torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(s1):
            for (a, w), o in zip(ops, outs):
                kn.gemm(a, w, o)
        with torch.cuda.stream(s2):
            for (a, w), o in zip(reversed(ops), reversed(outs2)):
                kn.gemm(a, w, o)
    torch.cuda.synchronize()
    for o, o2, r, (M, N, K) in zip(outs, outs2, refs, shapes):
        check(o, r, "stream 1", atol=2e-2 * math.sqrt(K / 1024.0))
        assert torch.equal(o, o2), "the two streams' results differ"


def test_nt_fused_split_under_graph_replay():
    """The split inside a captured graph (the small-batch step replays graphs): the workspace of the capture stream is the one the
    replays use; results equal the eager ones bit for bit."""
    a, w = rnd(1152, 4096, seed=1, std=0.5), rnd(1024, 4096, seed=2, std=0.5)
    bias = rnd(1024, dtype=torch.float32, seed=3)
    eager = torch.empty(1152, 1024, device=DEV, dtype=BF)
    kn.gemm(a, w, eager, bias=bias, epi=kn.EPI_GELU, aux=torch.empty_like(eager))
    out, aux = torch.empty_like(eager), torch.empty_like(eager)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        kn.gemm(a, w, out, bias=bias, epi=kn.EPI_GELU, aux=aux)          # warm-up on the capture stream: its workspace exists before the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(3):
            kn.gemm(a, w, out, bias=bias, epi=kn.EPI_GELU, aux=aux)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)


@pytest.mark.parametrize("R", [640, 1152, 2304])
@pytest.mark.parametrize("n_out,k_in", [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)])
def test_wgrad_at_few_rows_takes_the_rule_of_the_engine(R, n_out, k_in):
    """dW[n_out, k_in] += dy[R, n_out]^T x[R, k_in] with the split count engine.splitk_rule picks at few rows (unsplit where the
    128x128 tile list covers half the CUs), as Engine.wgrad issues it, against an fp32 product."""
    sk = splitk_rule(n_out, k_in, R)
    t = tiles128(n_out, k_in)
    assert sk == max(1, min(256 // t, (R // 32) // 16))
    dy, x = rnd(R, n_out, seed=1, std=0.5), rnd(R, k_in, seed=2, std=0.5)
    g0 = rnd(n_out, k_in, dtype=torch.float32, seed=3)
    out = g0.clone()
    if sk > 1:
        ws = torch.empty(sk * n_out, k_in, device=DEV)
        kn.gemm(dy, x, ws, a_t=True, b_t=True, splitk=sk, slabs=True)
        kn.slab_reduce(ws, sk, out, accumulate=True)
    else:
        kn.gemm(dy, x, out, a_t=True, b_t=True, accumulate=True)
    ref = g0 + dy.float().t() @ x.float()
    check(out, ref, "dW", rel=1e-4, atol=2e-3 * math.sqrt(R / 1024.0))
