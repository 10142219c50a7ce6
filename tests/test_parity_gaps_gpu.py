"""GPU: the parity gaps VERDICT r05 listed, closed:

  (a) DROPOUT ON (cfg/bart-large.json:23 dropout 0.1, modeling_multimodalsum.py:294,305,371,458,474,486,596 -- the setting bench.py
      times).  The HIP masks are a counter hash of (site seed, salt, row * D + column), not torch's Philox stream, so the oracle is
      handed the SAME masks (multimodalsum_amd/dropout.py restates the hash on the host; oracle/bart_oracle.DROPOUT_MASKS injects them
      in the order the reference reaches its dropout sites) and the fused step is compared tensor by tensor: f32 1e-3, bf16 3x the
      oracle's own bf16 emulation error + 1e-3 -- eager and under graph replay (salted seeds), multimodal and text-only.
  (b) BASELINE config 2 at its own width: TextSupervised at D 1024 / F 4096 / V 50265 / [2, 9, 128], 2 + 2 layers, f32 and bf16,
      against step_oracle.text_step_loss (src/text_pretrain.py:66-113) -- the twin of test_text_table_step_bf16_config3.
  (c) the Amazon fused multimodal step (133 table positions, I = 1; src/table_encoder.py:86-167) against the oracle's multimodal step
      with the Amazon table encoder, instead of a finiteness check.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodalsum_amd import dropout as mdrop
from multimodalsum_amd import synthetic as syn
from multimodalsum_amd.config import BartConfig
from multimodalsum_amd.formula_init import formula_state_dict
from oracle import bart_oracle as bo
from oracle import encoders_oracle as eo
from oracle import step_oracle as so
from tests.test_host_logic_cpu import tiny_cfg, oracle_cfg, f3_state

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StepMasks:
    """Dropout masks of one fused HIP step, served in the ORACLE's call order.

    Engine order of sites (engine.next_seed): encoder embedding; per encoder layer self-attention, FFN; decoder embedding; per decoder
    layer self-attention, cross-attention, FFN -- the NR leave-one-out passes are ONE decoder call whose rows are (business, pass, t).
    Oracle order: the same encoder sites ([Bn, S, D] for the embedding, time-major [S, Bn, D] inside the layers), then per pass i the
    decoder sites on [B, T, D] / [T, B, D]."""

    def __init__(self, seeds, p, B, NR, S, D, Le, Ld, reviews_mask, compact, salt=None):
        assert len(seeds) == 1 + 2 * Le + 1 + 3 * Ld, (len(seeds), Le, Ld)
        self.p, self.B, self.NR, self.S, self.D, self.salt = p, B, NR, S, D, salt
        self.enc_seeds, self.dec_seeds = seeds[:1 + 2 * Le], seeds[1 + 2 * Le:]
        Bn = B * NR
        flat = reviews_mask.reshape(-1).bool().cpu()
        self.live = flat
        # rows as the layer kernels saw them: compact index of a live row (padding-free encoder), else the padded row itself
        self.enc_rows = (torch.cumsum(flat.long(), 0) - 1).clamp(min=0) if compact else torch.arange(Bn * S)
        self.compact = compact
        self.calls = 0
        self.order = [("enc0", 0)] + [("enc", 1 + j) for j in range(2 * Le)]
        for i in range(NR):
            self.order += [("dec0", 0, i)] + [("dec", 1 + j, i) for j in range(3 * Ld)]

    def __call__(self, shape):
        site = self.order[self.calls]
        self.calls += 1
        B, NR, S, D = self.B, self.NR, self.S, self.D
        if site[0] in ("enc0", "enc"):
            Bn = B * NR
            if site[0] == "enc0":          # the embedding kernel runs on the padded layout, before the compaction
                m = mdrop.keep_mask(self.enc_seeds[0], Bn * S, D, self.p, self.salt).view(Bn, S, D)
                assert shape == (Bn, S, D), shape
                return m
            m = mdrop.keep_mask(self.enc_seeds[site[1]], self.enc_rows, D, self.p, self.salt)
            if self.compact:
                m = torch.where(self.live.view(-1, 1), m, torch.ones_like(m))      # rows the kernels never ran: nobody reads them
            assert shape == (S, Bn, D), shape
            return m.view(Bn, S, D).transpose(0, 1).contiguous()
        i = site[2]
        T = S
        rows = ((torch.arange(B).view(B, 1) * NR + i) * T + torch.arange(T).view(1, T)).reshape(-1)
        m = mdrop.keep_mask(self.dec_seeds[site[1]], rows, D, self.p, self.salt).view(B, T, D)
        if site[0] == "dec0":
            assert shape == (B, T, D), shape
            return m
        assert shape == (T, B, D), shape
        return m.transpose(0, 1).contiguous()


def _oracle_with_masks(fn, masks, emulate, dt):
    bo.DROPOUT_MASKS, bo.EMULATE_BF16 = masks, emulate
    masks.calls = 0
    try:
        return fn(dt)
    finally:
        bo.DROPOUT_MASKS, bo.EMULATE_BF16 = None, False
        assert masks.calls == len(masks.order), (masks.calls, len(masks.order))


def _state(sd, dt):
    return {k: (v.detach().to(dt).clone().requires_grad_(v.dim() > 0 and "running" not in k) if v.is_floating_point() else v.clone())
            for k, v in sd.items()}


def _compare(gh, g64, gemu, g32, dtype, skip=("img_encoder.resnet",), k_rows=None, k32=3):
    worst = []
    for n, ref in g64.items():
        if float(ref.abs().max()) <= 1e-9 or any(s in n for s in skip):
            continue
        assert n in gh and torch.isfinite(gh[n]).all(), n
        if dtype == torch.float32:
            e = float((gh[n].double() - ref).abs().max()) / float(ref.abs().max())
            e32 = float((g32[n].double() - ref).abs().max()) / float(ref.abs().max())
            worst.append((e / max(1e-3, k32 * e32), n, e, e32))
        else:
            nrm = float(ref.norm()) + 1e-30
            e, ee = float((gh[n].double() - ref).norm()) / nrm, float((gemu[n].double() - ref).norm()) / nrm
            k = (k_rows or {}).get(n, 3)
            worst.append((e / (k * ee + 1e-3), n, e, ee))
    worst.sort(reverse=True)
    assert len(worst) > 10
    assert worst[0][0] <= 1.0, "gradients beyond tolerance (ratio, name, HIP error, yardstick): %r" % (worst[:6],)
    return worst


# ------------------------------------------------------------------------------------------------
# (a) dropout on
# ------------------------------------------------------------------------------------------------
def test_host_mask_statement_matches_the_kernels():
    """dropout.keep_mask against the device, element by element: kernels.add_ln_fwd with p = 0.5 on rows of ones and a zero residual --
    a kept element (2.0) normalises to a positive value, a dropped one (0.0) to a negative one -- with and without a salt."""
    from multimodalsum_amd import kernels as kn
    R, D, p, seed = 37, 1024, 0.5, 0x5EED00012345
    x = torch.ones(R, D, device=DEV)
    res = torch.zeros(R, D, device=DEV)
    y, mean, rstd = torch.empty(R, D, device=DEV), torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    for salt in (None, 5):
        st = None if salt is None else torch.tensor([salt], dtype=torch.int64, device=DEV)
        kn.add_ln_fwd(x, res, g, b, y, mean, rstd, 1e-5, p, seed, salt=st)
        torch.cuda.synchronize()
        keep = mdrop.keep_mask(seed, R, D, p, salt)
        assert torch.equal((y > 0).cpu(), keep), ("salt", salt)
        assert 0.45 < keep.float().mean() < 0.55


def _dropout_step(model_kind, dtype, graphs):
    from multimodalsum_amd.modules import MultimodalSum, TextSupervised
    p = 0.1
    cfg = tiny_cfg(vocab=200, d=1024, ffn=256, layers=2, heads=16, maxpos=40, dropout=p)
    ocfg = oracle_cfg(cfg)
    B, NR, S, I = 2, 3, 24, 2
    bc = syn.yelp_batch(B, NR, S, I, cfg.vocab_size, seed=91, img_hw=224)      # (224: 196 positions per image -- BatchNorm statistics over a handful of positions of small images make the f32 ResNet stack chaotic)
    if model_kind == "multimodal":
        sd = f3_state(ocfg)
        model = MultimodalSum(config=cfg, label_smoothing=0.1, device=DEV, dtype=dtype, deterministic=(dtype == torch.float32))
        args = lambda b: (b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])
    else:
        sd = formula_state_dict(bo.bart_param_shapes(ocfg, False, prefix="bart_model."), std=0.02)
        model = TextSupervised(config=cfg, label_smoothing=0.1, device=DEV, dtype=dtype, deterministic=(dtype == torch.float32))
        args = lambda b: (b["reviews"], b["reviews_mask"], b["reviews_rating"])
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    model.train()
    e = model._engine
    assert e.p_drop() == p
    b = syn.batch_to(bc, DEV)
    if graphs:
        # first forward of a shape runs eagerly (warm-up), the second captures and replays, the third replays: the seeds are those
        # the CAPTURE drew (baked into the graph's kernel nodes), the salt is the device counter the forward graph bumps before its
        # first kernel -- the masks compared are those of the last replay
        model.enable_step_graphs()
        seeds = None
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            e.seed_log = []
            loss = model(*args(b))[0]
            loss.backward()
            if e.seed_log:
                seeds = list(e.seed_log)
        assert model._step_graphs.captures == 1
        salt = int(model._step_graphs.salt.item())
        assert salt == 2
    else:
        e.seed_log = []
        loss = model(*args(b))[0]
        loss.backward()
        salt, seeds = None, list(e.seed_log)
    torch.cuda.synchronize()
    compact = bool(getattr(model, "compact_encoder", True))
    masks = StepMasks(seeds, p, B, NR, S, cfg.d_model, cfg.encoder_layers, cfg.decoder_layers, bc["reviews_mask"], compact, salt)
    gh = {n: q.grad.detach().float().cpu() for n, q in model.named_parameters() if q.grad is not None}

    def run(dt):
        st = _state(sd, dt)
        if model_kind == "multimodal":
            ol = so.multimodal_step_loss(st, ocfg, bc["reviews"], bc["reviews_mask"], bc["reviews_rating"].to(dt), bc["field"],
                                         bc["field_value"], bc["img"].to(dt), bc["img_mask"], 0.1, training=True)
        else:
            ol = so.text_step_loss(st, ocfg, bc["reviews"], bc["reviews_mask"], bc["reviews_rating"].to(dt), 0.1, training=True)
        ol.backward()
        return float(ol.detach()), {k: v.grad for k, v in st.items() if getattr(v, "grad", None) is not None}

    l64, g64 = _oracle_with_masks(run, masks, False, torch.float64)
    l32, g32 = _oracle_with_masks(run, masks, False, torch.float32)
    # the same step WITHOUT dropout must be far from this one: the comparison below is not vacuous
    bo_cfg_p, ocfg.dropout = ocfg.dropout, 0.0
    l_nodrop, _ = run(torch.float64)
    ocfg.dropout = bo_cfg_p
    assert abs(l_nodrop - l64) > 20 * 1e-3 * abs(l64), (l_nodrop, l64)
    lh = float(loss.detach())
    if dtype == torch.float32:
        assert abs(lh - l64) <= 1e-3 * abs(l64), (lh, l64)
        _compare(gh, g64, None, g32, dtype)
    else:
        lemu, gemu = _oracle_with_masks(run, masks, True, torch.float32)
        assert abs(lh - l64) <= 3 * abs(lemu - l64) + 1e-3 * abs(l64), (lh, lemu, l64)
        # (single-row gradients -- the table's rating embedding, B = 2 memory rows -- are held to 5x, as in test_text_table_step_bf16_config3)
        _compare(gh, g64, gemu, g32, dtype, k_rows={"table_encoder.rating_embedding.weight": 5, "table_encoder.hours_embedding.weight": 5})


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("kind", ["multimodal", "text"])
def test_fused_step_with_dropout_on_vs_oracle_on_the_same_masks(kind, dtype):
    _dropout_step(kind, dtype, graphs=False)


def test_fused_step_with_dropout_on_under_graph_replay_bf16():
    """The timed path: captured graphs, salted seeds (every replay draws fresh masks); the second replay's masks rebuilt on the host."""
    _dropout_step("multimodal", torch.bfloat16, graphs=True)


# ------------------------------------------------------------------------------------------------
# (b) BASELINE config 2 at its own width
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_text_only_step_config2_at_width(dtype):
    """TextSupervised (src/text_pretrain.py:66-113) at D 1024 / F 4096 / H 16 / V 50265, S = T = 128, [2, 9, 128] reviews, 2 + 2 layers:
    the unimodal decoder branch (BartForEncConditionalGeneration: one key tensor, no gate) through the 256x256 GEMM kernels, the
    128-key attention kernels and the register-resident loss kernel `also.text_only_B128` times.  Loss and every gradient vs
    step_oracle.text_step_loss in fp64: f32 mode 1e-3 (3x the fp32 oracle's own error where larger), bf16 3x the emulation + 1e-3."""
    from multimodalsum_amd.modules import TextSupervised
    cfg = BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    cfg.dropout = 0.0
    cfg.encoder_layers = cfg.decoder_layers = 2
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=2, decoder_layers=2,
                      heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.0)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, False, prefix="bart_model."), std=0.02)
    bc = syn.yelp_batch(2, 9, 128, 1, cfg.vocab_size, seed=202, img_hw=8)
    b = syn.batch_to(bc, DEV)
    model = TextSupervised(config=cfg, label_smoothing=0.1, device=DEV, dtype=dtype, deterministic=(dtype == torch.float32))
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    model.train()
    loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"])[0]
    loss.backward()
    torch.cuda.synchronize()
    gh = {n: q.grad.detach().float().cpu() for n, q in model.named_parameters() if q.grad is not None}

    def run(dt, emulate=False):
        bo.EMULATE_BF16 = emulate
        try:
            st = _state(sd, dt)
            ol = so.text_step_loss(st, ocfg, bc["reviews"], bc["reviews_mask"], bc["reviews_rating"].to(dt), 0.1, training=True)
            ol.backward()
            return float(ol.detach()), {k: v.grad for k, v in st.items() if getattr(v, "grad", None) is not None}
        finally:
            bo.EMULATE_BF16 = False

    l64, g64 = run(torch.float64)
    lh = float(loss.detach())
    if dtype == torch.float32:
        l32, g32 = run(torch.float32)
        assert abs(lh - l64) <= 1e-3 * abs(l64), (lh, l64)
        _compare(gh, g64, None, g32, dtype)
    else:
        lemu, gemu = run(torch.float32, True)
        assert abs(lh - l64) <= 3 * abs(lemu - l64) + 1e-3 * abs(l64), (lh, lemu, l64)
        _compare(gh, g64, gemu, None, dtype)


# ------------------------------------------------------------------------------------------------
# (c) Amazon fused multimodal step
# ------------------------------------------------------------------------------------------------
def _amazon_oracle_step(st, ocfg, bc, field, fv, dt):
    """step_oracle.multimodal_step_loss with the Amazon table encoder in place of the Yelp one (multimodal_train.py:124-193 with
    --dataset amazon: AmazonTableEncoder, table_encoder.py:86-167; one image per product)."""
    B, NR, S = bc["reviews"].shape
    text_h = bo.bart_encoder(st, ocfg, bc["reviews"].view(B * NR, S), bc["reviews_mask"].view(B * NR, S), True,
                             prefix="bart_model.").view(B, NR, S, -1)
    table_h, table_m = eo.amazon_table_encoder(st, st["bart_model.model.shared.weight"], field, fv)
    img = bc["img"].to(dt)
    I = img.shape[1]
    img_h = eo.resnet101_features(st, img.reshape(-1, 3, img.shape[-2], img.shape[-1]), True, None).reshape(B, I, -1, ocfg.d_model)
    img_m = bc["img_mask"].unsqueeze(-1).repeat(1, 1, img_h.shape[2])
    rr = bc["reviews_rating"].to(dt)
    losses = []
    for i in range(NR):
        others = [j for j in range(NR) if j != i]
        rd = rr[:, i] - rr[:, others].mean(dim=1)
        logits = bo.multienc_forward(st, ocfg, text_h[:, others], bc["reviews_mask"][:, others], table_h.unsqueeze(1), table_m.unsqueeze(1),
                                     img_h, img_m, rd.unsqueeze(1), bc["reviews"][:, i], True, prefix="bart_model.")
        losses.append(bo.label_smoothing_loss(logits.view(-1, ocfg.vocab_size), bc["reviews"][:, i].reshape(-1), ocfg.vocab_size, 0.1))
    return torch.mean(torch.stack(losses))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_amazon_fused_multimodal_step_vs_oracle(dtype):
    from multimodalsum_amd.modules import MultimodalSum, AmazonTableEncoder
    cfg = tiny_cfg(vocab=200, d=1024, ffn=64, layers=1, heads=16, maxpos=32)
    ocfg = oracle_cfg(cfg)
    shapes = bo.bart_param_shapes(ocfg, True, prefix="bart_model.")
    shapes.update(eo.amazon_table_param_shapes())
    sd = formula_state_dict(shapes, std=0.02)
    sd.update(formula_state_dict(eo.resnet_param_shapes(1024), std=0.05))
    field, fv = syn.amazon_table_batch(2, cfg.vocab_size, seed=9)
    bc = syn.yelp_batch(2, 3, 16, 1, cfg.vocab_size, seed=41, img_hw=224)
    bc["img_mask"][:] = True
    bc["img_mask"][1, 0] = False                   # one product without an image: beta gate closed for it (:732-736)
    model = MultimodalSum(config=cfg, label_smoothing=0.1, device=DEV, dtype=dtype, TableEncoder=AmazonTableEncoder,
                          deterministic=(dtype == torch.float32))
    model.load_state_dict({k: v.detach() for k, v in sd.items()})
    model.train()
    b = syn.batch_to(bc, DEV)
    loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], field.to(DEV), [t.to(DEV) for t in fv], b["img"], b["img_mask"])[0]
    loss.backward()
    torch.cuda.synchronize()
    gh = {n: q.grad.detach().float().cpu() for n, q in model.named_parameters() if q.grad is not None}

    def run(dt, emulate=False):
        bo.EMULATE_BF16 = emulate
        try:
            st = _state(sd, dt)
            ol = _amazon_oracle_step(st, ocfg, bc, field, fv, dt)
            ol.backward()
            return float(ol.detach()), {k: v.grad for k, v in st.items() if getattr(v, "grad", None) is not None}
        finally:
            bo.EMULATE_BF16 = False

    l64, g64 = run(torch.float64)
    lh = float(loss.detach())
    for n in ("table_encoder.price_embedding.weight", "table_encoder.rating_embedding.weight", "table_encoder.fc.weight", "table_encoder.linear.weight"):
        assert n in g64 and float(g64[n].abs().max()) > 0, n
    if dtype == torch.float32:
        l32, g32 = run(torch.float32)
        assert abs(lh - l64) <= 1e-3 * abs(l64), (lh, l64)
        _compare(gh, g64, None, g32, dtype)
    else:
        lemu, gemu = run(torch.float32, True)
        assert abs(lh - l64) <= 3 * abs(lemu - l64) + 1e-3 * abs(l64), (lh, lemu, l64)
        _compare(gh, g64, gemu, None, dtype, k_rows={"table_encoder.price_embedding.weight": 5, "table_encoder.rating_embedding.weight": 5})


# ------------------------------------------------------------------------------------------------
# (d) ResNet101 stages 1-3: per-tensor gradients on well-conditioned statistics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_resnet_gradients_per_tensor_on_224px_images(dtype):
    """Row I (Resnet.forward, src/img_encoder.py:21-41) with BatchNorm statistics over 4 x 196 positions -- four 224 x 224 images, the
    size the model is fed -- instead of the handful of positions of the small images other tests use (where the fp32 stack is chaotic
    and only error DISTRIBUTIONS can be held): ImgSupervised step (img_pretrain.py:85-141), EVERY image-encoder gradient tensor
    against the oracle in fp64 -- f32 mode within max(1e-3, 4x the fp32 oracle's own error) of the tensor's largest entry, bf16 (the
    timed path: implicit convolutions, statistics from the GEMM epilogues) within 3x the oracle's bf16 emulation + 1e-3 in relative
    L2.  (The backbone definition itself stays unpinned: torchvision is in neither the reference tree nor the image.)"""
    from multimodalsum_amd.modules import ImgSupervised
    cfg = tiny_cfg(vocab=200, d=1024, ffn=64, layers=1, heads=16, maxpos=32)
    ocfg = oracle_cfg(cfg)
    labels = syn.token_batch(2, 12, cfg.vocab_size, seed=5, min_len=4)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, False, prefix="bart_model."), std=0.02)
    sd.update(formula_state_dict(eo.resnet_param_shapes(1024), std=0.05))
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn(2, 2, 3, 224, 224, generator=g)
    imask = torch.tensor([[True, True], [True, True]])
    im = ImgSupervised(config=cfg, label_smoothing=0.1, device=DEV, dtype=dtype, deterministic=(dtype == torch.float32))
    im.load_state_dict({k: v.detach() for k, v in sd.items()})
    im.train()
    loss = im(imgs.to(DEV), imask.to(DEV), labels=labels.to(DEV))[0]
    loss.backward()
    torch.cuda.synchronize()
    gh = {n: q.grad.detach().float().cpu() for n, q in im.named_parameters() if q.grad is not None}

    def run(dt, emulate=False):
        bo.EMULATE_BF16 = emulate
        try:
            st = _state(sd, dt)
            ih = eo.resnet101_features(st, imgs.reshape(-1, 3, 224, 224).to(dt), training=True).reshape(2, 2, -1, 1024)
            lg = bo.enc_forward(st, ocfg, ih, torch.zeros(2, 1, dtype=dt), imask.unsqueeze(-1).repeat(1, 1, ih.shape[2]), labels,
                                training=True, prefix="bart_model.")
            ls = bo.label_smoothing_loss(lg.view(-1, cfg.vocab_size), labels.view(-1), cfg.vocab_size, 0.1)
            ls.backward()
            return float(ls.detach()), {k: v.grad for k, v in st.items() if getattr(v, "grad", None) is not None}
        finally:
            bo.EMULATE_BF16 = False

    l64, g64 = run(torch.float64)
    lh = float(loss.detach())
    img = {n: r for n, r in g64.items() if n.startswith("img_encoder") and float(r.abs().max()) > 1e-12}
    assert len(img) > 90                                     # layer3's 23 bottlenecks + the projection
    if dtype == torch.float32:
        l32, g32 = run(torch.float32)
        assert abs(lh - l64) <= 1e-3 * abs(l64), (lh, l64)
        # (even on these statistics the fp32 stack of 23 bottlenecks is 2 - 3 % from fp64 on single layer3 tensors -- the oracle's OWN fp32 run,
        # measured -- and the HIP f32 path, a different summation order, lands within ~3x of that: held to 4x, per tensor)
        worst = _compare(gh, img, None, g32, dtype, skip=(), k32=4)
    else:
        lemu, gemu = run(torch.float32, True)
        assert abs(lh - l64) <= 3 * abs(lemu - l64) + 1e-3 * abs(l64), (lh, lemu, l64)
        worst = _compare(gh, img, gemu, None, dtype, skip=())
    print("resnet per-tensor gradients (%s): worst ratio to the bound %.3f on %s (error %.2e, yardstick %.2e)" % ((str(dtype),) + worst[0]))
