"""GPU: sequences of more than 128 tokens through the encoder (reference: BartEncoder.forward,
src/transformer/modeling_multimodalsum.py:346-404, takes any length up to the position table; src/test.py:56-60 tokenises Yelp
reviews to 160 -> 158 tokens after the strip at src/data_utils.py:48-52, Amazon to 118).  The attention kernels take query blocks
of at most 128 rows, so engine.encoder_fwd cuts a longer sequence into two query blocks that share the sequence's keys (<= 224):

  * encoder forward + backward at S = 158 (even: no internal padding), 141 and 159 (odd: one internal padding column), with
    ragged lengths (3 .. S tokens; an all-padding encoder row is NaN in the reference itself), f32 (1e-3) and bf16 (3x the oracle's own bf16 emulation error + 1e-3);
  * the padding-free (compact row) encoder of the fused steps at S = 158 against the padded one;
  * the decoder's cross-attention over 158-token text entities (teacher-forced pass, T = 24) against the oracle;
  * beam search on [B, 8, 158] text + table + images: tests/test_generation_gpu.py::test_generation_f32_on_158_token_reviews.
The decoder's own (causal) sequence stays <= 128 positions: the reference trains on 128-token targets
(src/multimodal_train.py:30) and test.py generates max_length <= 128; engine._self_block_fwd raises above that.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodalsum_amd import synthetic as syn
from multimodalsum_amd.formula_init import formula_state_dict, formula_tensor
from oracle import bart_oracle as bo
from tests.test_host_logic_cpu import tiny_cfg, oracle_cfg

DEV = "cuda"


def _ids(Bn, S, vocab, seed):
    ids = syn.token_batch(Bn, S, vocab, seed=seed, mean_len=0.8 * S, std_len=0.15 * S, min_len=S // 3)
    ids[0] = torch.randint(3, vocab, (S,), generator=torch.Generator().manual_seed(seed))      # one full-length row
    ids[Bn - 1, 3:] = 1                                                                           # one three-token row
    return ids


def _encoder_case(S, dtype, d=256, ffn=512, layers=2, heads=4):
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    cfg = tiny_cfg(vocab=300, d=d, ffn=ffn, layers=layers, heads=heads, maxpos=S + 4)
    ocfg = oracle_cfg(cfg)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=0.08)
    model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=dtype, deterministic=True)
    model.load_state_dict(sd)
    model.train()
    Bn = 5
    ids = _ids(Bn, S, cfg.vocab_size, 77 + S)
    mask = ids.ne(1)
    w = formula_tensor("long.w", (Bn, S, d), std=1.0) * mask.unsqueeze(-1)          # padded rows' outputs are unspecified (masked keys downstream)
    enc = model.model.encoder(input_ids=ids.to(DEV), attention_mask=mask.to(DEV))[0]
    assert enc.shape == (Bn, S, d)
    (enc.float() * w.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    enc_names = [k for k in sd if k.startswith("model.encoder.") or k == "model.shared.weight"]

    def oracle(quant):
        for k in enc_names:
            sd[k].grad = None
            sd[k].requires_grad_(True)
        bo.EMULATE_BF16 = quant
        try:
            o = bo.bart_encoder(sd, ocfg, ids, mask, training=True)
            (o * w).sum().backward()
        finally:
            bo.EMULATE_BF16 = False
        return o.detach(), {k: sd[k].grad.clone() for k in enc_names}

    return model, enc.detach().float().cpu(), mask, oracle, enc_names


def _err(a, b, mask=None):
    d = (a.double() - b.double()).abs()
    if mask is not None:
        d = d * mask.unsqueeze(-1)
    return d.max().item()


@pytest.mark.parametrize("S", [158, 141, 159, 224])
def test_encoder_longer_than_128_tokens_f32(S):
    model, enc, mask, oracle, names = _encoder_case(S, torch.float32)
    o, g = oracle(False)
    assert _err(enc, o, mask) <= 1e-3 * o.abs().max().item(), _err(enc, o, mask)
    named = dict(model.named_parameters())
    for k in names:
        ref = g[k]
        err = _err(named[k].grad.cpu(), ref)
        # (a key bias shifts every score of a row alike: its exact gradient is 0 and both sides hold rounding of the q-bias scale)
        atol = 1e-3 * g[k.replace("k_proj", "q_proj")].abs().max().item() if k.endswith("k_proj.bias") else 2e-6
        assert err <= atol + 1e-3 * ref.abs().max().item(), (k, err, ref.abs().max().item())


@pytest.mark.parametrize("S", [158, 159])
def test_encoder_longer_than_128_tokens_bf16(S):
    model, enc, mask, oracle, names = _encoder_case(S, torch.bfloat16)
    o, g = oracle(False)
    oq, gq = oracle(True)
    yard = _err(oq, o, mask)
    assert _err(enc, o, mask) <= 3 * yard + 1e-3 * o.abs().max().item(), (_err(enc, o, mask), yard)
    named = dict(model.named_parameters())
    for k in names:
        ref = g[k]
        if ref.numel() < 256 or k.endswith("k_proj.bias"):
            continue
        err, y = _err(named[k].grad.cpu(), ref), _err(gq[k], ref)
        assert err <= 3 * y + 1e-3 * ref.abs().max().item() + 1e-6, (k, err, y)


def test_encoder_over_224_tokens_is_refused():
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    cfg = tiny_cfg(vocab=300, d=256, ffn=128, layers=1, heads=4, maxpos=260)
    model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=torch.float32, deterministic=True)
    ids = torch.randint(3, 300, (2, 230))
    with pytest.raises(ValueError, match="224"):
        model.model.encoder(input_ids=ids.to(DEV), attention_mask=ids.ne(1).to(DEV))


@pytest.mark.parametrize("S", [158, 159])
def test_padding_free_encoder_equals_padded_at_158_tokens(S):
    """engine.encoder_fwd(compact=True) -- the fused steps' encoder on the live rows only, attention through int32 row maps --
    against the padded schedule, forward and every gradient, bf16 (the mode the row maps exist in)."""
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    cfg = tiny_cfg(vocab=300, d=256, ffn=512, layers=2, heads=4, maxpos=S + 4)
    sd = formula_state_dict(bo.bart_param_shapes(oracle_cfg(cfg), True, prefix=""), std=0.08)
    Bn = 6
    ids = _ids(Bn, S, cfg.vocab_size, 5).to(DEV)
    mask = ids.ne(1)
    dout = (formula_tensor("long.d", (Bn * S, 256), std=1.0).to(DEV) * mask.reshape(-1, 1)).to(torch.bfloat16)
    res = []
    for compact in (False, True):
        model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=torch.bfloat16, deterministic=True)
        model.load_state_dict(sd)
        e = model._engine
        e.sync_weights()
        e.arena.prepare_grads()
        e.touched = set()
        x, c = e.encoder_fwd(ids, mask, compact=compact)
        assert x.shape == (Bn * S, 256)
        e.encoder_bwd(c, dout.clone())
        torch.cuda.synchronize()
        res.append((x.float() * mask.reshape(-1, 1), e.arena.grad.clone()))
    (x0, g0), (x1, g1) = res
    assert (x0 - x1).abs().max().item() <= 2e-2 * x0.abs().max().item()
    assert (g0 - g1).abs().max().item() <= 2e-2 * g0.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_cross_attention_over_158_token_entities(dtype):
    """Teacher-forced multi-encoder pass (modeling_multimodalsum.py:819-869) whose text entities are 158 keys long -- the memory
    test.py's inputs produce -- with a table and images: logits and the gradients of the memory against the oracle."""
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    cfg = tiny_cfg(vocab=300, d=256, ffn=512, layers=2, heads=4, maxpos=200)
    ocfg = oracle_cfg(cfg)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=0.08)
    model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=dtype, deterministic=True)
    model.load_state_dict(sd)
    model.train()
    Bz, N, S, T, D = 2, 3, 158, 24, 256
    ids = syn.token_batch(Bz * N, S, cfg.vocab_size, seed=3, mean_len=120.0, std_len=30.0, min_len=40).view(Bz, N, S)
    text_m = ids.ne(1).clone()
    text_m[1, 2, :] = False
    text_h = formula_tensor("l.text_h", (Bz, N, S, D), std=1.0)
    table_h = formula_tensor("l.table_h", (Bz, 1, 47, D), std=1.0)
    img_h = formula_tensor("l.img_h", (Bz, 2, 196, D), std=1.0)
    table_m = torch.ones(Bz, 1, 47, dtype=torch.bool)
    img_m = torch.ones(Bz, 2, 196, dtype=torch.bool)
    img_m[0, 1] = False
    labels = syn.token_batch(Bz, T, cfg.vocab_size, seed=12, min_len=8)
    rd = torch.tensor([[0.5], [-1.25]])
    cast = lambda t: t.to(DEV).to(dtype)
    hd = [cast(text_h).requires_grad_(True), cast(table_h).requires_grad_(True), cast(img_h).requires_grad_(True)]
    out = model(hd[0], text_m.to(DEV), hd[1], table_m.to(DEV), hd[2], img_m.to(DEV), rating_diff=rd.to(DEV), labels=labels.to(DEV))
    logits = out[0]
    wl = formula_tensor("l.wl", tuple(logits.shape), std=1.0)
    (logits.float() * wl.to(DEV)).sum().backward()
    oh = [t.clone().requires_grad_(True) for t in (text_h, table_h, img_h)]

    def run(quant):
        bo.EMULATE_BF16 = quant
        try:
            for t in oh:
                t.grad = None
            ol = bo.multienc_forward(sd, ocfg, oh[0], text_m, oh[1], table_m, oh[2], img_m, rd, labels, training=True)
            (ol * wl).sum().backward()
        finally:
            bo.EMULATE_BF16 = False
        return ol.detach(), [t.grad.clone() for t in oh]

    ol, og = run(False)
    if dtype == torch.float32:
        assert _err(logits.cpu(), ol) <= 1e-3 * ol.abs().max().item()
        for t, g in zip(hd, og):
            assert _err(t.grad.cpu(), g) <= 1e-3 * g.abs().max().item() + 1e-6
    else:
        oq, gq = run(True)
        assert _err(logits.cpu(), ol) <= 3 * _err(oq, ol) + 1e-3 * ol.abs().max().item()
        for t, g, q in zip(hd, og, gq):
            assert _err(t.grad.cpu(), g) <= 3 * _err(q, g) + 1e-3 * g.abs().max().item() + 1e-6
