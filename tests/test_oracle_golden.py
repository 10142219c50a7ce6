"""CPU: the oracle restatement vs golden vectors produced by running the reference itself
(oracle/make_golden.py, development container).  This is what pins the oracle."""
import argparse
import os

import numpy as np
import torch

from multimodalsum_amd import synthetic as syn
from multimodalsum_amd.formula_init import formula_state_dict, formula_tensor
from oracle import bart_oracle as bo
from oracle import encoders_oracle as eo
from oracle import step_oracle as so

TOL = dict(rtol=2e-4, atol=2e-5)


def _load(golden_dir, name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _close(a, b, rtol=2e-4, atol=2e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    assert err <= atol + rtol * ref, "max err %.3e vs ref max %.3e" % (err, ref)


def test_f7_shift_tokens_right(golden_dir):
    g = _load(golden_dir, "f7_shift.npz")
    assert torch.equal(bo.shift_tokens_right(g["in_a"], 1, 0, 2), g["out_a"])
    assert torch.equal(bo.shift_tokens_right(g["in_b"], 1, 0, 2), g["out_b"])


def test_f5_label_smoothing(golden_dir):
    g = _load(golden_dir, "f5_loss.npz")
    V = 50265
    lg = formula_tensor("f5.logits", (8, V), std=2.0).requires_grad_(True)
    loss = bo.label_smoothing_loss(lg, g["target"], V, 0.1)
    loss.backward()
    _close(loss, g["loss"], 1e-6, 1e-6)
    _close(lg.grad[:, :64], g["grad_sample"], 1e-5, 1e-9)
    _close(lg.grad.double().abs().sum(), g["grad_checksum"], 1e-5, 0)


def f1_inputs():
    T, B, D = 5, 3, 64
    names = {}
    for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
        names["f1.%s.weight" % p] = (D, D)
        names["f1.%s.bias" % p] = (D,)
    for p in ("alpha_proj", "beta_proj"):
        names["f1.%s.weight" % p] = (D, 2 * D)
        names["f1.%s.bias" % p] = (D,)
    sd = formula_state_dict(names, std=0.15)
    q = formula_tensor("f1.query", (T, B, D), std=1.0)
    keys = [formula_tensor("f1.ktext", (7, 3, B, D), 1.0), formula_tensor("f1.ktab", (6, 1, B, D), 1.0),
            formula_tensor("f1.kimg", (4, 2, B, D), 1.0)]
    gout = formula_tensor("f1.gout", (T, B, D), std=1.0)
    return sd, q, keys, gout


def test_f1_cross_attention(golden_dir):
    g = _load(golden_dir, "f1_crossattn.npz")
    sd, q, keys, gout = f1_inputs()
    for v in sd.values():
        v.requires_grad_(True)
    q.requires_grad_(True)
    for k in keys:
        k.requires_grad_(True)
    out = bo.cross_attention(sd, "f1", q, keys, [g["ptext"], g["ptab"], g["pimg"]], 4, True)
    out.backward(gout)
    _close(out, g["out"], **TOL)
    _close(q.grad, g["gq"], **TOL)
    _close(keys[0].grad, g["gktext"], **TOL)
    _close(keys[1].grad, g["gktab"], **TOL)
    _close(keys[2].grad, g["gkimg"], **TOL)
    for n, v in sd.items():
        _close(v.grad, g["g_" + n[3:].replace(".", "_")], **TOL)


def tiny_cfg(vocab=100, d=64, ffn=128, layers=2, heads=4, maxpos=64):
    return bo.BartCfg(vocab_size=vocab, d_model=d, ffn_dim=ffn, encoder_layers=layers, decoder_layers=layers,
                      heads=heads, max_position_embeddings=maxpos, dropout=0.0)


def f2_setup(g):
    cfg = tiny_cfg()
    sd = formula_state_dict(bo.bart_param_shapes(cfg, True, prefix="f2."), std=0.08)
    table_h = formula_tensor("f2.table_h", (3, 1, 6, cfg.d_model), std=1.0)
    img_h = formula_tensor("f2.img_h", (3, 2, 4, cfg.d_model), std=1.0)
    return cfg, sd, table_h, img_h


def test_f2_decoder_pass(golden_dir):
    g = _load(golden_dir, "f2_decoder.npz")
    cfg, sd, table_h, img_h = f2_setup(g)
    for v in sd.values():
        v.requires_grad_(True)
    table_h.requires_grad_(True)
    img_h.requires_grad_(True)
    ids = g["ids"]
    Bz, N, S = ids.shape
    enc = bo.bart_encoder(sd, cfg, ids.view(-1, S), ids.view(-1, S).ne(1), prefix="f2.")
    _close(enc, g["enc_out"], **TOL)
    logits = bo.multienc_forward(sd, cfg, enc.view(Bz, N, S, -1), g["text_m"], table_h, g["table_m"], img_h,
                                 g["img_m"], g["rating_diff"], g["labels"], prefix="f2.")
    _close(logits, g["logits"], **TOL)
    loss = bo.label_smoothing_loss(logits.view(-1, cfg.vocab_size), g["labels"].view(-1), cfg.vocab_size, 0.1)
    _close(loss, g["loss"], 1e-5, 1e-6)
    loss.backward()
    _close(table_h.grad, g["g_table_h"], **TOL)
    _close(img_h.grad, g["g_img_h"], **TOL)
    for k in g:
        if k.startswith("g_model_"):
            name = "f2." + [n for n in sd if n[3:].replace(".", "_") == k[2:]][0][3:]
            _close(sd[name].grad, g[k], **TOL)


def test_g1_beam_search(golden_dir):
    """Beam-search restatement (oracle/generate_oracle.py) against token ids produced by the reference's own
    generate() on the F2 model: the test.py call shape (4 beams, 3-gram ban, early stopping) and a variant with
    min_length, length penalty 2, 2-gram ban and the non-early-stopping is_done branch."""
    from oracle import generate_oracle as go
    g = _load(golden_dir, "f2_decoder.npz")
    gg = _load(golden_dir, "g1_beam.npz")
    cfg, sd, table_h, img_h = f2_setup(g)
    ids = gg["ids"]
    Bz, N, S = ids.shape
    with torch.no_grad():
        enc = bo.bart_encoder(sd, cfg, ids.view(-1, S), ids.view(-1, S).ne(1), prefix="f2.").view(Bz, N, S, -1)
        _close(enc, gg["enc_eval"], **TOL)
        hid, msk = [enc, table_h, img_h], [gg["text_m"], gg["table_m"], gg["img_m"]]
        a = go.beam_search(sd, cfg, hid, msk, torch.zeros(Bz, 1), True, num_beams=4, max_length=14, no_repeat_ngram_size=3,
                           early_stopping=True, length_penalty=1.0, prefix="f2.")
        b = go.beam_search(sd, cfg, hid, msk, gg["rating_diff"], True, num_beams=2, max_length=10, min_length=4,
                           no_repeat_ngram_size=2, early_stopping=False, length_penalty=2.0, prefix="f2.")
    assert torch.equal(a, gg["gen_a"]), (a, gg["gen_a"])
    assert torch.equal(b, gg["gen_b"]), (b, gg["gen_b"])
    # sequence_score (the yardstick of the bf16 generation test's tie rule) reproduces the score the search itself gave its best
    # hypothesis, for both call shapes -- teacher-forced on the returned row, pads and the appended EOS stripped
    with torch.no_grad():
        for kw, rd in ((dict(num_beams=4, max_length=14, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0), torch.zeros(Bz, 1)),
                       (dict(num_beams=2, max_length=10, min_length=4, no_repeat_ngram_size=2, early_stopping=False, length_penalty=2.0),
                        gg["rating_diff"])):
            out, scores = go.beam_search(sd, cfg, hid, msk, rd, True, prefix="f2.", return_scores=True, **kw)
            for i in range(Bz):
                got = go.sequence_score(sd, cfg, out[i], [h[i:i + 1] for h in hid], [m[i:i + 1] for m in msk], rd[i:i + 1], True,
                                        kw["max_length"], kw.get("min_length", 0), kw["length_penalty"], prefix="f2.")
                assert abs(got - scores[i]) <= 1e-4 * abs(scores[i]) + 1e-5, (i, got, scores[i])


G3_SAMPLE_CASES = {      # do_sample = True: the recorded uniforms of the fixture take torch.multinomial's place (generate_oracle.inverse_cdf_draw)
    "sample_k": dict(max_length=14, no_repeat_ngram_size=2, top_k=20, temperature=0.8),
    "sample_kp": dict(max_length=16, min_length=5, no_repeat_ngram_size=0, top_k=40, top_p=0.85, temperature=1.3, repetition_penalty=1.2),
}

G3_CASES = {
    "greedy": dict(max_length=14, no_repeat_ngram_size=2),
    "greedy_min": dict(max_length=12, min_length=6, no_repeat_ngram_size=0),
    "greedy_bad": dict(max_length=14, no_repeat_ngram_size=2, bad_words=True),
    "greedy_rep": dict(max_length=14, no_repeat_ngram_size=0, repetition_penalty=1.7),
    "beam_bad": dict(num_beams=4, max_length=14, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0, bad_words=True),
    "beam_rep": dict(num_beams=3, max_length=12, no_repeat_ngram_size=0, early_stopping=False, length_penalty=1.5, repetition_penalty=1.3),
}


def g3_bad_words(gg):
    return [[int(t) for t in row if t >= 0] for row in gg["bad_words"].tolist()]


def test_g3_generate_modes(golden_dir):
    """The generate() modes beside test.py's beam search -- greedy decoding (num_beams = 1, _generate_no_beam_search), bad_words_ids and
    repetition_penalty in both searches, sampling (top-k / top-p / temperature, torch.multinomial pinned to recorded uniforms) -- restated (oracle/generate_oracle.py) and held to the token ids the REFERENCE's own
    generate() produced on the F2 model (tests/golden/g3_generate_modes.npz, oracle/make_golden_r6.py)."""
    from oracle import generate_oracle as go
    g = _load(golden_dir, "f2_decoder.npz")
    gg = _load(golden_dir, "g3_generate_modes.npz")
    cfg, sd, table_h, img_h = f2_setup(g)
    ids = gg["ids"]
    Bz, N, S = ids.shape
    bad = g3_bad_words(gg)
    with torch.no_grad():
        enc = bo.bart_encoder(sd, cfg, ids.view(-1, S), ids.view(-1, S).ne(1), prefix="f2.").view(Bz, N, S, -1)
        _close(enc, gg["enc_eval"], **TOL)
        hid, msk = [enc, table_h, img_h], [gg["text_m"], gg["table_m"], gg["img_m"]]
        for name, kw in G3_CASES.items():
            kw = dict(kw)
            bw = bad if kw.pop("bad_words", False) else None
            if "num_beams" in kw:
                out = go.beam_search(sd, cfg, hid, msk, gg["rating_diff"], True, prefix="f2.", bad_words_ids=bw, **kw)
            else:
                out = go.greedy_search(sd, cfg, hid, msk, gg["rating_diff"], True, prefix="f2.", bad_words_ids=bw, **kw)
            assert torch.equal(out, gg["gen_" + name]), (name, out, gg["gen_" + name])
        for name, kw in G3_SAMPLE_CASES.items():
            out = go.sample_search(sd, cfg, hid, msk, gg["rating_diff"], True, draws=gg["draws_" + name], prefix="f2.", **kw)
            assert torch.equal(out, gg["gen_" + name]), (name, out, gg["gen_" + name])


def test_f2_text_only(golden_dir):
    g = _load(golden_dir, "f2_decoder.npz")
    gt = _load(golden_dir, "f2_textonly.npz")
    cfg, sd, _, _ = f2_setup(g)
    sd = {k: v for k, v in sd.items() if "alpha_proj" not in k and "beta_proj" not in k}
    ids = g["ids"]
    Bz, N, S = ids.shape
    with torch.no_grad():
        enc = bo.bart_encoder(sd, cfg, ids.view(-1, S), ids.view(-1, S).ne(1), prefix="f2.")
        logits = bo.enc_forward(sd, cfg, enc.view(Bz, N, S, -1), g["rating_diff"], g["text_m"], g["labels"], prefix="f2.")
    _close(logits, gt["logits"], **TOL)


def test_table_encoder(golden_dir):
    g = _load(golden_dir, "table_yelp.npz")
    sd = formula_state_dict(eo.table_param_shapes(), std=0.02)
    for v in sd.values():
        v.requires_grad_(True)
    emb = formula_tensor("bart_model.model.shared.weight", (200, 1024), 0.02).requires_grad_(True)
    fv = [g["name"], g["category"], g["str_cat"], g["str_bool"], g["rating"], g["hours"]]
    # the synthetic generator must reproduce the committed inputs bit for bit
    field, fv2 = syn.table_batch(3, 200, seed=21)
    assert torch.equal(field, g["field"]) and all(torch.equal(a, b) for a, b in zip(fv, fv2))
    h, m = eo.yelp_table_encoder(sd, emb, g["field"], fv)
    assert torch.equal(m, g["mask"])
    _close(h, g["hiddens"], **TOL)
    h.backward(formula_tensor("table.gout", h.shape, std=1.0))
    assert emb.grad is None and bool(g["emb_grad_is_none"])
    _close(sd["table_encoder.rating_embedding.weight"].grad, g["g_rating"], **TOL)
    _close(sd["table_encoder.hours_embedding.weight"].grad, g["g_hours"], **TOL)
    _close(sd["table_encoder.fc.weight"].grad[:64], g["g_fc_w"], **TOL)
    _close(sd["table_encoder.fc.bias"].grad, g["g_fc_b"], **TOL)
    _close(sd["table_encoder.linear.weight"].grad[:64], g["g_linear"], **TOL)


def f3_state(cfg):
    shapes = bo.bart_param_shapes(cfg, True, prefix="bart_model.")
    shapes.update(eo.table_param_shapes())
    sd = formula_state_dict(shapes, std=0.02)
    sd.update(formula_state_dict(eo.resnet_param_shapes(1024), std=0.05))
    return sd


def test_f3_leave_one_out_step(golden_dir):
    g = _load(golden_dir, "f3_step.npz")
    cfg = tiny_cfg(vocab=200, d=1024, ffn=64, layers=1, heads=16, maxpos=32)
    sd = f3_state(cfg)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    b = syn.yelp_batch(int(g["B"]), int(g["NR"]), int(g["S"]), int(g["I"]), cfg.vocab_size, seed=int(g["seed"]),
                       img_hw=int(g["img_hw"]))
    loss = so.multimodal_step_loss(sd, cfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"],
                                   b["field_value"], b["img"], b["img_mask"], 0.1, training=True)
    _close(loss, g["loss"], 1e-5, 1e-6)
    loss.backward()
    P = "bart_model.model.decoder."
    _close(sd[P + "rating_embeddings"].grad, g["g_rating"], **TOL)
    _close(sd[P + "layers.0.encoder_attn.alpha_proj.weight"].grad[:32], g["g_alpha"], **TOL)
    _close(sd[P + "layers.0.encoder_attn.beta_proj.bias"].grad, g["g_beta_b"], **TOL)
    _close(sd[P + "layers.0.encoder_attn.k_proj.weight"].grad[:32], g["g_kproj"], **TOL)
    _close(sd["table_encoder.fc.weight"].grad[:16], g["g_table_fc"], **TOL)
    _close(sd["bart_model.model.shared.weight"].grad[:64], g["g_shared"], **TOL)
    _close(sd["img_encoder.linear.weight"].grad[:16], g["g_img_lin"], **TOL)
    _close(sd["bart_model.model.encoder.layers.0.self_attn.q_proj.weight"].grad[:16], g["g_enc_q"], **TOL)
    # stage 1/2 of the ResNet are detached: no gradient may reach them
    assert sd["img_encoder.resnet.layer2.0.conv1.weight"].grad is None
    assert sd["img_encoder.resnet.layer3.0.conv1.weight"].grad is not None


def test_c1_text_step(golden_dir):
    g = _load(golden_dir, "c1_textstep.npz")
    cfg = tiny_cfg(vocab=150, d=64, ffn=128, layers=2, heads=4, maxpos=80)
    sd = formula_state_dict(bo.bart_param_shapes(cfg, False, prefix="bart_model."), std=0.08)
    for v in sd.values():
        v.requires_grad_(True)
    b = syn.yelp_batch(2, 2, 64, 1, cfg.vocab_size, seed=int(g["seed"]), img_hw=8)
    loss = so.text_step_loss(sd, cfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], None, training=True)
    _close(loss, g["loss"], 1e-5, 1e-6)
    loss.backward()
    _close(sd["bart_model.model.shared.weight"].grad[:32], g["g_shared"], **TOL)
    _close(sd["bart_model.model.decoder.rating_embeddings"].grad, g["g_rating"], **TOL)
    _close(sd["bart_model.model.decoder.layers.1.encoder_attn.k_proj.weight"].grad, g["g_dec_k"], **TOL)


def test_f4_optimizer(golden_dir):
    g = _load(golden_dir, "f4_optim.npz")
    names = ["fc.weight", "fc.bias", "layer_norm.weight", "layer_norm.bias"]
    shapes = [(5, 6), (5,), (5,), (5,)]
    params = [formula_tensor("f4." + n, s, 0.5, 1.0 if n.endswith("layer_norm.weight") else 0.0).requires_grad_(True)
              for n, s in zip(names, shapes)]
    groups = so.q1_param_groups(zip(names, params))
    assert len(groups[0]["params"]) == int(g["n_group0"]) and len(groups[1]["params"]) == int(g["n_group1"]) == 0
    state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in groups[0]["params"]}
    for step in range(4):
        x = formula_tensor("f4.x%d" % step, (7, 6), 1.0)
        y = torch.nn.functional.layer_norm(torch.nn.functional.linear(x, params[0], params[1]), (5,), params[2],
                                           params[3]).pow(2).sum()
        # optimizer.zero_grad() (multimodal_train.py:359) only touches the optimiser's own params: with
        # Q1 the no-decay params are in no group, so their .grad ACCUMULATES across steps and keeps
        # inflating the clip norm (quirk Q1b, found while pinning this fixture).
        for p in groups[0]["params"]:
            p.grad = None
        y.backward()
        so.clip_grad_norm([p.grad for p in params], 1.0)
        lr = 1e-2 * so.linear_schedule_lambda(step, 2, 6)
        with torch.no_grad():
            for p in groups[0]["params"]:
                m, v = state[id(p)]
                so.adamw_step(p, p.grad, m, v, step + 1, lr, weight_decay=0.01)
        _close(torch.cat([p.detach().flatten() for p in params]), g["params"][step], 1e-5, 1e-6)


def test_amazon_table_encoder(golden_dir):
    """AmazonTableEncoder restatement against the reference run (133 positions, nested category means, masks, grads)."""
    g = _load(golden_dir, "table_amazon.npz")
    sd = formula_state_dict(eo.amazon_table_param_shapes(), std=0.02)
    for v in sd.values():
        v.requires_grad_(True)
    emb = formula_tensor("bart_model.model.shared.weight", (200, 1024), 0.02)
    fv = [g["price"], g["rating"], g["brand"], g["name"], g["category"], g["description"]]
    field, fv2 = syn.amazon_table_batch(3, 200, seed=22)
    assert torch.equal(field, g["field"]) and all(torch.equal(a, b) for a, b in zip(fv, fv2))
    h, m = eo.amazon_table_encoder(sd, emb, g["field"], fv)
    assert torch.equal(m, g["mask"]) and h.shape == (3, 133, 1024)
    _close(h[:, :12], g["hiddens"], **TOL)
    _close(h[:, -3:], g["hiddens_tail"], **TOL)
    _close(h.double().abs().sum(), g["hiddens_checksum"], 1e-5, 1e-3)
    h.backward(formula_tensor("table.gout.amazon", h.shape, std=1.0))
    _close(sd["table_encoder.price_embedding.weight"].grad, g["g_price"], **TOL)
    _close(sd["table_encoder.rating_embedding.weight"].grad, g["g_rating"], **TOL)
    _close(sd["table_encoder.fc.weight"].grad[:64], g["g_fc_w"], **TOL)
    _close(sd["table_encoder.fc.bias"].grad, g["g_fc_b"], **TOL)
    _close(sd["table_encoder.linear.weight"].grad[:64], g["g_linear"], **TOL)


def p2_setup():
    """Inputs and weights of oracle/make_golden_r5.py's p2 fixture, rebuilt from the same closed forms / seeds."""
    cfg = bo.BartCfg(vocab_size=200, d_model=1024, ffn_dim=64, encoder_layers=1, decoder_layers=1, heads=16, max_position_embeddings=32, dropout=0.0)
    labels = syn.token_batch(2, 12, cfg.vocab_size, seed=5, min_len=4)
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn(2, 2, 3, 224, 224, generator=g)
    imask = torch.tensor([[True, True], [True, False]])
    imgs = imgs * imask[:, :, None, None, None].float()
    tlabels = syn.token_batch(3, 12, cfg.vocab_size, seed=6, min_len=4)
    field, fv = syn.table_batch(3, cfg.vocab_size, seed=9)
    shapes = bo.bart_param_shapes(cfg, False, prefix="bart_model.")
    shapes.update(eo.table_param_shapes())
    sd = formula_state_dict(shapes, std=0.02)
    sd.update(formula_state_dict(eo.resnet_param_shapes(cfg.d_model), std=0.05))
    return cfg, sd, labels, imgs, imask, tlabels, field, fv


def test_p2_pretrain_wrappers(golden_dir):
    """The reference's own step-2 wrapper classes (img_pretrain.ImgSupervised with the stand-in backbone, table_pretrain.TableSupervised:
    oracle/make_golden_r5.py) against the oracle composition the HIP wrappers are tested with: encoder -> unimodal decoder branch with a
    zero rating difference -> label-smoothing loss (img_pretrain.py:85-141, table_pretrain.py:84-129)."""
    g = _load(golden_dir, "p2_pretrain_wrappers.npz")
    cfg, sd, labels, imgs, imask, tlabels, field, fv = p2_setup()
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() > 0 and "running" not in k:
            v.requires_grad_(True)
    D = "bart_model.model.decoder.layers.0."
    ih = eo.resnet101_features(sd, imgs.reshape(-1, 3, 224, 224), training=True).reshape(2, 2, -1, cfg.d_model)
    lg = bo.enc_forward(sd, cfg, ih, torch.zeros(2, 1), imask.unsqueeze(-1).repeat(1, 1, ih.shape[2]), labels, training=True, prefix="bart_model.")
    li = bo.label_smoothing_loss(lg.view(-1, cfg.vocab_size), labels.view(-1), cfg.vocab_size, 0.1)
    li.backward()
    _close(li.detach(), g["img_loss"], 1e-5, 1e-6)
    # the stand-in backbone IS this restatement, so the ResNet gradients agree to rounding; they pin the wrapper's plumbing (mask, reshape)
    _close(sd["img_encoder.linear.weight"].grad[:16], g["img_g_linear"], **TOL)
    _close(sd["img_encoder.resnet.layer3.22.conv3.weight"].grad[:16, :, 0, 0], g["img_g_l3_22_conv3"], **TOL)
    _close(sd["img_encoder.resnet.layer3.22.bn3.weight"].grad, g["img_g_l3_22_bn3_w"], **TOL)
    _close(sd[D + "encoder_attn.k_proj.weight"].grad[:32], g["img_g_kproj"], **TOL)
    _close(sd["bart_model.model.shared.weight"].grad[:64], g["img_g_shared"], **TOL)
    _close(sd[D + "fc1.weight"].grad[:16], g["img_g_fc1"], **TOL)
    for v in sd.values():
        v.grad = None
    th, tmask = eo.yelp_table_encoder(sd, sd["bart_model.model.shared.weight"], field, fv)
    lt_ = bo.enc_forward(sd, cfg, th.unsqueeze(1), torch.zeros(3, 1), tmask.unsqueeze(1), tlabels, training=True, prefix="bart_model.")
    lt = bo.label_smoothing_loss(lt_.view(-1, cfg.vocab_size), tlabels.view(-1), cfg.vocab_size, 0.1)
    lt.backward()
    _close(lt.detach(), g["tab_loss"], 1e-5, 1e-6)
    _close(sd["table_encoder.fc.weight"].grad[:16], g["tab_g_fc"], **TOL)
    _close(sd["table_encoder.fc.bias"].grad, g["tab_g_fc_b"], **TOL)
    _close(sd["table_encoder.linear.weight"].grad[:16], g["tab_g_linear"], **TOL)
    _close(sd["table_encoder.rating_embedding.weight"].grad, g["tab_g_rating"], **TOL)
    _close(sd["table_encoder.hours_embedding.weight"].grad, g["tab_g_hours"], **TOL)
    _close(sd[D + "encoder_attn.k_proj.weight"].grad[:32], g["tab_g_kproj"], **TOL)
    _close(sd["bart_model.model.shared.weight"].grad[:64], g["tab_g_shared"], **TOL)


def test_g2_generate_full_size_short(golden_dir):
    """The oracle's beam search on the full-size generation fixture (cfg/bart-large.json, 12 + 12 layers, BASELINE config 5's memory for two
    businesses; oracle/make_golden_r5.py ran the REFERENCE's generate() on it) at max_length 32: ids equal to the reference's.  The
    max_length 128 ids of the same fixture were compared with the oracle's when the fixture was made (`oracle_ids`, minutes of CPU) and are
    what the GPU test holds the HIP search to."""
    from multimodalsum_amd.config import BartConfig
    from oracle import generate_oracle as go
    from oracle.gen_fixture import G2, g2_inputs, g2_kwargs
    g = _load(golden_dir, "g2_generate_full.npz")
    assert torch.equal(g["ids"], g["oracle_ids"]) and g["ids"].shape[0] == G2["B"] and g["ids"].shape[1] <= G2["max_length"]
    cfg = BartConfig.from_json_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg", "bart-large.json"))
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=cfg.encoder_layers,
                      decoder_layers=cfg.decoder_layers, heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.0)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=G2["std"])
    text_h, text_m, table_h, table_m, img_h, img_m = g2_inputs(cfg, int(g["seed"]))
    with torch.no_grad():
        out = go.beam_search(sd, ocfg, [text_h, table_h, img_h], [text_m, table_m, img_m], torch.zeros(G2["B"], 1), True,
                             decoder_start_token_id=cfg.decoder_start_token_id, **g2_kwargs(G2["short_length"]))
    assert torch.equal(out, g["ids_short"]), (out[:, :12], g["ids_short"][:, :12])

