#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE (development container only).

    python oracle/make_golden.py            # writes tests/golden/*.npz

Imports /root/reference/src with the three shims of SURVEY.md Appendix C (apex, torchvision,
transformers.AdamW), fills every model with the RNG-free formula weights of
`multimodalsum_amd.formula_init`, runs the fixtures F1..F7 of SURVEY.md section 8c and stores
inputs + expected outputs as small .npz files.  Nothing from /root/reference is copied: a fixture
holds data (inputs, expected outputs) only, and both the oracle and the HIP path regenerate the
weights from the same closed form.

The ResNet101 backbone cannot be imported (torchvision absent): fixtures that need image features
feed the reference a stand-in `img_encoder` returning features computed by the oracle restatement,
so they pin the orchestration (masks, beta gate, leave-one-out) but not the backbone.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/src"

from multimodalsum_amd.formula_init import formula_tensor, formula_state_dict  # noqa: E402
from oracle import encoders_oracle as eo  # noqa: E402
from multimodalsum_amd import synthetic as syn  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    import transformers  # noqa: F401  (must precede the torchvision stub)
    from transformer.optimization import AdamW as _A
    transformers.AdamW = _A
    apex = types.ModuleType("apex")
    ap = types.ModuleType("apex.parallel")
    ap.DistributedDataParallel = lambda m, **k: m
    apex.parallel = ap
    sys.modules["apex"] = apex
    sys.modules["apex.parallel"] = ap
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvm = types.ModuleType("torchvision.models")
    for n in ["Compose", "RandomResizedCrop", "RandomRotation", "RandomHorizontalFlip", "ColorJitter",
              "ToTensor", "Normalize", "Resize", "CenterCrop"]:
        setattr(tvt, n, lambda *a, **k: None)
    tv.transforms, tv.models = tvt, tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.models": tvm})
    import multimodal_train as mt
    import text_pretrain as tp
    from transformer import modeling_multimodalsum as mm
    from transformer.configuration_bart import BartConfig
    import table_encoder as te
    import utils as ru
    return mt, tp, mm, BartConfig, te, ru, _A


def canonical(name):
    """Aliased parameters take the name of the tensor they alias."""
    for alias in ("model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight"):
        if name.endswith(alias):
            return name[: -len(alias)] + "model.shared.weight"
    if name.endswith("bart_embedding.weight"):
        return "bart_model.model.shared.weight"
    return name


def load_formula(module, prefix="", std=0.02):
    """Fill an nn.Module with formula weights keyed by (prefix + its own state_dict names)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        name = canonical(prefix + k)
        new[k] = formula_state_dict({name: tuple(v.shape)}, std=std)[name]
    module.load_state_dict(new)
    return module


def npz(path, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def tiny_cfg(BartConfig, vocab=100, d=64, ffn=128, layers=2, heads=4, maxpos=64):
    return BartConfig(vocab_size=vocab, d_model=d, encoder_ffn_dim=ffn, decoder_ffn_dim=ffn,
                      encoder_layers=layers, decoder_layers=layers, encoder_attention_heads=heads,
                      decoder_attention_heads=heads, max_position_embeddings=maxpos, dropout=0.0,
                      attention_dropout=0.0, activation_dropout=0.0, activation_function="gelu",
                      normalize_embedding=True, scale_embedding=False, static_position_embeddings=False,
                      pad_token_id=1, bos_token_id=0, eos_token_id=2, extra_pos_embeddings=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--full", action="store_true", help="also run the full-size BART-large spot check (minutes)")
    ap.add_argument("--only-full", action="store_true", help="run only the full-size fixtures F8 / F8b")
    ap.add_argument("--only-contract", action="store_true", help="run only the state_dict key/shape contract fixture")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    mt, tp, mm, BartConfig, te, ru, RefAdamW = import_reference()
    G = lambda n: os.path.join(args.out, n)  # noqa: E731
    if args.only_full:
        if not os.path.exists(G("f8_fullsize.npz")):
            full_size_spot_check(mm, BartConfig, ru, G)
        full_size_step(mt, mm, te, BartConfig, G)
        return
    state_dict_contract(mt, tp, mm, te, BartConfig, RefAdamW, G)
    if args.only_contract:
        return

    # ---- F7: shift_tokens_right / padding mask helpers ------------------------------------
    cases = torch.tensor([
        [5, 6, 7, 8, 9, 10],      # no eos, no pad
        [5, 6, 7, 2, 1, 1],       # eos then pads
        [5, 6, 7, 8, 9, 2],       # eos last
        [5, 2, 1, 1, 1, 1],
    ])
    cases_bos = cases.clone()
    cases_bos[:, 0] = 0
    outs = [mm.shift_tokens_right(c, 1, 0, 2) for c in (cases, cases_bos)]
    npz(G("f7_shift.npz"), in_a=cases, out_a=outs[0], in_b=cases_bos, out_b=outs[1])

    # ---- F5: label-smoothing loss ------------------------------------------------------------
    V = 50265
    logits = formula_tensor("f5.logits", (8, V), std=2.0)
    target = torch.tensor([3, 1, 50264, 0, 1, 777, 2, 1])
    lg = logits.clone().requires_grad_(True)
    loss = ru.LabelSmoothingLoss(V, smoothing=0.1)(lg, target)
    loss.backward()
    npz(G("f5_loss.npz"), target=target, loss=loss, grad_checksum=lg.grad.double().abs().sum(),
        grad_sample=lg.grad[:, :64], logits_seed_std=np.float32(2.0))

    # ---- F1: multimodal cross-attention unit -------------------------------------------------
    D, H, T, B = 64, 4, 5, 3
    att = mm.SelfAttention(D, H, encoder_decoder_attention=True, multimodal=True)
    load_formula(att, prefix="f1.", std=0.15)
    q = formula_tensor("f1.query", (T, B, D), std=1.0).requires_grad_(True)
    ktext = formula_tensor("f1.ktext", (7, 3, B, D), std=1.0).requires_grad_(True)
    ktab = formula_tensor("f1.ktab", (6, 1, B, D), std=1.0).requires_grad_(True)
    kimg = formula_tensor("f1.kimg", (4, 2, B, D), std=1.0).requires_grad_(True)
    ptext = torch.zeros(B, 3, 7, dtype=torch.bool)
    ptext[0, 1, :] = True          # null entity
    ptext[1, 0, 4:] = True         # partially padded
    ptext[2, 2, 6:] = True
    ptab = torch.zeros(B, 1, 6, dtype=torch.bool)
    ptab[1, 0, :] = True           # business without table
    ptab[0, 0, 3:] = True
    pimg = torch.zeros(B, 2, 4, dtype=torch.bool)
    pimg[2, :, :] = True           # business without images
    pimg[0, 1, :] = True           # one missing image
    out, _ = att(q, [ktext, ktab, kimg], [ptext, ptab, pimg])
    gout = formula_tensor("f1.gout", out.shape, std=1.0)
    out.backward(gout)
    grads = {("g_" + n.replace(".", "_")): p.grad for n, p in att.named_parameters()}
    npz(G("f1_crossattn.npz"), ptext=ptext, ptab=ptab, pimg=pimg, out=out, gq=q.grad, gktext=ktext.grad,
        gktab=ktab.grad, gkimg=kimg.grad, **grads)

    # ---- F2: tiny multi-encoder decoder pass + tiny encoder ---------------------------------
    cfg = tiny_cfg(BartConfig)
    model = mm.BartForMultiEncConditionalGeneration(cfg)
    load_formula(model, prefix="f2.", std=0.08)
    model.train()  # dropout=0 in cfg; exercises the training path
    Bz, N, S, Tt = 3, 3, 8, 10
    ids = syn.token_batch(Bz * N, S, cfg.vocab_size, seed=11, min_len=3).view(Bz, N, S)
    labels = syn.token_batch(Bz, Tt, cfg.vocab_size, seed=12, min_len=4)
    labels[0] = torch.tensor([7, 8, 9, 10, 11, 12, 13, 14, 15, 16])  # row 0 full length (no pad)
    enc_out = model.model.encoder(input_ids=ids.view(-1, S), attention_mask=ids.view(-1, S).ne(1))[0]
    text_h = enc_out.view(Bz, N, S, -1)
    text_m = ids.ne(1).clone()
    text_m[1, 2, :] = False  # a null review entity
    table_h = formula_tensor("f2.table_h", (Bz, 1, 6, cfg.d_model), std=1.0).requires_grad_(True)
    table_m = torch.ones(Bz, 1, 6, dtype=torch.bool)
    table_m[2] = False
    img_h = formula_tensor("f2.img_h", (Bz, 2, 4, cfg.d_model), std=1.0).requires_grad_(True)
    img_m = torch.ones(Bz, 2, 4, dtype=torch.bool)
    img_m[0] = False
    img_m[1, 1] = False
    rating_diff = torch.tensor([[0.5], [-1.25], [2.0]])
    logits = model(text_h, text_m, table_h, table_m, img_h, img_m, rating_diff=rating_diff, labels=labels)[0]
    lossf = ru.LabelSmoothingLoss(cfg.vocab_size, smoothing=0.1)
    loss = lossf(logits.view(-1, cfg.vocab_size), labels.view(-1))
    loss.backward()
    sel = ["model.shared.weight", "model.decoder.rating_embeddings",
           "model.decoder.layers.0.encoder_attn.alpha_proj.weight",
           "model.decoder.layers.1.encoder_attn.k_proj.weight",
           "model.decoder.layers.1.encoder_attn.k_proj.bias",
           "model.decoder.layers.0.self_attn.q_proj.weight",
           "model.decoder.layers.1.fc1.weight", "model.decoder.layers.0.final_layer_norm.weight",
           "model.encoder.layers.0.self_attn.v_proj.weight", "model.encoder.layers.1.fc2.bias",
           "model.encoder.embed_positions.weight", "model.decoder.embed_positions.weight",
           "model.encoder.layernorm_embedding.weight"]
    named = dict(model.named_parameters())
    gsel = {"g_" + n.replace(".", "_"): named[n].grad for n in sel}
    npz(G("f2_decoder.npz"), ids=ids, labels=labels, text_m=text_m, table_m=table_m, img_m=img_m,
        rating_diff=rating_diff, enc_out=enc_out, logits=logits, loss=loss, g_table_h=table_h.grad,
        g_img_h=img_h.grad, **gsel)

    # ---- G1: beam-search generation with the same tiny model (test.py:153-158 call shape) -----------------------
    model.eval()
    with torch.no_grad():
        enc_eval = model.model.encoder(input_ids=ids.view(-1, S), attention_mask=ids.view(-1, S).ne(1))[0].view(Bz, N, S, -1)
        zeros = torch.zeros(Bz, 1)
        gen_a = model.generate(enc_eval, text_m, table_h.detach(), table_m, img_h.detach(), img_m, rating_diff=zeros, num_beams=4,
                               length_penalty=1.0, max_length=14, no_repeat_ngram_size=3, early_stopping=True)
        gen_b = model.generate(enc_eval, text_m, table_h.detach(), table_m, img_h.detach(), img_m, rating_diff=rating_diff, num_beams=2,
                               length_penalty=2.0, max_length=10, min_length=4, no_repeat_ngram_size=2, early_stopping=False)
    npz(G("g1_beam.npz"), ids=ids, text_m=text_m, table_m=table_m, img_m=img_m, rating_diff=rating_diff, enc_eval=enc_eval,
        gen_a=gen_a, gen_b=gen_b)
    model.train()

    # text-only variant (BartForEncConditionalGeneration) on the same weights ------------------
    tmodel = mm.BartForEncConditionalGeneration(cfg)
    load_formula(tmodel, prefix="f2.", std=0.08)
    tmodel.train()
    enc_out_t = tmodel.model.encoder(input_ids=ids.view(-1, S), attention_mask=ids.view(-1, S).ne(1))[0]
    tl = tmodel(enc_out_t.view(Bz, N, S, -1), rating_diff, text_m, labels=labels)[0]
    npz(G("f2_textonly.npz"), logits=tl)

    # ---- Table encoder -----------------------------------------------------------------------
    emb = torch.nn.Embedding(200, 1024, padding_idx=1)
    emb.weight.data.copy_(formula_tensor("bart_model.model.shared.weight", (200, 1024), 0.02))
    tenc = te.YelpTableEncoder(emb)
    load_formula(tenc, prefix="table_encoder.", std=0.02)
    emb.weight.data.copy_(formula_tensor("bart_model.model.shared.weight", (200, 1024), 0.02))
    field, fv = syn.table_batch(3, 200, seed=21)
    th, tm = tenc(field, fv)
    gt = formula_tensor("table.gout", th.shape, std=1.0)
    th.backward(gt)
    npz(G("table_yelp.npz"), field=field, name=fv[0], category=fv[1], str_cat=fv[2], str_bool=fv[3],
        rating=fv[4], hours=fv[5], hiddens=th, mask=tm,
        g_rating=tenc.rating_embedding.weight.grad, g_hours=tenc.hours_embedding.weight.grad,
        g_fc_w=tenc.fc.weight.grad[:64], g_fc_b=tenc.fc.bias.grad, g_linear=tenc.linear.weight.grad[:64],
        emb_grad_is_none=np.bool_(emb.weight.grad is None))

    # ---- Amazon table encoder (table_encoder.py:86-167) --------------------------------------------------
    aenc = te.AmazonTableEncoder(emb)
    load_formula(aenc, prefix="table_encoder.", std=0.02)
    emb.weight.data.copy_(formula_tensor("bart_model.model.shared.weight", (200, 1024), 0.02))
    afield, afv = syn.amazon_table_batch(3, 200, seed=22)
    ah, am = aenc(afield, afv)
    ah.backward(formula_tensor("table.gout.amazon", ah.shape, std=1.0))
    npz(G("table_amazon.npz"), field=afield, price=afv[0], rating=afv[1], brand=afv[2], name=afv[3], category=afv[4],
        description=afv[5], hiddens=ah[:, :12], hiddens_tail=ah[:, -3:], hiddens_checksum=ah.double().abs().sum(), mask=am,
        g_price=aenc.price_embedding.weight.grad, g_rating=aenc.rating_embedding.weight.grad,
        g_fc_w=aenc.fc.weight.grad[:64], g_fc_b=aenc.fc.bias.grad, g_linear=aenc.linear.weight.grad[:64])

    # ---- F3: leave-one-out multimodal step (MultimodalSum.forward) ---------------------------
    cfg3 = tiny_cfg(BartConfig, vocab=200, d=1024, ffn=64, layers=1, heads=16, maxpos=32)
    ms = mt.MultimodalSum.__new__(mt.MultimodalSum)
    torch.nn.Module.__init__(ms)
    ms.bart_model = mm.BartForMultiEncConditionalGeneration(cfg3)
    load_formula(ms.bart_model, prefix="bart_model.", std=0.02)
    ms.table_encoder = te.YelpTableEncoder(ms.bart_model.model.shared)
    for n, p in ms.table_encoder.named_parameters():
        if not n.startswith("bart_embedding"):
            p.data.copy_(formula_tensor("table_encoder." + n, p.shape, 0.02))
    B3, NR3, S3, I3 = 2, 3, 16, 2
    batch = syn.yelp_batch(B3, NR3, S3, I3, cfg3.vocab_size, seed=31, img_hw=224)
    rs_shapes = eo.resnet_param_shapes(1024)
    rs_sd = formula_state_dict(rs_shapes, std=0.05)

    class StandInImg(torch.nn.Module):
        """Oracle restatement standing in for the un-importable torchvision backbone."""

        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Parameter(rs_sd["img_encoder.linear.weight"].clone())

        def forward(self, x):
            sd = dict(rs_sd)
            sd["img_encoder.linear.weight"] = self.lin
            return eo.resnet101_features(sd, x, training=True)

    ms.img_encoder = StandInImg()
    mt.args = argparse.Namespace(label_smoothing=0.1)
    ms.train()
    field3, fv3 = batch["field"], batch["field_value"]
    loss3 = ms(batch["reviews"], batch["reviews_mask"], batch["reviews_rating"], field3, fv3, batch["img"],
               batch["img_mask"])[0]
    loss3.backward()
    named3 = dict(ms.named_parameters())
    npz(G("f3_step.npz"), seed=np.int64(31), B=np.int64(B3), NR=np.int64(NR3), S=np.int64(S3), I=np.int64(I3),
        img_hw=np.int64(224), loss=loss3,
        g_rating=named3["bart_model.model.decoder.rating_embeddings"].grad,
        g_alpha=named3["bart_model.model.decoder.layers.0.encoder_attn.alpha_proj.weight"].grad[:32],
        g_beta_b=named3["bart_model.model.decoder.layers.0.encoder_attn.beta_proj.bias"].grad,
        g_kproj=named3["bart_model.model.decoder.layers.0.encoder_attn.k_proj.weight"].grad[:32],
        g_table_fc=named3["table_encoder.fc.weight"].grad[:16],
        g_shared=named3["bart_model.model.shared.weight"].grad[:64],
        g_img_lin=named3["img_encoder.lin"].grad[:16],
        g_enc_q=named3["bart_model.model.encoder.layers.0.self_attn.q_proj.weight"].grad[:16])

    # text-only step (TextSupervised.forward counterpart, BASELINE config 1 shape) --------------
    tsm = tp.TextSupervised.__new__(tp.TextSupervised)
    torch.nn.Module.__init__(tsm)
    cfg1 = tiny_cfg(BartConfig, vocab=150, d=64, ffn=128, layers=2, heads=4, maxpos=80)
    tsm.bart_model = mm.BartForEncConditionalGeneration(cfg1)
    load_formula(tsm.bart_model, prefix="bart_model.", std=0.08)
    tp.args = argparse.Namespace(label_smoothing=None)
    tsm.train()
    tb = syn.yelp_batch(2, 2, 64, 1, cfg1.vocab_size, seed=41, img_hw=8)
    tl1 = tsm(tb["reviews"], tb["reviews_mask"], tb["reviews_rating"])[0]
    tl1.backward()
    nt = dict(tsm.named_parameters())
    npz(G("c1_textstep.npz"), seed=np.int64(41), loss=tl1,
        g_shared=nt["bart_model.model.shared.weight"].grad[:32],
        g_rating=nt["bart_model.model.decoder.rating_embeddings"].grad,
        g_dec_k=nt["bart_model.model.decoder.layers.1.encoder_attn.k_proj.weight"].grad)

    # ---- F4: optimiser (HF AdamW + linear warm-up, Q1 grouping) -------------------------------
    toy = torch.nn.ModuleDict({"fc": torch.nn.Linear(6, 5), "layer_norm": torch.nn.LayerNorm(5)})
    for n, p in toy.named_parameters():
        p.data.copy_(formula_tensor("f4." + n, p.shape, 0.5, 1.0 if n.endswith("layer_norm.weight") else 0.0))
    import train_utils as tu
    no_decay = ['bias', 'bn1.weight', 'bn2.weight', 'bn3.weight', 'layer_norm.weight', 'layernorm_embedding.weight']
    opt = tu.get_optimizer(1e-2, no_decay, toy.named_parameters(), None)
    from transformer.optimization import get_linear_schedule_with_warmup
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=6)
    hist = []
    for step in range(4):
        x = formula_tensor("f4.x%d" % step, (7, 6), 1.0)
        y = toy["layer_norm"](toy["fc"](x)).pow(2).sum()
        opt.zero_grad()
        y.backward()
        torch.nn.utils.clip_grad_norm_(toy.parameters(), 1.0)
        opt.step()
        sch.step()
        hist.append(torch.cat([p.detach().flatten() for p in toy.parameters()]))
    npz(G("f4_optim.npz"), params=torch.stack(hist), n_group0=np.int64(len(opt.param_groups[0]["params"])),
        n_group1=np.int64(len(opt.param_groups[1]["params"])))

    if args.full:
        full_size_spot_check(mm, BartConfig, ru, G)
        full_size_step(mt, mm, te, BartConfig, G)


def full_size_spot_check(mm, BartConfig, ru, G):
    """F8: BART-large shape, formula weights, B=1, one leave-one-out pass -> 64 logits + checksum."""
    cfg = BartConfig.from_json_file("/root/reference/cfg/bart-large.json")
    cfg.dropout = 0.0
    model = mm.BartForMultiEncConditionalGeneration(cfg)
    load_formula(model, prefix="bart_model.", std=0.02)
    model.eval()
    b = syn.yelp_batch(1, 9, 128, 4, cfg.vocab_size, seed=1234, img_hw=8)
    with torch.no_grad():
        enc = model.model.encoder(input_ids=b["reviews"].view(-1, 128), attention_mask=b["reviews_mask"].view(-1, 128))[0]
        text_h = enc.view(1, 9, 128, -1)
        table_h = formula_tensor("f8.table_h", (1, 1, 47, 1024), 1.0)
        img_h = formula_tensor("f8.img_h", (1, 4, 196, 1024), 1.0)
        table_m = torch.ones(1, 1, 47, dtype=torch.bool)
        img_m = torch.ones(1, 4, 196, dtype=torch.bool)
        img_m[0, 3] = False
        others = list(range(1, 9))
        rd = (b["reviews_rating"][:, 0] - b["reviews_rating"][:, others].mean(dim=1)).unsqueeze(1)
        logits = model(text_h[:, others], b["reviews_mask"][:, others], table_h, table_m, img_h, img_m,
                       rating_diff=rd, labels=b["reviews"][:, 0])[0]
        loss = ru.LabelSmoothingLoss(cfg.vocab_size, 0.1)(logits.view(-1, cfg.vocab_size), b["reviews"][:, 0].view(-1))
    npz(G("f8_fullsize.npz"), seed=np.int64(1234), enc_sample=enc[:, :4, :32], enc_abs_sum=enc.double().abs().sum(),
        logits_sample=logits[0, :8, :64], logits_abs_sum=logits.double().abs().sum(), loss=loss, rating_diff=rd)


def state_dict_contract(mt, tp, mm, te, BartConfig, RefAdamW, G):
    """The checkpoint contract: state_dict keys and shapes of the reference's own modules at a small
    config (key names do not depend on the sizes), and the layout of optimizer.state_dict() that train_utils.py:97 saves.
    img_encoder.* is absent: torchvision cannot be imported here (its keys are pinned by the published resnet101 layout
    only, like the backbone's arithmetic)."""
    import json
    cfg = tiny_cfg(BartConfig, vocab=200, d=1024, ffn=64, layers=2, heads=16, maxpos=32)
    out = {}
    ms = mt.MultimodalSum.__new__(mt.MultimodalSum)
    torch.nn.Module.__init__(ms)
    ms.bart_model = mm.BartForMultiEncConditionalGeneration(cfg)
    ms.table_encoder = te.YelpTableEncoder(ms.bart_model.model.shared)
    out["multimodal_yelp"] = [[k, list(v.shape)] for k, v in ms.state_dict().items()]
    ms.table_encoder = te.AmazonTableEncoder(ms.bart_model.model.shared)
    out["multimodal_amazon"] = [[k, list(v.shape)] for k, v in ms.state_dict().items()]
    ts = tp.TextSupervised.__new__(tp.TextSupervised)
    torch.nn.Module.__init__(ts)
    ts.bart_model = mm.BartForEncConditionalGeneration(cfg)
    out["text"] = [[k, list(v.shape)] for k, v in ts.state_dict().items()]
    ms.table_encoder = te.YelpTableEncoder(ms.bart_model.model.shared)
    out["named_parameters_multimodal_yelp"] = [n for n, _ in ms.named_parameters()]
    # optimizer.state_dict() layout after one step (HF AdamW, optimization.py:225-232)
    lin = torch.nn.Linear(3, 2)
    opt = RefAdamW([{"params": [lin.weight], "weight_decay": 0.01}, {"params": [lin.bias], "weight_decay": 0.0}], lr=1e-3)
    lin(torch.ones(1, 3)).sum().backward()
    opt.step()
    osd = opt.state_dict()
    out["optimizer"] = {"top": sorted(osd), "state_entry": sorted(osd["state"][0]), "group_keys": sorted(osd["param_groups"][0]),
                        "step_after_one": int(osd["state"][0]["step"]), "exp_avg_shape": list(osd["state"][0]["exp_avg"].shape)}
    with open(G("state_dict_contract.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", G("state_dict_contract.json"))


F8_GRADS = [
    "bart_model.model.shared.weight",
    "bart_model.model.decoder.rating_embeddings",
    "bart_model.model.encoder.embed_positions.weight",
    "bart_model.model.encoder.layers.0.self_attn.q_proj.weight",
    "bart_model.model.encoder.layers.5.fc1.weight",
    "bart_model.model.encoder.layers.11.fc2.weight",
    "bart_model.model.encoder.layers.11.final_layer_norm.weight",
    "bart_model.model.decoder.layers.0.self_attn.v_proj.weight",
    "bart_model.model.decoder.layers.0.encoder_attn.k_proj.weight",
    "bart_model.model.decoder.layers.3.encoder_attn.alpha_proj.weight",
    "bart_model.model.decoder.layers.6.encoder_attn.beta_proj.bias",
    "bart_model.model.decoder.layers.6.encoder_attn.out_proj.weight",
    "bart_model.model.decoder.layers.11.fc1.weight",
    "bart_model.model.decoder.layers.11.fc2.bias",
    "bart_model.model.decoder.layers.11.encoder_attn_layer_norm.weight",
    "table_encoder.fc.weight",
    "table_encoder.rating_embedding.weight",
]


def full_size_step(mt, mm, te, BartConfig, G):
    """F8b: the reference's MultimodalSum.forward + backward (multimodal_train.py:124-163) at cfg/bart-large.json,
    B=1, 9 reviews x 128 tokens, 4 images of 224x224 (backbone = the oracle restatement, as in F3), formula weights,
    dropout 0, train mode.  Stores the loss, and for a spread of parameters a gradient slice + the gradient's L1 norm --
    from the reference run as it is (fp32) AND from the same modules converted to fp64 (`module.double()`): the fp64 run
    is the exact value, the difference between the two is the reference's own fp32 rounding error, tensor by tensor, which
    is what a 1e-3 comparison has to be read against (some gradients -- rating_embeddings: the nine leave-one-out rating
    differences of a business sum to zero -- are sums of cancelling terms and are not reproducible to 1e-3 in fp32 by any
    implementation, the reference included)."""
    import time
    cfg = BartConfig.from_json_file("/root/reference/cfg/bart-large.json")
    cfg.dropout = 0.0
    ms = mt.MultimodalSum.__new__(mt.MultimodalSum)
    torch.nn.Module.__init__(ms)
    ms.bart_model = mm.BartForMultiEncConditionalGeneration(cfg)
    load_formula(ms.bart_model, prefix="bart_model.", std=0.02)
    ms.table_encoder = te.YelpTableEncoder(ms.bart_model.model.shared)
    for n, p in ms.table_encoder.named_parameters():
        if not n.startswith("bart_embedding"):
            p.data.copy_(formula_tensor("table_encoder." + n, p.shape, 0.02))
    rs_sd = formula_state_dict(eo.resnet_param_shapes(1024), std=0.05)

    class StandInImg(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Parameter(rs_sd["img_encoder.linear.weight"].clone())

        def forward(self, x):
            sd = {k: (v.to(x.dtype) if v.is_floating_point() else v) for k, v in rs_sd.items()}
            sd["img_encoder.linear.weight"] = self.lin
            return eo.resnet101_features(sd, x, training=True)

    ms.img_encoder = StandInImg()
    mt.args = argparse.Namespace(label_smoothing=0.1)
    ms.train()
    b = syn.yelp_batch(1, 9, 128, 4, cfg.vocab_size, seed=1234, img_hw=224)
    if not bool(b["img_mask"].any()):
        b["img_mask"][0, 0] = True            # keep the image branch live in this fixture
    out = {"seed": np.int64(1234), "img_mask": b["img_mask"]}
    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        if dt == torch.float64:
            ms.double()

            class _CastLinear(torch.nn.Linear):      # table_encoder.py:62,65 feed `x.float()` to these two: let them accept it in fp64
                def forward(self, x):
                    return torch.nn.functional.linear(x.to(self.weight.dtype), self.weight, self.bias)
            ms.table_encoder.rating_embedding.__class__ = _CastLinear
            ms.table_encoder.hours_embedding.__class__ = _CastLinear
        for p in ms.parameters():
            p.grad = None
        t0 = time.time()
        loss = ms(b["reviews"], b["reviews_mask"], b["reviews_rating"].to(dt), b["field"], b["field_value"], b["img"].to(dt), b["img_mask"])[0]
        loss.backward()
        print("full-size reference step (%s): %.1f s, loss %.9f" % (dt, time.time() - t0, float(loss.detach())))
        named = dict(ms.named_parameters())
        named["img_encoder.linear.weight"] = named["img_encoder.lin"]
        out["loss" + tag] = loss.detach()
        for n in F8_GRADS + ["img_encoder.linear.weight"]:
            g = named[n].grad
            key = n.replace(".", "_")
            flat = g.reshape(-1) if g.dim() < 2 else g.reshape(g.shape[0], -1)
            out["g%s_%s" % (tag, key)] = flat[:256] if g.dim() < 2 else flat[:8, :256]
            out["l1%s_%s" % (tag, key)] = g.double().abs().sum()
    npz(G("f8_fullstep.npz"), **out)


if __name__ == "__main__":
    main()
