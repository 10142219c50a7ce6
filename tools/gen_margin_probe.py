#!/usr/bin/env python3
"""CPU only (development container): how long the generation oracle takes at the sizes of the id-exact GPU tests and how far its
rankings are from ties (smallest gap between consecutive candidates over all decode steps).  usage: gen_margin_probe.py layers max_length [std] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import synthetic as syn
from multimodalsum_amd.config import BartConfig
from multimodalsum_amd.formula_init import formula_state_dict, formula_tensor
from oracle import bart_oracle as bo, generate_oracle as go

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
layers, max_length = int(sys.argv[1]), int(sys.argv[2])
std = float(sys.argv[3]) if len(sys.argv) > 3 else 0.06
Bz = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=layers, decoder_layers=layers,
                  heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.0)
sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=std)
N, S = 8, 128
ids = syn.token_batch(Bz * N, S, cfg.vocab_size, seed=21, mean_len=75.0, std_len=20.0, min_len=32).view(Bz, N, S)
text_m = ids.ne(1).clone()
table_h = formula_tensor("g.table_h", (Bz, 1, 47, cfg.d_model), std=1.0)
img_h = formula_tensor("g.img_h", (Bz, 2, 196, cfg.d_model), std=1.0)
table_m = torch.ones(Bz, 1, 47, dtype=torch.bool)
img_m = torch.ones(Bz, 2, 196, dtype=torch.bool)
img_m[Bz - 1, 1] = False
kw = dict(num_beams=4, max_length=max_length, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0)
rd = torch.zeros(Bz, 1)
with torch.no_grad():
    t0 = time.time()
    oenc = bo.bart_encoder(sd, ocfg, ids.view(-1, S), text_m.view(-1, S)).view(Bz, N, S, -1)
    t1 = time.time()
    margins = []
    ref = go.beam_search(sd, ocfg, [oenc, table_h, img_h], [text_m, table_m, img_m], rd, True, decoder_start_token_id=cfg.bos_token_id,
                         margins=margins, **kw)
    t2 = time.time()
print("encoder %.1f s, beam search %.1f s, out shape %s" % (t1 - t0, t2 - t1, tuple(ref.shape)))
print("steps %d, min margin %.3e, 5 smallest %s" % (len(margins), min(margins), ["%.2e" % m for m in sorted(margins)[:5]]))
print(ref[:, :40])
