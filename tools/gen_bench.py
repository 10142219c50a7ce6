#!/usr/bin/env python3
"""Generation throughput at BASELINE config 5 (test.py:153-158 call): B businesses x (8 reviews x 128 tok, table, 4 images),
num_beams=4, max_length=128, no_repeat_ngram_size=3, early_stopping=True; random-init BART-large, bf16.
Reports generated tokens/s and summaries/s (encoders + decode).  usage: gen_bench.py [B] [max_length]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodalsum_amd as mm
from multimodalsum_amd import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
max_length = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = mm.BartConfig.from_json_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfg", "bart-large.json"))
model = mm.MultimodalSum(config=cfg, label_smoothing=0.1, device="cuda", dtype=torch.bfloat16)
model.eval()
b = syn.batch_to(syn.yelp_batch(B, 8, 128, 4, cfg.vocab_size, seed=7, img_hw=224), "cuda")


def run():
    with torch.no_grad():
        _, th, tm, tabh, tabm, ih, im = model.get_multimodal_outputs(b["reviews"], b["reviews_mask"], b["field"], b["field_value"], b["img"], b["img_mask"])
        rd = torch.zeros(B, 1, device="cuda")
        return model.bart_model.generate(th, tm, tabh, tabm, ih, im, rating_diff=rd, num_beams=4, length_penalty=1.0, max_length=max_length,
                                         no_repeat_ngram_size=3, early_stopping=True)


out = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = out.shape[1] - 1
print("B=%d beams=4 max_length=%d: %d decode steps in %.3f s -> %.1f tokens/s (best hypotheses), %.2f summaries/s, %.2f ms/step"
      % (B, max_length, steps, dt, B * steps / dt, B / dt, dt / max(steps, 1) * 1e3))
