"""Tiny end-to-end run of the hot path used by __graft_entry__.smoke(): one fused multimodal training
step (forward + backward + clip + AdamW) on cuda:0 at a small BART-large-width configuration."""
import torch

from . import optim, synthetic as syn
from .config import BartConfig
from .modules import MultimodalSum


def tiny_config(dropout=0.0):
    return BartConfig(vocab_size=200, d_model=1024, encoder_ffn_dim=64, decoder_ffn_dim=64, encoder_layers=1, decoder_layers=1,
                      encoder_attention_heads=16, decoder_attention_heads=16, max_position_embeddings=32, dropout=dropout)


def run_step(state_dict=None, dtype=torch.float32, seed=31, B=2, NR=3, S=16, I=2, img_hw=224):
    """Returns (model, cpu batch, loss) after one full training step's forward/backward."""
    cfg = tiny_config()
    model = MultimodalSum(config=cfg, label_smoothing=0.1, device="cuda:0", dtype=dtype, deterministic=True)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    model.train()
    bc = syn.yelp_batch(B, NR, S, I, cfg.vocab_size, seed=seed, img_hw=img_hw)
    b = syn.batch_to(bc, "cuda:0")
    loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])[0]
    loss.backward()
    torch.cuda.synchronize()
    return model, bc, loss
