"""TEST INFRASTRUCTURE -- CPU oracle for the leave-one-out training step and the optimiser.

Restates /root/reference/src/multimodal_train.py:124-193 (MultimodalSum.forward +
get_multimodal_outputs), /root/reference/src/text_pretrain.py:71-113 (TextSupervised.forward),
/root/reference/src/transformer/optimization.py:70-96,208-267 (linear warm-up schedule, HF AdamW)
and /root/reference/src/train_utils.py:49-57 (get_optimizer, with quirk Q1).
Only tests/, smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import torch

from . import bart_oracle as bo
from . import encoders_oracle as eo


def multimodal_outputs(sd, cfg, reviews, reviews_mask, field, field_value, img, img_mask, training,
                       running=None):
    """get_multimodal_outputs (multimodal_train.py:165-193)."""
    B, NR, S = reviews.shape
    text_h = bo.bart_encoder(sd, cfg, reviews.view(B * NR, S), reviews_mask.view(B * NR, S), training,
                             prefix="bart_model.").view(B, NR, S, -1)
    table_h, table_m = eo.yelp_table_encoder(sd, sd["bart_model.model.shared.weight"], field, field_value)
    I = img.shape[1]
    img_h = eo.resnet101_features(sd, img.reshape(-1, 3, img.shape[-2], img.shape[-1]), training, running)
    img_h = img_h.reshape(B, I, -1, cfg.d_model)
    img_m = img_mask.unsqueeze(-1).repeat(1, 1, img_h.shape[2])
    return text_h, reviews_mask, table_h.unsqueeze(1), table_m.unsqueeze(1), img_h, img_m


def multimodal_step_loss(sd, cfg, reviews, reviews_mask, reviews_rating, field, field_value, img, img_mask,
                         label_smoothing=0.1, training=False, running=None, return_parts=False):
    """MultimodalSum.forward (multimodal_train.py:124-163): NR leave-one-out decoder passes."""
    text_h, text_m, table_h, table_m, img_h, img_m = multimodal_outputs(
        sd, cfg, reviews, reviews_mask, field, field_value, img, img_mask, training, running)
    NR = reviews.shape[1]
    losses = []
    for i in range(NR):
        others = [j for j in range(NR) if j != i]
        rating_diff = reviews_rating[:, i] - reviews_rating[:, others].mean(dim=1)
        logits = bo.multienc_forward(sd, cfg, text_h[:, others], text_m[:, others], table_h, table_m, img_h, img_m,
                                     rating_diff.unsqueeze(1), reviews[:, i], training, prefix="bart_model.")
        losses.append(bo.label_smoothing_loss(logits.view(-1, cfg.vocab_size), reviews[:, i].reshape(-1),
                                              cfg.vocab_size, label_smoothing))
    loss = torch.mean(torch.stack(losses))
    return (loss, losses) if return_parts else loss


def text_step_loss(sd, cfg, reviews, reviews_mask, reviews_rating, label_smoothing=None, training=False,
                   prefix="bart_model.", return_parts=False):
    """TextSupervised.forward (text_pretrain.py:71-113): text-only leave-one-out.  With
    label_smoothing None the reference uses nn.CrossEntropyLoss (text_pretrain.py:97)."""
    B, NR, S = reviews.shape
    text_h = bo.bart_encoder(sd, cfg, reviews.view(B * NR, S), reviews_mask.view(B * NR, S), training,
                             prefix=prefix).view(B, NR, S, -1)
    losses = []
    for i in range(NR):
        others = [j for j in range(NR) if j != i]
        rating_diff = reviews_rating[:, i] - reviews_rating[:, others].mean(dim=1)
        logits = bo.enc_forward(sd, cfg, text_h[:, others], rating_diff.unsqueeze(1), reviews_mask[:, others],
                                reviews[:, i], training, prefix=prefix)
        flat, tgt = logits.view(-1, cfg.vocab_size), reviews[:, i].reshape(-1)
        if label_smoothing is None:
            losses.append(torch.nn.functional.cross_entropy(flat, tgt))
        else:
            losses.append(bo.label_smoothing_loss(flat, tgt, cfg.vocab_size, label_smoothing))
    loss = torch.mean(torch.stack(losses))
    return (loss, losses) if return_parts else loss


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
NO_DECAY = ('bias', 'bn1.weight', 'bn2.weight', 'bn3.weight', 'layer_norm.weight',
            'layernorm_embedding.weight')  # multimodal_train.py:462


def q1_param_groups(named_parameters):
    """get_optimizer (train_utils.py:49-57).  Q1: the reference passes a *generator* that the first
    comprehension exhausts, so the no-decay group is EMPTY and those parameters are never updated.
    Q1b (consequence): `optimizer.zero_grad()` never clears their .grad either, so those gradients
    accumulate over steps and are still seen by clip_grad_norm_(model.parameters())."""
    named = list(named_parameters)
    decay = [p for n, p in named if not any(nd in n for nd in NO_DECAY)]
    return [{"params": decay, "weight_decay": 0.01}, {"params": [], "weight_decay": 0.0}]


def linear_schedule_lambda(step, warmup, total):
    """get_linear_schedule_with_warmup (optimization.py:88-94)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0):
    """One HF-AdamW update in place (optimization.py:240-265): decoupled decay applied AFTER the
    Adam step using the already-updated p; eps outside the bias-corrected sqrt."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (multimodal_train.py:361-362)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in grads:
            g.mul_(coef)
    return total
