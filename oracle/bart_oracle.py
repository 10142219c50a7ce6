"""TEST INFRASTRUCTURE -- CPU oracle (PyTorch fp32) for the MultimodalSum training hot path.

This file restates, op for op, the arithmetic of the reference's modified BART
(/root/reference/src/transformer/modeling_multimodalsum.py) as plain functions over a
state_dict with the reference's key names.  It is the *checker* for the HIP path: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package never does (and fails loudly without its HIP library).

Pinning: the reference has no tests for this path, so the oracle is pinned
against outputs of the reference itself, run in the development container by
`oracle/make_golden.py`, committed under `tests/golden/` and checked by
`tests/test_oracle_golden.py`.

Everything is literal: 9 sequential leave-one-out decoder passes, K/V re-projected in every
pass, q projected once per modality call, unfused loss.  No restructuring happens here.
"""
import math

import torch
import torch.nn.functional as F

NEG_FILL_CROSS = -2.0 ** 16  # modeling_multimodalsum.py:844 (masked_fill value, not additive)

# Generation only: the reference projects the cross-attention keys / values of a layer ONCE per generate() call and keeps them in
# the layer's cache (modeling_multimodalsum.py:804-815, _use_saved_state :889-920); the restatement re-runs the whole decoder over
# the prefix at every step, so inside `with kv_memo():` the K / V projections of a memory tensor are computed once per (layer, tensor)
# and reused -- identical arithmetic, ~10x less time at BART-large width.  Off (None) everywhere else (training steps need autograd).
KV_MEMO = None


class kv_memo:
    def __enter__(self):
        global KV_MEMO
        self.prev, KV_MEMO = KV_MEMO, {}
        return self

    def __exit__(self, *exc):
        global KV_MEMO
        KV_MEMO = self.prev
        return False


class BartCfg:
    """Subset of BartConfig the path reads (/root/reference/cfg/bart-large.json)."""

    def __init__(self, vocab_size=50265, d_model=1024, ffn_dim=4096, encoder_layers=12,
                 decoder_layers=12, heads=16, max_position_embeddings=1024, pad_token_id=1,
                 bos_token_id=0, eos_token_id=2, dropout=0.1, extra_pos_embeddings=2):
        self.vocab_size = vocab_size
        self.d_model = d_model
        self.ffn_dim = ffn_dim
        self.encoder_layers = encoder_layers
        self.decoder_layers = decoder_layers
        self.heads = heads
        self.max_position_embeddings = max_position_embeddings
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.dropout = dropout
        self.extra_pos_embeddings = extra_pos_embeddings


def bart_param_shapes(cfg, multimodal=True, prefix=""):
    """Names/shapes exactly as `BartForMultiEncConditionalGeneration.state_dict()` dumps them
    (SURVEY.md section 8b).  embed_tokens alias `shared` and are listed once under `shared`."""
    D, Fd, V = cfg.d_model, cfg.ffn_dim, cfg.vocab_size
    P = cfg.max_position_embeddings + cfg.extra_pos_embeddings
    s = {}
    s[prefix + "model.shared.weight"] = (V, D)
    for side, nl in (("encoder", cfg.encoder_layers), ("decoder", cfg.decoder_layers)):
        b = prefix + "model.%s." % side
        s[b + "embed_positions.weight"] = (P, D)
        s[b + "layernorm_embedding.weight"] = (D,)
        s[b + "layernorm_embedding.bias"] = (D,)
        if side == "decoder":
            s[b + "rating_embeddings"] = (D,)
        for i in range(nl):
            lb = b + "layers.%d." % i
            attns = ["self_attn"] + (["encoder_attn"] if side == "decoder" else [])
            for a in attns:
                for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
                    s[lb + a + "." + p + ".weight"] = (D, D)
                    s[lb + a + "." + p + ".bias"] = (D,)
                if a == "encoder_attn" and multimodal:
                    for p in ("alpha_proj", "beta_proj"):
                        s[lb + a + "." + p + ".weight"] = (D, 2 * D)
                        s[lb + a + "." + p + ".bias"] = (D,)
                s[lb + a + "_layer_norm.weight"] = (D,)
                s[lb + a + "_layer_norm.bias"] = (D,)
            s[lb + "fc1.weight"] = (Fd, D)
            s[lb + "fc1.bias"] = (Fd,)
            s[lb + "fc2.weight"] = (D, Fd)
            s[lb + "fc2.bias"] = (D,)
            s[lb + "final_layer_norm.weight"] = (D,)
            s[lb + "final_layer_norm.bias"] = (D,)
    return s


# bf16 emulation (tests only): with EMULATE_BF16 on, every Linear of the path rounds its operands and its result to bf16
# (f32 accumulation), forward and backward, the way ANY bf16 implementation of the same algorithm must; so do the attention
# products (attn_matmul), the LayerNorm outputs a bf16 model keeps in memory (store) and the gradient of a tensor with several
# consumers, which such a model accumulates in bf16 (fan_out).  The difference
# between the emulated and the plain f32 run is the yardstick the bf16 HIP path is held to (tests/test_bench_shapes_gpu.py):
# its error against the f32 oracle may be a small multiple of this one, per tensor.  Off (None) = the reference's arithmetic.
EMULATE_BF16 = False


def _q(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _QuantLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xq, wq = _q(x), _q(w)
        ctx.save_for_backward(xq, wq)
        ctx.has_bias = b is not None
        y = F.linear(xq, wq, b)
        return _q(y)

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved_tensors
        dyq = _q(dy)
        dx = _q(dyq @ wq)
        dw = dyq.reshape(-1, dyq.shape[-1]).t() @ xq.reshape(-1, xq.shape[-1])
        db = dyq.reshape(-1, dyq.shape[-1]).sum(0) if ctx.has_bias else None
        return dx, dw, db


def linear(x, w, b=None):
    if EMULATE_BF16:
        return _QuantLinear.apply(x, w, b)
    return F.linear(x, w, b)


def _unbroadcast(g, shape):
    """Sum a gradient over the dimensions torch.matmul broadcast."""
    while g.dim() > len(shape):
        g = g.sum(0)
    for i, (gs, ss) in enumerate(zip(g.shape, shape)):
        if ss == 1 and gs != 1:
            g = g.sum(i, keepdim=True)
    return g


class _QuantMatmul(torch.autograd.Function):
    """The attention products (scores = q k^T, out = p v) as ANY bf16 matrix-core implementation runs them: both operands
    rounded to bf16 (the probabilities and, in backward, the score gradients included), f32 accumulation, forward and backward."""

    @staticmethod
    def forward(ctx, a, b):
        aq, bq = _q(a), _q(b)
        ctx.save_for_backward(aq, bq)
        return torch.matmul(aq, bq)

    @staticmethod
    def backward(ctx, g):
        aq, bq = ctx.saved_tensors
        gq = _q(g)
        return _unbroadcast(torch.matmul(gq, bq.transpose(-1, -2)), aq.shape), _unbroadcast(torch.matmul(aq.transpose(-1, -2), gq), bq.shape)


class _QuantStore(torch.autograd.Function):
    """An activation a bf16 model keeps in memory: rounded to bf16 when written (forward) and so is its gradient (backward)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _q(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_q(g) if ctx.bwd else g), None, None


def store(x, fwd=True, bwd=True):
    """Identity; with the bf16 emulation on (tests only), a tensor stored in bf16 (its gradient too)."""
    if EMULATE_BF16:
        return _QuantStore.apply(x, fwd, bwd)
    return x


class _QuantFanOut(torch.autograd.Function):
    """A tensor read by n consumers (the encoder memories, read by every decoder layer): a bf16 model accumulates the consumers'
    gradients in the tensor's own dtype, in the order the backward pass produces them (last consumer first)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.clone() for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        acc = None
        for g in reversed(gs):
            if g is None:
                continue
            acc = _q(g) if acc is None else _q(acc + g)
        return acc, None


def fan_out(x, n):
    if EMULATE_BF16:
        return list(_QuantFanOut.apply(x, n))
    return [x] * n


def attn_matmul(a, b):
    """torch.matmul for the two products of an attention; with the bf16 emulation on (tests only), with bf16 operands."""
    if EMULATE_BF16:
        return _QuantMatmul.apply(a, b)
    return torch.matmul(a, b)


def _lin(sd, name, x):
    return linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x):
    # emulation: the normalised rows are stored in bf16; so is the gradient of the sum they were computed from
    return store(F.layer_norm(store(x, fwd=False), (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5))


# Injected dropout masks (tests only): None = torch's own stream; otherwise a callable(shape) -> bool keep tensor, called once per dropout
# site IN THE ORDER THE REFERENCE REACHES THEM (encoder: embedding :371, per layer self-attention :294, FFN :305; every decoder pass:
# embedding :596, per layer self-attention :458, cross-attention :474, FFN :486).  The HIP kernels' masks are a counter hash, not
# Philox (multimodalsum_amd/dropout.py): handing the same masks to this restatement is how a dropout-on step is compared tensor by tensor.
DROPOUT_MASKS = None


def _drop(x, p, training):
    if DROPOUT_MASKS is not None and training and p > 0.0:
        keep = DROPOUT_MASKS(tuple(x.shape))
        assert keep.shape == x.shape and keep.dtype == torch.bool
        return x * (keep.to(x.dtype) * (1.0 / (1.0 - p)))          # F.dropout: mask * 1/(1-p), elementwise (the kernels apply it in registers: nothing stored)
    return F.dropout(x, p=p, training=training)


# --------------------------------------------------------------------------------------------
# decoder input preparation  (modeling_multimodalsum.py:160-181, 225-254)
# --------------------------------------------------------------------------------------------
def shift_tokens_right(input_ids, pad, bos, eos):
    """Q6: first token chosen from ROW 0 for the whole batch; last non-pad token -> pad; shift."""
    n_real = input_ids.ne(pad).sum(dim=1)
    last = (n_real - 1).unsqueeze(-1)  # negative index wraps like scatter_ would fail; reference
    # rows always hold >=1 real token, so last >= 0
    onehot = torch.zeros_like(input_ids).scatter_(1, last, 1).bool()
    body = torch.where(onehot, torch.full_like(input_ids, pad), input_ids)
    first = bos if int(input_ids[0, 0]) != bos else eos
    out = input_ids.clone()
    out[:, 0] = first
    out[:, 1:] = body[:, :-1]
    return out


def decoder_inputs_from_labels(cfg, labels):
    dec_in = shift_tokens_right(labels, cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id)
    pad_mask = dec_in.eq(cfg.pad_token_id)
    if not bool(pad_mask.any()):
        pad_mask = None
    T = dec_in.shape[1]
    causal = torch.triu(torch.full((T, T), float("-inf")), 1)
    return dec_in, pad_mask, causal


# --------------------------------------------------------------------------------------------
# attention  (modeling_multimodalsum.py:672-920)
# --------------------------------------------------------------------------------------------
def self_attention(sd, pre, x, heads, key_pad=None, causal=None):
    """x [T,B,D] time-major.  key_pad [B,T] True=pad -> -inf (:836-837); causal additive (:822-827)."""
    T, B, D = x.shape
    hd = D // heads
    q = _lin(sd, pre + ".q_proj", x) * hd ** -0.5
    k = _lin(sd, pre + ".k_proj", x)
    v = _lin(sd, pre + ".v_proj", x)

    def sh(t):
        return t.contiguous().view(T, B * heads, hd).transpose(0, 1)

    q, k, v = sh(q), sh(k), sh(v)
    w = attn_matmul(q, k.transpose(1, 2))  # [B*H,T,T]   (torch.bmm, :819)
    if causal is not None:
        w = (w.view(B, heads, T, T) + causal).view(B * heads, T, T)
    if key_pad is not None:
        w = w.view(B, heads, T, T).masked_fill(key_pad[:, None, None, :], float("-inf")).view(B * heads, T, T)
    p = F.softmax(w, dim=-1)
    o = attn_matmul(p, v)  # [B*H,T,hd]   (torch.bmm, :869)
    o = o.transpose(0, 1).contiguous().view(T, B, D)
    return _lin(sd, pre + ".out_proj", o)


def entity_cross_attention_heads(sd, pre, x, keys, pad, heads):
    """Per-entity softmax + entity mean (:839-869).

    x [T,B,D]; keys [S,N,B,D]; pad [B,N,S] bool True=pad or None.
    Returns the head-merged, pre-out_proj tensor [T,B,D]."""
    T, B, D = x.shape
    S, N = keys.shape[0], keys.shape[1]
    hd = D // heads
    q = _lin(sd, pre + ".q_proj", x) * hd ** -0.5           # [T,B,D]
    memo_key = (pre, keys.data_ptr(), tuple(keys.shape), tuple(keys.stride()), keys.dtype) if (KV_MEMO is not None and not torch.is_grad_enabled()) else None
    if memo_key is not None and memo_key in KV_MEMO:
        k, v = KV_MEMO[memo_key]
    else:
        k = _lin(sd, pre + ".k_proj", keys)                   # [S,N,B,D]
        v = _lin(sd, pre + ".v_proj", keys)
        if memo_key is not None:
            KV_MEMO[memo_key] = (k, v)
    qh = q.view(T, B, heads, hd).permute(1, 2, 0, 3)          # [B,H,T,hd]
    kh = k.view(S, N, B, heads, hd).permute(1, 2, 3, 0, 4)    # [N,B,H,S,hd]
    vh = v.view(S, N, B, heads, hd).permute(1, 2, 3, 0, 4)
    a = attn_matmul(qh.unsqueeze(0), kh.transpose(-1, -2))      # [N,B,H,T,S] = einsum("bhtd,nbhsd->nbhts")
    if pad is not None:
        a = a.masked_fill(pad.transpose(0, 1)[:, :, None, None, :], NEG_FILL_CROSS)
    p = F.softmax(a, dim=-1)
    o = attn_matmul(p, vh)                                     # [N,B,H,T,hd] = einsum("nbhts,nbhsd->nbhtd")
    if pad is not None:
        null = pad.all(dim=-1)                                 # [B,N]  (:858)
        valid = (~null).transpose(0, 1).to(o.dtype)            # [N,B]
        o_sum = (o * valid[:, :, None, None, None]).sum(dim=0)
        cnt = valid.sum(dim=0)                                 # [B]
        cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)  # (:864-865)
        o = o_sum / cnt[:, None, None, None]
    else:
        o = o.mean(dim=0)
    return o.permute(2, 0, 1, 3).contiguous().view(T, B, D)    # heads -> D  (:884)


def cross_attention(sd, pre, x, keys, pads, heads, multimodal):
    """A3: three modalities with SHARED q/k/v/out weights + alpha/beta gated fusion (:722-745)."""
    if not multimodal:
        return _lin(sd, pre + ".out_proj", entity_cross_attention_heads(sd, pre, x, keys, pads, heads))
    ys = []
    for m in range(3):
        ys.append(_lin(sd, pre + ".out_proj", entity_cross_attention_heads(sd, pre, x, keys[m], pads[m], heads)))
    y_text, y_table, y_img = ys
    non_table = pads[1].all(dim=2)[:, 0]                        # [B]   (:732)
    non_img = pads[2].all(dim=2).all(dim=1)                     # [B]   (:735)
    alpha = F.relu(torch.tanh(_lin(sd, pre + ".alpha_proj", torch.cat([y_text, y_table], dim=-1))))
    beta = F.relu(torch.tanh(_lin(sd, pre + ".beta_proj", torch.cat([y_text, y_img], dim=-1))))
    alpha = alpha.masked_fill(non_table[None, :, None], 0.0)
    beta = beta.masked_fill(non_img[None, :, None], 0.0)
    return y_text + alpha * y_table + beta * y_img


# --------------------------------------------------------------------------------------------
# encoder / decoder stacks
# --------------------------------------------------------------------------------------------
def bart_encoder(sd, cfg, input_ids, attention_mask, training=False, prefix=""):
    """BartEncoder.forward (:346-404).  input_ids [Bn,S]; attention_mask 1=keep.  -> [Bn,S,D]."""
    b = prefix + "model.encoder."
    pad = attention_mask.eq(0) if attention_mask is not None else None
    S = input_ids.shape[1]
    pos = torch.arange(S) + cfg.extra_pos_embeddings
    x = F.embedding(input_ids, sd[prefix + "model.shared.weight"], padding_idx=cfg.pad_token_id) \
        + sd[b + "embed_positions.weight"][pos]      # padding_idx: no lookup-grad for the pad row (:1001)
    x = _ln(sd, b + "layernorm_embedding", x)
    x = _drop(x, cfg.dropout, training).transpose(0, 1)
    for i in range(cfg.encoder_layers):
        lb = b + "layers.%d" % i
        a = self_attention(sd, lb + ".self_attn", x, cfg.heads, key_pad=pad)
        x = _ln(sd, lb + ".self_attn_layer_norm", x + _drop(a, cfg.dropout, training))
        h = F.gelu(_lin(sd, lb + ".fc1", x))
        h = _lin(sd, lb + ".fc2", h)
        x = _ln(sd, lb + ".final_layer_norm", x + _drop(h, cfg.dropout, training))
    return x.transpose(0, 1)


def bart_decoder(sd, cfg, dec_in, hiddens, masks, dec_pad, causal, rating_diff, multimodal,
                 training=False, prefix=""):
    """BartDecoder.forward (:530-660), training path (no cache).

    hiddens: list of [B,N,S,D] (multimodal) or one tensor; masks: same structure, 1/True=keep."""
    b = prefix + "model.decoder."
    T = dec_in.shape[1]
    pos = torch.arange(T) + cfg.extra_pos_embeddings
    x = F.embedding(dec_in, sd[prefix + "model.shared.weight"], padding_idx=cfg.pad_token_id) \
        + sd[b + "embed_positions.weight"][pos]
    if rating_diff is not None:
        x = x + (rating_diff * sd[b + "rating_embeddings"]).unsqueeze(1)   # (:591-593)
    x = _ln(sd, b + "layernorm_embedding", x)
    x = _drop(x, cfg.dropout, training).transpose(0, 1)                    # [T,B,D]
    nl = cfg.decoder_layers
    if multimodal:
        keys = [fan_out(h.transpose(0, -2), nl) for h in hiddens]           # [S,N,B,D], one handle per layer (the same tensor)
        pads = [m.eq(0) for m in masks]
    else:
        keys = fan_out(hiddens.transpose(0, -2), nl)
        pads = masks.eq(0) if masks is not None else None
    for i in range(cfg.decoder_layers):
        lb = b + "layers.%d" % i
        a = self_attention(sd, lb + ".self_attn", x, cfg.heads, key_pad=dec_pad, causal=causal)
        x = _ln(sd, lb + ".self_attn_layer_norm", x + _drop(a, cfg.dropout, training))
        c = cross_attention(sd, lb + ".encoder_attn", x, [km[i] for km in keys] if multimodal else keys[i], pads, cfg.heads, multimodal)
        x = _ln(sd, lb + ".encoder_attn_layer_norm", x + _drop(c, cfg.dropout, training))
        h = F.gelu(_lin(sd, lb + ".fc1", x))
        h = _lin(sd, lb + ".fc2", h)
        x = _ln(sd, lb + ".final_layer_norm", x + _drop(h, cfg.dropout, training))
    return x.transpose(0, 1)                                                # [B,T,D]


def multienc_forward(sd, cfg, text_h, text_m, table_h, table_m, img_h, img_m, rating_diff, labels,
                     training=False, prefix=""):
    """BartForMultiEncConditionalGeneration.forward (:2206-2292), labels path -> lm_logits [B,T,V]."""
    dec_in, dec_pad, causal = decoder_inputs_from_labels(cfg, labels)
    h = bart_decoder(sd, cfg, dec_in, [text_h, table_h, img_h], [text_m, table_m, img_m], dec_pad, causal,
                     rating_diff, True, training, prefix)
    return linear(h, sd[prefix + "model.shared.weight"])  # final_logits_bias is a zero buffer (:2189)


def enc_forward(sd, cfg, enc_h, rating_diff, enc_m, labels, training=False, prefix=""):
    """BartForEncConditionalGeneration.forward (:1317-1396): text-only variant, single key tensor."""
    dec_in, dec_pad, causal = decoder_inputs_from_labels(cfg, labels)
    h = bart_decoder(sd, cfg, dec_in, enc_h, enc_m, dec_pad, causal, rating_diff, False, training, prefix)
    return linear(h, sd[prefix + "model.shared.weight"])


# --------------------------------------------------------------------------------------------
# loss  (/root/reference/src/utils.py:24-38)
# --------------------------------------------------------------------------------------------
def label_smoothing_loss(logits, target, classes, smoothing):
    """Pads are NOT ignored (Q2).  logits [R,V], target [R]."""
    logp = logits.log_softmax(dim=-1)
    with torch.no_grad():
        td = torch.full_like(logp, smoothing / (classes - 1))
        td.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    return torch.mean(torch.sum(-td * logp, dim=-1))
