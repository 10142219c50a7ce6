"""Drop-in nn.Modules: the reference's constructors, forward() signatures, attribute paths and
state_dict keys over the HIP engine.

  BartForMultiEncConditionalGeneration  <- modeling_multimodalsum.py:2181-2292
  BartForEncConditionalGeneration       <- modeling_multimodalsum.py:1292-1396
  YelpTableEncoder                      <- table_encoder.py:5-83
  Resnet                                <- img_encoder.py:4-41
  MultimodalSum / TextSupervised        <- multimodal_train.py:111-193 / text_pretrain.py:66-113

Every forward runs HIP kernels only; torch.autograd sees one coarse node per sub-program whose
backward is the engine's explicit reverse schedule (parameter gradients are written into the flat
gradient arena and attached as `p.grad` views).
"""
import json
import os

import torch
import torch.nn as nn

from . import kernels as kn
from .config import BartConfig
from .engine import Engine, resnet_blocks
from .formula_init import formula_state_dict


class _Node(nn.Module):
    pass


def _attach(root, dotted, tensor, is_buffer=False):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Node())
        node = getattr(node, p)
    if is_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], tensor)


def _get(root, dotted):
    node = root
    for p in dotted.split("."):
        node = getattr(node, p)
    return node


# ------------------------------------------------------------------------------------------------
# checkpoint contract: strict loads with an explicit alias map
# ------------------------------------------------------------------------------------------------
_STAGE_ALIASES = (("stage1.0.", "resnet.conv1."), ("stage1.1.", "resnet.bn1."), ("stage1.4.", "resnet.layer1."),
                  ("stage2.0.", "resnet.layer2."), ("stage3.0.", "resnet.layer3."))


def canonical_key(key):
    """The key of the tensor an aliased state_dict entry shares storage with (None if `key` is not an alias).  The reference
    registers the same Parameter under several names: encoder/decoder.embed_tokens and table_encoder.bart_embedding alias
    model.shared (modeling_multimodalsum.py:1001-1006, multimodal_train.py:117), img_encoder.stage{1,2,3} alias
    img_encoder.resnet.* (img_encoder.py:21-24)."""
    for tail in ("model.encoder.embed_tokens.weight", "model.decoder.embed_tokens.weight"):
        if key.endswith(tail):
            return key[:-len(tail)] + "model.shared.weight"
    if key.endswith("table_encoder.bart_embedding.weight"):
        return key[:-len("table_encoder.bart_embedding.weight")] + "bart_model.model.shared.weight"
    for alias, real in _STAGE_ALIASES:
        i = key.find(alias)
        if i >= 0 and (i == 0 or key[i - 1] == "."):
            return key[:i] + real + key[i + len(alias):]
    return None


def complete_aliases(module, state_dict):
    """A copy of `state_dict` in which every aliased key the module expects, and the dict lacks, is filled from its canonical
    tensor (checkpoints written by code that de-duplicates shared tensors, oracle-side dicts keyed by canonical names);
    `final_logits_bias` -- a constant zero buffer (:2189) -- is filled with zeros when absent."""
    sd = dict(state_dict)
    for k, v in module.state_dict().items():
        if k in sd:
            continue
        c = canonical_key(k)
        if c is not None and c in sd:
            sd[k] = sd[c]
        elif k.endswith("final_logits_bias"):
            sd[k] = torch.zeros_like(v)
    return sd


def load_pretrained(module, path, allowed_missing=()):
    """`module.load_state_dict(torch.load(<path>/pytorch_model.bin))` as the reference does it for its stage hand-offs
    (multimodal_train.py:116-122): the file must exist, no key may be unexpected, and the only keys that may be missing are
    those matching `allowed_missing` (for BART the reference's `authorized_missing_keys`, :2183, plus the parameters its
    fine-tuning adds to facebook/bart-large)."""
    import re
    ckpt = os.path.join(str(path), "pytorch_model.bin")
    if not os.path.isfile(ckpt):
        raise FileNotFoundError("pretrained weights %r not found (%s is missing; hub names cannot be fetched: no network)" % (str(path), ckpt))
    sd = complete_aliases(module, torch.load(ckpt, map_location="cpu"))
    res = nn.Module.load_state_dict(module, sd, strict=False)
    bad_missing = [k for k in res.missing_keys if not any(re.search(p, k) for p in allowed_missing)]
    if res.unexpected_keys or bad_missing:
        raise RuntimeError("checkpoint %s does not match %s: unexpected keys %s, missing keys %s"
                           % (ckpt, type(module).__name__, sorted(res.unexpected_keys)[:8], sorted(bad_missing)[:8]))
    module._engine.mark_weights_changed()
    return res


def bart_reference_names(cfg, multimodal):
    """Parameter names of the reference's BART in ITS registration order (BartModel.__init__ :1001-1006, BartEncoder :330-344,
    EncoderLayer :262-274, BartDecoder :510-528, DecoderLayer :409-430, SelfAttention :695-704): named_parameters() order
    decides the parameter indices inside optimizer.state_dict(), so it is part of the checkpoint contract.  The arena keeps
    its own order (decay group first, q/k/v adjacent); only the module tree follows this one."""
    names = ["model.shared.weight"]
    for side, nl in (("encoder", cfg.encoder_layers), ("decoder", cfg.decoder_layers)):
        b = "model.%s." % side
        if side == "decoder":
            names.append(b + "rating_embeddings")
        names.append(b + "embed_positions.weight")
        for i in range(nl):
            lb = b + "layers.%d." % i
            for att in (("self_attn",) if side == "encoder" else ("self_attn", "encoder_attn")):
                projs = ("k_proj", "v_proj", "q_proj", "out_proj")
                if att == "encoder_attn" and multimodal:
                    projs += ("alpha_proj", "beta_proj")
                for pr in projs:
                    names += [lb + att + "." + pr + ".weight", lb + att + "." + pr + ".bias"]
                names += [lb + att + "_layer_norm.weight", lb + att + "_layer_norm.bias"]
            names += [lb + "fc1.weight", lb + "fc1.bias", lb + "fc2.weight", lb + "fc2.bias",
                      lb + "final_layer_norm.weight", lb + "final_layer_norm.bias"]
        names += [b + "layernorm_embedding.weight", b + "layernorm_embedding.bias"]
    return names


def resnet_reference_names():
    """torchvision resnet101 parameter names in its registration order (conv1, bn1, layer1..4, fc), then the projection."""
    names = ["resnet.conv1.weight", "resnet.bn1.weight", "resnet.bn1.bias"]
    for li, bi, inp, pl, stride, down in resnet_blocks():
        b = "resnet.layer%d.%d." % (li, bi)
        for j in (1, 2, 3):
            names += [b + "conv%d.weight" % j, b + "bn%d.weight" % j, b + "bn%d.bias" % j]
        if down:
            names += [b + "downsample.0.weight", b + "downsample.1.weight", b + "downsample.1.bias"]
    return names + ["resnet.fc.weight", "resnet.fc.bias", "linear.weight"]


def _attach_ordered(root, engine, prefix, ordered):
    """Register the arena parameters named prefix + n, n in `ordered`, on `root` in that order (all of them: the two name
    lists must agree)."""
    have = {n for n in engine.arena.params if n.startswith(prefix)}
    want = [prefix + n for n in ordered]
    assert have == set(want), sorted(have ^ set(want))[:6]
    for n in want:
        _attach(root, n[len(prefix):], engine.arena.params[n])


BART_ALLOWED_MISSING = (r"alpha_proj", r"beta_proj", r"rating_embeddings$", r"final_logits_bias$")


def _anchor(engine):
    if not hasattr(engine, "_anchor"):
        engine._anchor = torch.zeros(1, device=engine.device, requires_grad=True)
    return engine._anchor


def _begin_backward(engine):
    if not getattr(engine, "_grads_ready", False):
        engine.arena.prepare_grads()
        engine.touched = set()
        engine._grads_ready = True
        if engine.post_backward_hooks:
            # fires once, after the LAST node of this backward pass (what torch DDP uses as well)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: [cb() for cb in engine.post_backward_hooks])


def _end_backward(engine):
    engine.arena.attach_grads(engine.touched)


def _new_forward(engine):
    engine._grads_ready = False
    engine.sync_weights()


# ------------------------------------------------------------------------------------------------
# coarse autograd nodes
# ------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, engine, ids, mask):
        h, c = engine.encoder_fwd(ids, mask)
        ctx.engine, ctx.c = engine, c
        return h.view(ids.shape[0], ids.shape[1], -1)

    @staticmethod
    def backward(ctx, dh):
        e = ctx.engine
        _begin_backward(e)
        e.encoder_bwd(ctx.c, dh.reshape(-1, dh.shape[-1]).to(e.dtype).contiguous())
        _end_backward(e)
        return None, None, None, None


class _TableFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, engine, field, *fv):
        y, c = engine.table_fwd(field, list(fv))
        ctx.engine, ctx.c = engine, c
        ctx.mark_non_differentiable(c.mask)
        return y.view(c.B, engine.table_positions, -1), c.mask

    @staticmethod
    def backward(ctx, dy, _dmask):
        e = ctx.engine
        _begin_backward(e)
        e.table_bwd(ctx.c, dy.reshape(-1, dy.shape[-1]).to(e.dtype).contiguous())
        _end_backward(e)
        return (None,) * (3 + 6)


class _ImgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, engine, img):
        y, c = engine.img_fwd(img)
        ctx.engine, ctx.c = engine, c
        return y.view(img.shape[0], -1, y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        e = ctx.engine
        _begin_backward(e)
        e.img_bwd(ctx.c, dy.reshape(-1, dy.shape[-1]).to(e.dtype).contiguous())
        _end_backward(e)
        return None, None, None


class _DecoderLogitsFn(torch.autograd.Function):
    """decoder + LM head -> logits [Bd,T,V]; inputs: the per-modality encoder hiddens."""

    @staticmethod
    def forward(ctx, anchor, engine, dec_ids, dec_pad, rating_diff, pads, qpb, exclude_self, *hiddens):
        e = engine
        D = e.cfg.d_model
        B = hiddens[0].shape[0]
        layout = e.make_memory(B, [(h.shape[1], h.shape[2]) for h in hiddens])
        mem = e.empty(layout.rows, D)
        for m, h in enumerate(hiddens):
            n = h.shape[0] * h.shape[1] * h.shape[2]
            mem[layout.offs[m]:layout.offs[m] + n].copy_(h.reshape(n, D))
        hL, c = e.decoder_fwd(dec_ids, dec_pad, rating_diff, mem, layout, pads, qpb, exclude_self)
        logits = e.lm_logits_fwd(hL)
        ctx.engine, ctx.c, ctx.hL, ctx.shapes = e, c, hL, [tuple(h.shape) for h in hiddens]
        V = e.cfg.vocab_size
        return logits[:, :V].float().view(dec_ids.shape[0], dec_ids.shape[1], V)

    @staticmethod
    def backward(ctx, dlogits):
        e = ctx.engine
        V, D = e.cfg.vocab_size, e.cfg.d_model
        _begin_backward(e)
        dl = e.zeros(ctx.hL.shape[0], e.Vpad)
        dl[:, :V].copy_(dlogits.reshape(-1, V))
        dh = e.lm_head_bwd(ctx.hL, dl)
        dmem = e.decoder_bwd(ctx.c, dh)
        _end_backward(e)
        outs, L = [], ctx.c.layout
        for m, shp in enumerate(ctx.shapes):
            n = shp[0] * shp[1] * shp[2]
            outs.append(dmem[L.offs[m]:L.offs[m] + n].view(shp))
        return (None,) * 8 + tuple(outs)


def shift_tokens_right_batched(labels, first_rows, pad, bos, eos):
    """Quirk Q6 (modeling_multimodalsum.py:225-246): the last non-pad token becomes pad, tokens shift
    right, and the first token is BOS unless `first_rows`' first token already is BOS (then EOS).
    labels [..., T]; first_rows: the tensor whose [..., 0] decides per leading index (the reference
    looks at row 0 of each pass batch)."""
    n_real = labels.ne(pad).sum(dim=-1, keepdim=True)
    pos = torch.arange(labels.shape[-1], device=labels.device)
    body = torch.where(pos == (n_real - 1), torch.full_like(labels, pad), labels)
    first = torch.where(first_rows[..., 0] != bos, torch.full_like(first_rows[..., 0], bos), torch.full_like(first_rows[..., 0], eos))
    out = torch.empty_like(labels)
    out[..., 0] = first.expand(labels.shape[:-1])
    out[..., 1:] = body[..., :-1]
    return out


# ------------------------------------------------------------------------------------------------
# module tree
# ------------------------------------------------------------------------------------------------
class BartEncoder(_Node):
    def forward(self, input_ids, attention_mask=None, output_attentions=False, output_hidden_states=False, return_dict=False):
        e = self._engine
        _new_forward(e)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        return (_EncoderFn.apply(_anchor(e), e, input_ids, attention_mask),)


class _BartBase(nn.Module):
    multimodal = True

    def __init__(self, config, engine=None, prefix="", device="cuda", dtype=torch.bfloat16, deterministic=False):
        super().__init__()
        self.config = config
        self._own_engine = engine is None
        if engine is None:
            engine = Engine(config, device=device, compute_dtype=dtype, multimodal=self.multimodal, bart_prefix="",
                            deterministic=deterministic)
        object.__setattr__(self, "_engine", engine)
        self._prefix = prefix
        self._build_tree()

    def _build_tree(self):
        e, pre = self._engine, self._prefix
        model = _Node()
        self.add_module("model", model)
        _attach(self, "model.shared.weight", e.arena.params[pre + "model.shared.weight"])   # registered first, as in BartModel.__init__ (:1001)
        shared = model.shared.weight
        enc = BartEncoder()
        object.__setattr__(enc, "_engine", e)
        model.add_module("encoder", enc)
        model.add_module("decoder", _Node())
        for side in ("encoder", "decoder"):
            tok = _Node()
            tok.register_parameter("weight", shared)     # aliases of `shared` (state_dict keys encoder/decoder.embed_tokens.weight),
            getattr(model, side).add_module("embed_tokens", tok)       # the first child of either side (:330, :510)
        have = {n[len(pre):] for n in e.arena.params if n.startswith(pre + "model.")}
        want = bart_reference_names(self.config, self.multimodal)
        assert have == set(want), sorted(have ^ set(want))[:6]
        for n in want:
            _attach(self, n, e.arena.params[pre + n])
        self.register_buffer("final_logits_bias", e.buffers[pre + "final_logits_bias"])

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, config=None, **kw):
        """Construction contract of multimodal_train.py:116.  `config` is a BartConfig or a JSON path.
        The directory must hold pytorch_model.bin (there is no network to fetch hub names such as
        'facebook/bart-large': that raises); for a random-init model construct the class directly."""
        if isinstance(config, str):
            config = BartConfig.from_json_file(config)
        elif config is None:
            cfgp = os.path.join(str(pretrained_model_name_or_path), "config.json")
            config = BartConfig.from_json_file(cfgp)
        m = cls(config, **kw)
        init_formula(m)                     # parameters a bart-large checkpoint lacks (alpha/beta, rating) keep this init
        load_pretrained(m, pretrained_model_name_or_path, BART_ALLOWED_MISSING)
        m.eval()
        return m

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(complete_aliases(self, state_dict), strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def train(self, mode=True):
        super().train(mode)
        if self._own_engine:
            self._engine.training = mode
        return self

    def _decode(self, hiddens, masks, rating_diff, labels, decoder_input_ids=None, decoder_attention_mask=None):
        e = self._engine
        _new_forward(e)
        cfg = self.config
        if labels is None:
            raise NotImplementedError("single cached decoder steps are driven by generate() (multimodalsum_amd/generation.py); "
                                      "forward() implements the labels (training / scoring) path")
        if decoder_input_ids is None:
            decoder_input_ids = shift_tokens_right_batched(labels, labels[:1], cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id)
        if decoder_attention_mask is None:
            dec_pad = decoder_input_ids.eq(cfg.pad_token_id).to(torch.uint8).contiguous()
        else:
            dec_pad = decoder_attention_mask.eq(0).to(torch.uint8).contiguous()
        pads = [m.eq(0).to(torch.uint8).contiguous() for m in masks]
        rd = rating_diff.reshape(-1).float().contiguous() if rating_diff is not None else None
        hs = [h.to(e.dtype) for h in hiddens]
        return _DecoderLogitsFn.apply(_anchor(e), e, decoder_input_ids, dec_pad, rd, pads, 1, False, *hs)


    @torch.no_grad()
    def _generate(self, hiddens, masks, rating_diff, input_ids=None, max_length=None, min_length=None, do_sample=None,
                  early_stopping=None, num_beams=None, temperature=None, top_k=None, top_p=None, repetition_penalty=None,
                  bad_words_ids=None, bos_token_id=None, pad_token_id=None, eos_token_id=None, length_penalty=None,
                  no_repeat_ngram_size=None, num_return_sequences=None, decoder_start_token_id=None, use_cache=None, trace=None,
                  sample_draws=None, **unused):
        """generate() of the reference (modeling_multimodalsum.py:2295-2693 / :1398-1700): beam search as test.py:156-158 calls it,
        greedy decoding (num_beams == 1), bad_words_ids and repetition_penalty in either search, and (round 6) sampling with
        num_beams == 1: temperature, top_k in 1 .. 63 (the reference's default is 50) and top_p, one draw per row and step from
        torch.rand on the host -- torch.multinomial's own stream is not reproducible across implementations; `sample_draws`
        (callable(step, B) -> B uniforms; not a reference argument) lets tests and fixtures pin the draws.  Beam sampling, top_k = 0,
        prompts (input_ids) and more than one returned sequence are not built and raise."""
        from .generation import beam_search, greedy_search, sample_search
        cfg, e = self.config, self._engine
        pick = lambda v, d: d if v is None else v                                    # noqa: E731
        max_length, min_length = pick(max_length, cfg.max_length), pick(min_length, cfg.min_length)
        num_beams, early_stopping = pick(num_beams, cfg.num_beams), pick(early_stopping, cfg.early_stopping)
        length_penalty = pick(length_penalty, cfg.length_penalty)
        no_repeat_ngram_size = pick(no_repeat_ngram_size, cfg.no_repeat_ngram_size)
        start = pick(decoder_start_token_id, cfg.decoder_start_token_id)
        start = cfg.bos_token_id if start is None else start
        repetition_penalty = pick(repetition_penalty, getattr(cfg, "repetition_penalty", 1.0))
        do_sample = pick(do_sample, getattr(cfg, "do_sample", False))
        temperature = pick(temperature, getattr(cfg, "temperature", 1.0))
        top_k, top_p = pick(top_k, getattr(cfg, "top_k", 50)), pick(top_p, getattr(cfg, "top_p", 1.0))
        if (do_sample and num_beams != 1) or input_ids is not None or (num_return_sequences not in (None, 1)) or num_beams < 1:
            raise NotImplementedError("generate(): beam search, greedy decoding and sampling with num_beams = 1 (one returned sequence, no "
                                      "prompt) are built; beam sampling is not")
        assert max_length > 1 and min_length >= 0 and length_penalty > 0 and no_repeat_ngram_size >= 0 and repetition_penalty > 0
        if bad_words_ids is not None:
            assert all(isinstance(w, (list, tuple)) and len(w) > 0 and all(int(t) >= 0 for t in w) for w in bad_words_ids), \
                "`bad_words_ids` is a list of non-empty lists of non-negative token ids"
        was_training = e.training
        e.training = False
        try:
            _new_forward(e)
            B = hiddens[0].shape[0]
            layout = e.make_memory(B, [(h.shape[1], h.shape[2]) for h in hiddens])
            pads = [m.eq(0).to(torch.uint8).contiguous() for m in masks]
            if do_sample:
                return sample_search(e, hiddens, layout, pads, rating_diff, max_length, min_length, no_repeat_ngram_size, int(start),
                                     bad_words_ids=bad_words_ids, repetition_penalty=float(repetition_penalty), temperature=float(temperature),
                                     top_k=int(top_k), top_p=float(top_p), draws=sample_draws)
            if num_beams == 1:
                return greedy_search(e, hiddens, layout, pads, rating_diff, max_length, min_length, no_repeat_ngram_size, int(start),
                                     bad_words_ids=bad_words_ids, repetition_penalty=float(repetition_penalty))
            return beam_search(e, hiddens, layout, pads, rating_diff, num_beams, max_length, min_length, no_repeat_ngram_size,
                               bool(early_stopping), float(length_penalty), int(start), trace=trace, bad_words_ids=bad_words_ids,
                               repetition_penalty=float(repetition_penalty))
        finally:
            e.training = was_training


class BartForMultiEncConditionalGeneration(_BartBase):
    multimodal = True

    def generate(self, text_hiddens, text_attention_mask, table_hiddens, table_attention_mask, img_hiddens, img_attention_mask,
                 input_ids=None, rating_diff=None, **kw):
        return self._generate([text_hiddens, table_hiddens, img_hiddens], [text_attention_mask, table_attention_mask, img_attention_mask],
                              rating_diff, input_ids=input_ids, **kw)

    def forward(self, text_hiddens, text_attention_mask, table_hiddens, table_attention_mask, img_hiddens, img_attention_mask,
                rating_diff=None, decoder_input_ids=None, decoder_attention_mask=None, decoder_past_key_values=None, labels=None,
                use_cache=None, output_attentions=False, output_hidden_states=False, return_dict=False, **unused):
        logits = self._decode([text_hiddens, table_hiddens, img_hiddens],
                              [text_attention_mask, table_attention_mask, img_attention_mask], rating_diff, labels,
                              decoder_input_ids, decoder_attention_mask)
        return (logits,)


class BartForEncConditionalGeneration(_BartBase):
    multimodal = False

    def generate(self, encoder_hiddens, attention_mask=None, input_ids=None, rating_diff=None, **kw):
        if attention_mask is None:
            attention_mask = torch.ones(encoder_hiddens.shape[:3], dtype=torch.bool, device=encoder_hiddens.device)
        return self._generate([encoder_hiddens], [attention_mask], rating_diff, input_ids=input_ids, **kw)

    def forward(self, encoder_hiddens, rating_diff=None, encoder_attention_mask=None, decoder_input_ids=None,
                decoder_attention_mask=None, decoder_past_key_values=None, labels=None, use_cache=None, output_attentions=False,
                output_hidden_states=False, return_dict=False, **unused):
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(encoder_hiddens.shape[:3], dtype=torch.bool, device=encoder_hiddens.device)
        logits = self._decode([encoder_hiddens], [encoder_attention_mask], rating_diff, labels, decoder_input_ids,
                              decoder_attention_mask)
        return (logits,)


class YelpTableEncoder(nn.Module):
    """TableEncoder(bart_model.model.shared): must alias the embedding Parameter (multimodal_train.py:117)."""
    kind = "yelp"

    def __init__(self, bart_embedding, engine=None):
        super().__init__()
        if engine is None:
            raise RuntimeError("the table encoder needs the engine that owns the aliased embedding (build it through MultimodalSum)")
        if engine.table_kind != self.kind:
            raise RuntimeError("engine was built for the %s table encoder, not %s" % (engine.table_kind, self.kind))
        object.__setattr__(self, "_engine", engine)
        emb = _Node()
        emb.register_parameter("weight", bart_embedding.weight if hasattr(bart_embedding, "weight") else bart_embedding)
        self.add_module("bart_embedding", emb)
        from .engine import table_specs
        _attach_ordered(self, engine, "table_encoder.", [n[len("table_encoder."):] for n, _ in table_specs(kind=self.kind)])

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def forward(self, field, field_value):
        e = self._engine
        _new_forward(e)
        y, mask = _TableFn.apply(_anchor(e), e, field, *field_value)
        return y, mask.bool()


class AmazonTableEncoder(YelpTableEncoder):
    """AmazonTableEncoder(bart_model.model.shared) (table_encoder.py:86-167): field [6,1], field_value = [price [B,11],
    rating [B,4], brand [B,12], name [B,32], category [B,3,8,12], description [B,128]] -> ([B,133,D], [B,133] bool).
    Needs an engine built for it: MultimodalSum(..., TableEncoder=AmazonTableEncoder)."""
    kind = "amazon"


class Resnet(nn.Module):
    def __init__(self, embedding_dim, model="resnet101", stage=3, engine=None):
        super().__init__()
        assert model == "resnet101" and stage == 3, "the hot path uses resnet101 stages 1-3 (img_encoder.py:5,21-26)"
        if engine is None:
            raise RuntimeError("Resnet needs an engine (build it through MultimodalSum)")
        object.__setattr__(self, "_engine", engine)
        pre = "img_encoder."
        _attach_ordered(self, engine, pre, resnet_reference_names())
        for name, b in engine.buffers.items():
            if name.startswith(pre):
                _attach(self, name[len(pre):], b, is_buffer=True)
        r = self.resnet
        r.add_module("relu", nn.ReLU(inplace=True))
        r.add_module("maxpool", nn.MaxPool2d(3, 2, 1))
        for li in (1, 2, 3, 4):
            layer = getattr(r, "layer%d" % li)
            seq = nn.Sequential(*[getattr(layer, str(i)) for i in range(len(layer._modules))])
            r._modules["layer%d" % li] = seq
        # aliased views of the same modules, as the reference registers them (img_encoder.py:21-24)
        self.stage1 = nn.Sequential(r.conv1, r.bn1, r.relu, r.maxpool, r.layer1)
        self.stage2 = nn.Sequential(r.layer2)
        self.stage3 = nn.Sequential(r.layer3)

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(complete_aliases(self, state_dict), strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def forward(self, x):
        e = self._engine
        _new_forward(e)
        return _ImgFn.apply(_anchor(e), e, x)


# ------------------------------------------------------------------------------------------------
# step wrappers
# ------------------------------------------------------------------------------------------------
class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, batch):
        ctx.model = model
        graphs = getattr(model, "_step_graphs", None)
        ctx.entry = graphs.forward(batch) if (graphs is not None and model._engine.training) else None
        if ctx.entry is not None:
            ctx.saved, ctx.serial = None, ctx.entry.serial
            return ctx.entry.saved.loss.reshape(()).clone()      # the graph's loss buffer is overwritten by the next replay
        ctx.saved = model._step_fwd(*batch, compact=_compact(model))
        return ctx.saved.loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        # The upstream gradient (1 for loss.backward() as multimodal_train.py:360 calls it; anything else for loss / k,
        # weighted sums of losses, loss scaling) stays on the device: the LM-head backward products multiply by it.
        up = dloss.detach().reshape(1).to(torch.float32)
        if ctx.entry is not None:
            ctx.model._step_graphs.backward(ctx.entry, _begin_backward, _end_backward, ctx.serial, upstream=up)
        else:
            ctx.model._step_bwd(ctx.saved, upstream=up)
        return None, None, None


def _compact(model):
    """Whether the fused step runs its text-encoder layers and cross-attention K/V projections on the valid rows only
    (training steps; `model.compact_encoder = False` switches it off)."""
    return bool(getattr(model, "compact_encoder", True)) and model._engine.training


def _layer_chunks(L, per):
    """[(lo, hi)] from the top layer down, `per` layers each (the last one takes what is left); one chunk when per is falsy."""
    if not per or per >= L:
        return [(0, L)]
    out, hi = [], L
    while hi > 0:
        lo = max(0, hi - per)
        out.append((lo, hi))
        hi = lo
    return out


def _segment_layers(model):
    """Layers per gradient segment of the fused step's backward (`model.grad_segment_layers`; default 3 under a data-parallel wrapper,
    otherwise 0 = one segment per stack).  A segment is one captured graph and one notification to the data-parallel wrapper: with three layers per segment
    (~150 MB of f32 gradients) the exchange of a segment hides under the next one and only the last -- the encoder's bottom
    layers and the tied embedding, which the encoder's embedding backward finalises -- is exchanged in the open."""
    v = getattr(model, "grad_segment_layers", None)
    if v is None:
        # nobody listens for finished segments (no data-parallel wrapper): one segment per stack -- every captured graph costs a launch
        # (40 - 160 us between graphs in the kernel trace), and the finer split only exists to overlap the gradient exchange
        v = 3 if model._engine.segment_hooks else 0
    return int(v)


def _decoder_segments(e, s, st, release, upstream, per):
    """(callable, finished-parameter prefixes) of the LM head + decoder backward; leaves st["dmem"]."""
    dec = e.bp + "model.decoder."
    chunks = _layer_chunks(e.cfg.decoder_layers, per)
    segs = []
    for ci, (lo, hi) in enumerate(chunks):
        def run(lo=lo, hi=hi, top=(ci == 0)):
            if top:
                st["dh"] = e.lm_head_bwd(s.hL, s.dlogits, upstream=upstream)
                if release:
                    s.dlogits = None
            r = e.decoder_bwd(s.dec, st.get("dh"), split=(lo, hi, st.get("dec_carry")))
            st["dh"] = None
            if lo > 0:
                st["dec_carry"] = r
            else:
                st["dec_carry"], st["dmem"] = None, r
        prefixes = [dec + "layers.%d." % i for i in range(lo, hi)]
        if lo == 0:
            prefixes += [dec + "embed_positions", dec + "layernorm_embedding", dec + "rating_embeddings"]
        segs.append((run, prefixes))
    return segs


def _run_segments(engine, segments):
    _begin_backward(engine)
    for fn, prefixes in segments:
        fn()
        _end_backward(engine)
        engine.segment_ready(prefixes)


class _StepGraphMixin:
    """enable_step_graphs(): replay the fused step from captured HIP graphs (graphs.StepGraphs)."""

    def enable_step_graphs(self, enabled=True, max_live=2):
        """max_live: captured graph sets kept at once (one per distinct input SHAPE; token and image counts do not matter:
        they are device-side row counts the kernels read when they run)."""
        from .graphs import StepGraphs
        object.__setattr__(self, "_step_graphs", StepGraphs(self, max_live) if enabled else None)
        return self


class MultimodalSum(_StepGraphMixin, nn.Module):
    """MultimodalSum(bart_pretrained, table_pretrained, img_pretrained, TableEncoder)  (multimodal_train.py:111-122).

    forward(reviews, reviews_mask, reviews_rating, field, field_value, img, img_mask) -> (loss,)
    runs the whole leave-one-out step as one fused schedule (see engine.py)."""

    def __init__(self, bart_pretrained=None, table_pretrained=None, img_pretrained=None, TableEncoder=YelpTableEncoder,
                 config="cfg/bart-large.json", label_smoothing=0.1, device="cuda", dtype=torch.bfloat16, deterministic=False):
        super().__init__()
        cfg = config if isinstance(config, BartConfig) else BartConfig.from_json_file(config)
        e = Engine(cfg, device=device, compute_dtype=dtype, multimodal=True, with_table=getattr(TableEncoder, "kind", "yelp"), with_img=True,
                   bart_prefix="bart_model.", deterministic=deterministic)
        object.__setattr__(self, "_engine", e)
        self.label_smoothing = label_smoothing
        self.bart_model = BartForMultiEncConditionalGeneration(cfg, engine=e, prefix="bart_model.")
        self.table_encoder = TableEncoder(self.bart_model.model.shared, engine=e)
        self.img_encoder = Resnet(cfg.d_model, engine=e)
        init_formula(self)
        # stage hand-off files (multimodal_train.py:116-122): BART through from_pretrained's lenient key check, the table and
        # image encoders strictly, exactly as the reference loads them; None = keep the formula init (tests, bench)
        if bart_pretrained is not None:
            load_pretrained(self.bart_model, bart_pretrained, BART_ALLOWED_MISSING)
        if table_pretrained is not None:
            load_pretrained(self.table_encoder, table_pretrained)
        if img_pretrained is not None:
            load_pretrained(self.img_encoder, img_pretrained)
        e.mark_weights_changed()

    def train(self, mode=True):
        super().train(mode)
        self._engine.training = mode
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(complete_aliases(self, state_dict), strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def forward(self, reviews, reviews_mask, reviews_rating, field, field_value, img, img_mask, **unused):
        e = self._engine
        _new_forward(e)
        loss = _StepFn.apply(_anchor(e), self, (reviews, reviews_mask, reviews_rating, field, field_value, img, img_mask))
        return (loss,)

    def get_multimodal_outputs(self, reviews, reviews_mask, field, field_value, img, img_mask):
        """Drop-in, un-fused variant (multimodal_train.py:165-193) built from the coarse modules."""
        B, NR, S = reviews.shape
        text_h = self.bart_model.model.encoder(input_ids=reviews.view(B * NR, S), attention_mask=reviews_mask.view(B * NR, S))[0]
        text_h = text_h.view(B, NR, S, -1)
        table_h, table_m = self.table_encoder(field, field_value)
        I = img.size(1)
        img_h = self.img_encoder(img.reshape(-1, 3, img.shape[-2], img.shape[-1])).reshape(B, I, -1, self.bart_model.config.d_model)
        img_m = img_mask.unsqueeze(-1).repeat(1, 1, img_h.size(2))
        return NR, text_h, reviews_mask, table_h.unsqueeze(1), table_m.unsqueeze(1), img_h, img_m

    # ---- fused step -------------------------------------------------------------------------------
    def _step_fwd(self, reviews, reviews_mask, reviews_rating, field, field_value, img, img_mask, compact=False):
        e, cfg = self._engine, self._engine.cfg
        B, NR, S = reviews.shape
        I = img.shape[1]
        D = cfg.d_model
        imgs = img.reshape(-1, 3, img.shape[-2], img.shape[-1])
        hw = ((imgs.shape[-2] + 6 - 7) // 2 + 1, (imgs.shape[-1] + 6 - 7) // 2 + 1)
        for _ in range(3):   # maxpool, layer2, layer3 each halve the resolution
            hw = ((hw[0] + 2 - 3) // 2 + 1, (hw[1] + 2 - 3) // 2 + 1)
        P = hw[0] * hw[1]
        s = type("Saved", (), {})()
        TP = e.table_positions
        s.layout = e.make_memory(B, [(NR, S), (1, TP), (I, P)])
        s.mem = e.empty(s.layout.rows, D)
        o1, o2 = s.layout.offs[1], s.layout.offs[2]
        # The table and image encoders are ~700 small, strictly sequential launches (3.6 % of the FLOPs): they run on a
        # second stream beside the text encoder's large GEMMs (a parallel branch of the captured forward graph) and
        # fill the bubbles those leave; all three write disjoint row ranges of the memory matrix.
        main = torch.cuda.current_stream() if s.mem.is_cuda else None
        side = e.side_stream() if main is not None else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _, s.tab = e.table_fwd(field, field_value, out=s.mem[o1:o2])
                _, s.img = e.img_fwd(imgs, out=s.mem[o2:], img_mask=img_mask.reshape(-1))
        _, s.enc = e.encoder_fwd(reviews.reshape(B * NR, S), reviews_mask.reshape(B * NR, S), out=s.mem[:o1], compact=compact)
        if side is not None:
            main.wait_stream(side)
        else:
            _, s.tab = e.table_fwd(field, field_value, out=s.mem[o1:o2])
            _, s.img = e.img_fwd(imgs, out=s.mem[o2:], img_mask=img_mask.reshape(-1))
        pads = [reviews_mask.eq(0).to(torch.uint8).contiguous(), (1 - s.tab.mask).view(B, 1, TP).contiguous(),
                img_mask.eq(0).to(torch.uint8).unsqueeze(-1).expand(B, I, P).contiguous()]
        dec_in = shift_tokens_right_batched(reviews, reviews[:1], cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id)
        dec_in = dec_in.reshape(B * NR, S)
        dec_pad = dec_in.eq(cfg.pad_token_id).to(torch.uint8).contiguous()
        r = reviews_rating.float()
        rating_diff = (r - (r.sum(dim=1, keepdim=True) - r) / (NR - 1)).reshape(-1).contiguous()     # multimodal_train.py:153-156
        hL, s.dec = e.decoder_fwd(dec_in, dec_pad, rating_diff, s.mem, s.layout, pads, NR, True, compact_mem=compact)
        s.hL = hL
        s.loss, s.seq_loss, s.dlogits = e.lm_loss_fwd(hL, reviews.reshape(B * NR, S), self.label_smoothing, B * NR)
        return s

    def _step_bwd_segments(self, s, release=True, upstream=None):
        """The backward schedule as (callable, finished-parameter prefixes) segments, in execution order.
        upstream: device f32 scalar multiplied into every gradient (the gradient arriving at the loss)."""
        e = self._engine
        o1, o2 = s.layout.offs[1], s.layout.offs[2]
        st = {}
        per = _segment_layers(self)
        segs = _decoder_segments(e, s, st, release, upstream, per)
        L = e.cfg.encoder_layers
        enc = e.bp + "model.encoder."
        for ci, (lo, hi) in enumerate(_layer_chunks(L, per)):
            def run(lo=lo, hi=hi, top=(ci == 0)):
                if not top:
                    st["dx"] = e.encoder_bwd(s.enc, None, split=(lo, hi, st["dx"]))
                    return
                # image + table backward (layer3 and the projections only: small, sequential kernels) beside the text encoder's
                main = torch.cuda.current_stream() if st["dmem"].is_cuda else None
                side = e.side_stream() if main is not None else None
                if side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        e.img_bwd(s.img, st["dmem"][o2:])
                        e.table_bwd(s.tab, st["dmem"][o1:o2])
                    st["dx"] = e.encoder_bwd(s.enc, st["dmem"][:o1], split=(lo, hi, None))
                    main.wait_stream(side)
                else:
                    e.img_bwd(s.img, st["dmem"][o2:])
                    e.table_bwd(s.tab, st["dmem"][o1:o2])
                    st["dx"] = e.encoder_bwd(s.enc, st["dmem"][:o1], split=(lo, hi, None))
            prefixes = [enc + "layers.%d." % i for i in range(lo, hi)]
            if ci == 0:
                prefixes = ["img_encoder.", "table_encoder."] + prefixes
            if lo == 0:       # the encoder's embedding backward is the last contribution to the tied embedding: final only here
                prefixes += [enc + "embed", enc + "layernorm_embedding", e.bp + "model.shared."]
            segs.append((run, prefixes))
        return segs

    def _step_bwd(self, s, upstream=None):
        _run_segments(self._engine, self._step_bwd_segments(s, upstream=upstream))


class TextSupervised(_StepGraphMixin, nn.Module):
    """TextSupervised (text_pretrain.py:66-113): text-only leave-one-out step, fused the same way."""

    def __init__(self, bart_pretrained=None, config="cfg/bart-large.json", label_smoothing=None, device="cuda",
                 dtype=torch.bfloat16, deterministic=False):
        super().__init__()
        cfg = config if isinstance(config, BartConfig) else BartConfig.from_json_file(config)
        e = Engine(cfg, device=device, compute_dtype=dtype, multimodal=False, bart_prefix="bart_model.", deterministic=deterministic)
        object.__setattr__(self, "_engine", e)
        self.label_smoothing = label_smoothing
        self.bart_model = BartForEncConditionalGeneration(cfg, engine=e, prefix="bart_model.")
        init_formula(self)
        if bart_pretrained is not None:
            load_pretrained(self.bart_model, bart_pretrained, BART_ALLOWED_MISSING)
        e.mark_weights_changed()

    def train(self, mode=True):
        super().train(mode)
        self._engine.training = mode
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(complete_aliases(self, state_dict), strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def forward(self, reviews, reviews_mask, reviews_rating, **unused):
        e = self._engine
        _new_forward(e)
        return (_StepFn.apply(_anchor(e), self, (reviews, reviews_mask, reviews_rating)),)

    def _step_fwd(self, reviews, reviews_mask, reviews_rating, compact=False):
        e, cfg = self._engine, self._engine.cfg
        B, NR, S = reviews.shape
        s = type("Saved", (), {})()
        s.layout = e.make_memory(B, [(NR, S)])
        s.mem = e.empty(s.layout.rows, cfg.d_model)
        _, s.enc = e.encoder_fwd(reviews.reshape(B * NR, S), reviews_mask.reshape(B * NR, S), out=s.mem, compact=compact)
        pads = [reviews_mask.eq(0).to(torch.uint8).contiguous()]
        dec_in = shift_tokens_right_batched(reviews, reviews[:1], cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id).reshape(B * NR, S)
        dec_pad = dec_in.eq(cfg.pad_token_id).to(torch.uint8).contiguous()
        r = reviews_rating.float()
        rating_diff = (r - (r.sum(dim=1, keepdim=True) - r) / (NR - 1)).reshape(-1).contiguous()
        s.hL, s.dec = e.decoder_fwd(dec_in, dec_pad, rating_diff, s.mem, s.layout, pads, NR, True, compact_mem=compact)
        s.loss, s.seq_loss, s.dlogits = e.lm_loss_fwd(s.hL, reviews.reshape(B * NR, S), self.label_smoothing, B * NR)
        return s

    def _step_bwd_segments(self, s, release=True, upstream=None):
        e = self._engine
        st = {}
        per = _segment_layers(self)
        segs = _decoder_segments(e, s, st, release, upstream, per)
        enc = e.bp + "model.encoder."
        for ci, (lo, hi) in enumerate(_layer_chunks(e.cfg.encoder_layers, per)):
            def run(lo=lo, hi=hi, top=(ci == 0)):
                st["dx"] = e.encoder_bwd(s.enc, st["dmem"] if top else None, split=(lo, hi, None if top else st["dx"]))
            prefixes = [enc + "layers.%d." % i for i in range(lo, hi)]
            if lo == 0:
                prefixes += [enc + "embed", enc + "layernorm_embedding", e.bp + "model.shared."]
            segs.append((run, prefixes))
        return segs

    def _step_bwd(self, s, upstream=None):
        _run_segments(self._engine, self._step_bwd_segments(s, upstream=upstream))


class _LossFn(torch.autograd.Function):
    """LabelSmoothingLoss.forward (utils.py:32-38) on a logits tensor, fused HIP kernel (forward + gradient)."""

    @staticmethod
    def forward(ctx, pred, target, classes, smoothing):
        R = pred.shape[0]
        work = pred.detach().to(torch.float32).contiguous().clone()
        rows = torch.empty(R, dtype=torch.float32, device=pred.device)
        kn.ls_loss(work, target.reshape(-1).contiguous(), rows, classes, float(smoothing), 1.0 / R, True)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        kn.segment_sum(rows, loss, 1, R, 1.0 / R)
        ctx.save_for_backward(work)
        ctx.in_dtype = pred.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (work,) = ctx.saved_tensors
        return (work * g).to(ctx.in_dtype), None, None, None


class LabelSmoothingLoss(nn.Module):
    """Drop-in for /root/reference/src/utils.py:24-38 (pads are not ignored, quirk Q2)."""

    def __init__(self, classes, smoothing=0.0, dim=-1):
        super().__init__()
        self.cls, self.smoothing = classes, smoothing

    def forward(self, pred, target):
        return _LossFn.apply(pred, target, self.cls, self.smoothing)


class _SingleModality(nn.Module):
    """Step-2 pretraining wrappers: one modality encoder feeds the text-only decoder
    through the unimodal branch of the cross-attention (modeling_multimodalsum.py:746-749)."""

    def __init__(self, bart_pretrained, config, label_smoothing, device, dtype, deterministic, with_img, with_table):
        super().__init__()
        cfg = config if isinstance(config, BartConfig) else BartConfig.from_json_file(config)
        e = Engine(cfg, device=device, compute_dtype=dtype, multimodal=False, with_table=with_table, with_img=with_img,
                   bart_prefix="bart_model.", deterministic=deterministic)
        object.__setattr__(self, "_engine", e)
        self.label_smoothing = label_smoothing
        self.bart_model = BartForEncConditionalGeneration(cfg, engine=e, prefix="bart_model.")
        if with_img:
            self.img_encoder = Resnet(cfg.d_model, engine=e)
        if with_table:
            self.table_encoder = (AmazonTableEncoder if with_table == "amazon" else YelpTableEncoder)(self.bart_model.model.shared, engine=e)
        init_formula(self)
        if bart_pretrained is not None:
            load_pretrained(self.bart_model, bart_pretrained, BART_ALLOWED_MISSING)
        e.mark_weights_changed()

    def train(self, mode=True):
        super().train(mode)
        self._engine.training = mode
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(complete_aliases(self, state_dict), strict=strict, **kw)
        self._engine.mark_weights_changed()
        return r

    def _loss(self, hiddens, mask, labels):
        bsz = hiddens.shape[0]
        rating_diff = torch.zeros(bsz, 1, device=hiddens.device)
        logits = self.bart_model(hiddens, rating_diff, mask, labels=labels)[0]
        V = self.bart_model.config.vocab_size
        return (LabelSmoothingLoss(V, self.label_smoothing or 0.0)(logits.view(-1, V), labels.reshape(-1)),)


class ImgSupervised(_SingleModality):
    """img_pretrain.py:85-141: forward(input_imgs [B,I,3,224,224], input_imgs_mask [B,I], labels=[B,T]) -> (loss,)."""

    def __init__(self, bart_pretrained=None, config="cfg/bart-large.json", label_smoothing=0.1, device="cuda", dtype=torch.bfloat16,
                 deterministic=False):
        super().__init__(bart_pretrained, config, label_smoothing, device, dtype, deterministic, True, False)

    def forward(self, input_imgs, input_imgs_mask=None, labels=None, **unused):
        bsz, n = input_imgs.shape[:2]
        h = self.img_encoder(input_imgs.reshape(-1, 3, input_imgs.shape[-2], input_imgs.shape[-1]))
        h = h.reshape(bsz, n, -1, self.bart_model.config.d_model)
        mask = None if input_imgs_mask is None else input_imgs_mask.unsqueeze(-1).repeat(1, 1, h.shape[2])
        return self._loss(h, mask, labels)


class TableSupervised(_SingleModality):
    """table_pretrain.py:84-129: forward(field, field_value, labels=[B,T]) -> (loss,)."""

    def __init__(self, bart_pretrained=None, config="cfg/bart-large.json", label_smoothing=0.1, device="cuda", dtype=torch.bfloat16,
                 deterministic=False, TableEncoder=None):
        super().__init__(bart_pretrained, config, label_smoothing, device, dtype, deterministic, False,
                         getattr(TableEncoder, "kind", "yelp") if TableEncoder is not None else True)

    def forward(self, field, field_value, labels=None, **unused):
        h, m = self.table_encoder(field, field_value)
        return self._loss(h.unsqueeze(1), m.unsqueeze(1), labels)


def init_formula(module, std=0.02, prefix=None):
    """Fill every arena parameter / buffer with the RNG-free formula init (keyed by arena names)."""
    e = module._engine
    pre = prefix
    names = {n: e.arena.shapes[n] for n in e.arena.params}
    # the closed form is integer hashing + f64 arithmetic: evaluated ON THE DEVICE it gives bit-identical values (exact integer ops,
    # IEEE f64 multiply / add) in milliseconds -- on the host it is ~50 s for the 505 M parameters, and `bench.py --gpus 8` would run
    # it eight times on the node's cores although rank 0 broadcasts its parameters anyway
    dev = e.device
    if pre is not None:
        sd = formula_state_dict({pre + n[len(e.bp):] if n.startswith(e.bp) else n: s for n, s in names.items()}, std=std, device=dev)
        sd = {(e.bp + k[len(pre):]) if k.startswith(pre) else k: v for k, v in sd.items()}
    else:
        sd = formula_state_dict(names, std=std, device=dev)
    with torch.no_grad():
        for n, v in sd.items():
            e.arena.params[n].copy_(v.to(e.device))
        for n, b in e.buffers.items():
            if n.endswith("running_var"):
                b.fill_(1.0)
            elif n.endswith("running_mean") or n.endswith("final_logits_bias"):
                b.zero_()
    e.mark_weights_changed()
