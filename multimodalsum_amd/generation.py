"""Beam-search generation on the HIP decoder kernels.

Replaces BartForMultiEncConditionalGeneration.generate / _generate_beam_search and the text-only
BartForEncConditionalGeneration.generate (/root/reference/src/transformer/modeling_multimodalsum.py:2295-2693,
2803-3067, 1398-1700; score post-processing /root/reference/src/transformer/generation_utils.py:57-98,848-868;
BeamHypotheses :948-993) as called by /root/reference/src/test.py:153-158 (greedy beam search, do_sample=False).

MI355X-first differences from the reference (token ids identical):
* the encoder tensors are NOT expanded num_beams times and NOT re-gathered every step (:2599-2627, :2996-3010): all
  hypotheses of a business read the same memory rows; the entity-attention kernel maps hypothesis row -> business
  with its `qpb` (= num_beams) argument, exactly as the training step maps leave-one-out passes;
* cross-attention K/V of every layer are projected once per call for the un-expanded memory (reference: once, but
  for the num_beams-times expanded tensors), self-attention K/V live in per-layer caches [rows, max_length, D] that a
  beam reorder gathers in place of the reference's list-of-dict `_reorder_cache`;
* one decode step = embed+LN (position = current length - 1), per layer fused qkv GEMM -> cache append -> entity
  attention over the cache (keys beyond the current length masked) -> out_proj -> add+LN, the per-entity
  cross-attention + gate, the FFN, then the tied LM head.
The hypothesis bookkeeping stays on the host like the reference's (one device->host transfer of the 2*num_beams
candidates per step); log-softmax / ban / top-k use torch ops on the [rows, V] logits (fusing them into one kernel is
the remaining section-8f item).
"""
import torch

from . import kernels as kn


class _Hypotheses:
    """n-best finished hypotheses of one batch entry (generation_utils.py:948-993)."""

    def __init__(self, num_beams, max_length, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.items, self.worst = [], 1e9

    def add(self, tokens, sum_logprobs):
        score = sum_logprobs / len(tokens) ** self.length_penalty
        if len(self.items) < self.num_beams or score > self.worst:
            self.items.append((score, tokens))
            if len(self.items) > self.num_beams:
                order = sorted((s, i) for i, (s, _) in enumerate(self.items))
                del self.items[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.items) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst >= best_sum_logprobs / cur_len ** self.length_penalty


def _banned_ngram_tokens(rows, n, cur_len):
    if cur_len + 1 < n:
        return [[] for _ in rows]
    out = []
    for toks in rows:
        seen = {}
        for i in range(len(toks) - n + 1):
            seen.setdefault(tuple(toks[i:i + n - 1]), []).append(toks[i + n - 1])
        out.append(seen.get(tuple(toks[cur_len + 1 - n:cur_len]), []))
    return out


class DecodeSession:
    """KV-cached single-token decoder over an engine's weights; rows = B * rows_per_business hypotheses.

    Every buffer is static (memory matrix, masks, cross K/V of all layers, two ping-pong sets of self-attention
    caches, token / beam-index inputs, logits), so a decode step at position t is the same launch sequence on the
    same addresses in every generate() call: on the GPU it is captured once per position into a HIP graph
    (first use) and replayed afterwards -- a step is ~250 launches of a few microseconds each, i.e. purely
    launch-bound when issued from Python.  Sessions are cached on the engine per (memory layout, beams, max_length)."""

    def __init__(self, engine, layout, rows_per_business, max_length, has_rating):
        e, cfg = engine, engine.cfg
        if max_length > 224:
            raise ValueError("max_length > 224 exceeds the attention kernel's key tile")
        self.e, self.L, self.qpb, self.Tmax = e, layout, rows_per_business, max_length
        D, R = cfg.d_model, layout.B * rows_per_business
        self.rows = R
        dev = e.device
        self.mem = e.empty(layout.rows, D)
        self.pads = [torch.empty(layout.B, N, S, dtype=torch.uint8, device=dev) for (N, S) in layout.mods]
        self.nulls = [e.empty(layout.B * N, dtype=torch.uint8) for (N, S) in layout.mods]
        self.no_table = self.nulls[1] if e.multimodal else None
        self.no_img = e.empty(layout.B, dtype=torch.uint8) if e.multimodal else None
        self.rd = e.empty(R, dtype=torch.float32) if has_rating else None
        self.kv = [e.empty(layout.rows, 2 * D) for _ in range(cfg.decoder_layers)]
        # self-attention caches, two sets: step t reads/extends set t & 1 after gathering it from set (t-1) & 1 by beam index;
        # zero-filled once (masked keys carry probability 0, so whatever they hold must stay finite)
        self.kc = [[e.zeros(R * max_length, D) for _ in range(cfg.decoder_layers)] for _ in range(2)]
        self.vc = [[e.zeros(R * max_length, D) for _ in range(cfg.decoder_layers)] for _ in range(2)]
        self.self_pad = torch.ones(R, max_length, dtype=torch.uint8, device=dev)
        self.mean, self.rstd = e.empty(R, dtype=torch.float32), e.empty(R, dtype=torch.float32)
        self.tokens = torch.zeros(R, 1, dtype=torch.long, device=dev)
        self.beam_idx = torch.arange(R, dtype=torch.long, device=dev)
        self.logits = e.empty(R, e.Vpad)
        self.use_graphs = dev.type == "cuda" and __import__("os").environ.get("MMSUM_DECODE_GRAPHS") != "0"
        self.graphs, self.pool, self.warm = {}, None, False

    def begin(self, hiddens, pads, rating_diff):
        """New generate() call: load the memory, masks and rating difference, project the cross-attention K/V."""
        e, cfg, a, L = self.e, self.e.cfg, self.e.arena, self.L
        D = cfg.d_model
        for m, h in enumerate(hiddens):
            n = h.shape[0] * h.shape[1] * h.shape[2]
            self.mem[L.offs[m]:L.offs[m] + n].copy_(h.reshape(n, D))
        for dst, src in zip(self.pads, pads):
            dst.copy_(src.reshape(dst.shape))
        if self.rd is not None:
            self.rd.copy_(rating_diff.reshape(L.B, 1).float().repeat_interleave(self.qpb, dim=0).reshape(-1))
        for (N, S), pad, nul in zip(L.mods, self.pads, self.nulls):
            kn.entity_null(pad, nul, L.B * N, S)
        if e.multimodal:
            N2, S2 = L.mods[2]
            kn.entity_null(self.pads[2], self.no_img, L.B, N2 * S2)
        b = e.bp + "model.decoder."
        for i in range(cfg.decoder_layers):                     # (:810-815 caches them after the first step)
            _, k, v = e._attn_names(b + "layers.%d." % i, "encoder_attn")
            kn.gemm(self.mem, a.wspan(k + ".weight", v + ".weight", (2 * D, D)), self.kv[i],
                    bias=a.span(a.data, k + ".bias", v + ".bias", (2 * D,)))
        self.self_pad.fill_(1)

    def step(self, tokens, beam_idx, t):
        """tokens [rows] int64 = token at position t of every hypothesis, which continues hypothesis beam_idx[row] of the
        previous step (None at t = 0).  -> next-token logits [rows, V] f32."""
        self.tokens.copy_(tokens.view(-1, 1))
        if beam_idx is not None:
            self.beam_idx.copy_(beam_idx)
        if not self.use_graphs:
            self._step(t)
        else:
            if not self.warm:                                    # one eager pass first: lazy kernel attributes, allocator warm-up
                self._step(t)
                self.warm = True
            g = self.graphs.get(t)
            if g is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool):
                    self._step(t)
                if self.pool is None:
                    self.pool = g.pool()
                self.graphs[t] = g
            g.replay()
        return self.logits[:, :self.e.cfg.vocab_size].float()

    def _step(self, t):
        e, cfg, a = self.e, self.e.cfg, self.e.arena
        D, H, R, Tm = cfg.d_model, cfg.heads, self.rows, self.Tmax
        b = e.bp + "model.decoder."
        scale = 64 ** -0.5
        cur, prev = t & 1, (t - 1) & 1
        kc, vc = self.kc[cur], self.vc[cur]
        if t > 0:                                                # beam reorder (:2996-3003, _reorder_cache) = gather into the other cache set
            for i in range(cfg.decoder_layers):
                torch.index_select(self.kc[prev][i].view(R, Tm, D), 0, self.beam_idx, out=kc[i].view(R, Tm, D))
                torch.index_select(self.vc[prev][i].view(R, Tm, D), 0, self.beam_idx, out=vc[i].view(R, Tm, D))
        x = e.empty(R, D)
        kn.embed_ln_fwd(self.tokens, a.w(e.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), self.rd,
                        a.w(b + "rating_embeddings") if self.rd is not None else None, a.f32(b + "layernorm_embedding.weight"),
                        a.f32(b + "layernorm_embedding.bias"), x, self.mean, self.rstd, R, 1, cfg.extra_pos_embeddings + t, 1e-5, 0.0, 0)
        self.self_pad[:, t] = 0
        nm = len(self.L.mods)
        for i in range(cfg.decoder_layers):
            lb = b + "layers.%d." % i
            # ---- self-attention over the cache (:442-461 with layer_state, :776-815)
            q, k, v = e._attn_names(lb, "self_attn")
            qkv = e.empty(R, 3 * D)
            kn.gemm(x, a.wspan(q + ".weight", v + ".weight", (3 * D, D)), qkv, bias=a.span(a.data, q + ".bias", v + ".bias", (3 * D,)))
            kc[i].view(R, Tm, D)[:, t].copy_(qkv[:, D:2 * D])
            vc[i].view(R, Tm, D)[:, t].copy_(qkv[:, 2 * D:])
            att = e.empty(R, D)
            d = kn.make_attn_desc(qkv[:, :D], kc[i], vc[i], att, self.self_pad, None, R, 1, 1, 1, Tm, H, False, False, scale)
            kn.attn_fwd(d, x)
            o = e.empty(R, D)
            kn.gemm(att, a.w(lb + "self_attn.out_proj.weight"), o, bias=a.f32(lb + "self_attn.out_proj.bias"))
            y = e.empty(R, D)
            kn.add_ln_fwd(o, x, a.f32(lb + "self_attn_layer_norm.weight"), a.f32(lb + "self_attn_layer_norm.bias"), y, self.mean, self.rstd,
                          1e-5, 0.0, 0)
            x = y
            # ---- per-entity cross-attention + entity mean (+ gate)  (:711-750, :819-886)
            q, _, _ = e._attn_names(lb, "encoder_attn")
            pre = lb + "encoder_attn."
            cq = e.empty(R, D)
            kn.gemm(x, a.w(q + ".weight"), cq, bias=a.f32(q + ".bias"))
            heads = e.empty(nm * R, D)
            for m, ((N, S), pad) in enumerate(zip(self.L.mods, self.pads)):
                rows = slice(self.L.offs[m], self.L.offs[m] + self.L.B * N * S)
                # the hypotheses of a business are consecutive rows: one query block of `qpb` rows per business reads the
                # business's K/V once for all of them
                d = kn.make_attn_desc(cq, self.kv[i][rows, :D], self.kv[i][rows, D:], heads[m * R:(m + 1) * R], pad, self.nulls[m],
                                      self.L.B, self.qpb, 1, N, S, H, False, False, scale)
                kn.attn_fwd(d, x)
            yy = e.empty(nm * R, D)
            kn.gemm(heads, a.w(pre + "out_proj.weight"), yy, bias=a.f32(pre + "out_proj.bias"))
            if e.multimodal:
                yt, ytab, yimg = yy[:R], yy[R:2 * R], yy[2 * R:]
                pa, pb = e.empty(R, D), e.empty(R, D)
                kn.gemm(yt, a.w(pre + "alpha_proj.weight"), pa, a2=ytab, bias=a.f32(pre + "alpha_proj.bias"))
                kn.gemm(yt, a.w(pre + "beta_proj.weight"), pb, a2=yimg, bias=a.f32(pre + "beta_proj.bias"))
                c = e.empty(R, D)
                kn.gate_fwd(pa, pb, yt, ytab, yimg, self.no_table, self.no_img, c, self.qpb)
            else:
                c = yy
            y = e.empty(R, D)
            kn.add_ln_fwd(c, x, a.f32(lb + "encoder_attn_layer_norm.weight"), a.f32(lb + "encoder_attn_layer_norm.bias"), y, self.mean,
                          self.rstd, 1e-5, 0.0, 0)
            x = y
            # ---- FFN (:479-489)
            Fd = a.shapes[lb + "fc1.weight"][0]
            h = e.empty(R, Fd)
            kn.gemm(x, a.w(lb + "fc1.weight"), h, bias=a.f32(lb + "fc1.bias"), epi=kn.EPI_GELU)
            f = e.empty(R, D)
            kn.gemm(h, a.w(lb + "fc2.weight"), f, bias=a.f32(lb + "fc2.bias"))
            y = e.empty(R, D)
            kn.add_ln_fwd(f, x, a.f32(lb + "final_layer_norm.weight"), a.f32(lb + "final_layer_norm.bias"), y, self.mean, self.rstd, 1e-5,
                          0.0, 0)
            x = y
        kn.gemm(x, a.w(e.bp + "model.shared.weight"), self.logits[:, :cfg.vocab_size],
                bias=e.buffers[e.bp + "final_logits_bias"].reshape(-1))                                      # :2281


def _session(engine, layout, num_beams, max_length, has_rating):
    key = (tuple(layout.mods), layout.B, num_beams, max_length, has_rating)
    cache = engine.__dict__.setdefault("_decode_sessions", {})
    if key not in cache:
        if len(cache) >= 4:                          # static buffers + graphs per shape: keep only a few
            cache.pop(next(iter(cache)))
        cache[key] = DecodeSession(engine, layout, num_beams, max_length, has_rating)
    return cache[key]


@torch.no_grad()
def beam_search(engine, hiddens, layout, pads, rating_diff, num_beams, max_length, min_length, no_repeat_ngram_size, early_stopping,
                length_penalty, decoder_start_token_id):
    """Greedy beam search (_generate_beam_search :2803-3067).  Returns LongTensor [B, L] on the engine's device."""
    cfg = engine.cfg
    pad, bos, eos, V = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id, cfg.vocab_size
    dev = engine.device
    B = layout.B
    R = B * num_beams
    sess = _session(engine, layout, num_beams, max_length, rating_diff is not None)
    sess.begin(hiddens, pads, rating_diff)
    beam_idx_dev = None
    rows = [[decoder_start_token_id] for _ in range(R)]                    # host copy of input_ids
    last = torch.full((R,), decoder_start_token_id, dtype=torch.long, device=dev)
    hyps = [_Hypotheses(num_beams, max_length, length_penalty, early_stopping) for _ in range(B)]
    beam_scores = torch.zeros(B, num_beams, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    host_scores = beam_scores.tolist()
    done = [False] * B
    cur_len = 1
    neg_inf = float("-inf")
    while cur_len < max_length:
        logits = sess.step(last, beam_idx_dev, cur_len - 1)
        if cur_len == 1:                                                   # force BOS (:3084-3086)
            keep = logits[:, bos].clone()
            logits.fill_(neg_inf)
            logits[:, bos] = keep
        if cur_len == max_length - 1 and eos is not None:                  # force EOS (:3087-3088)
            keep = logits[:, eos].clone()
            logits.fill_(neg_inf)
            logits[:, eos] = keep
        scores = torch.log_softmax(logits, dim=-1)
        if eos is not None and cur_len < min_length:
            scores[:, eos] = neg_inf
        if no_repeat_ngram_size > 0:
            banned = _banned_ngram_tokens(rows, no_repeat_ngram_size, cur_len)
            ri = [i for i, bt in enumerate(banned) for _ in bt]
            if ri:
                ci = [tk for bt in banned for tk in bt]
                scores[torch.tensor(ri, device=dev), torch.tensor(ci, device=dev)] = neg_inf
        cand = (scores + beam_scores[:, None]).view(B, num_beams * V)
        top_s, top_i = torch.topk(cand, 2 * num_beams, dim=1, largest=True, sorted=True)
        top_s, top_i = top_s.tolist(), top_i.tolist()                      # the step's one device->host transfer
        nxt = []
        for b in range(B):
            if done[b]:
                nxt.extend([(0.0, pad, 0)] * num_beams)
                continue
            sent = []
            for rank, (tok_id, sc) in enumerate(zip(top_i[b], top_s[b])):
                beam, tok = tok_id // V, tok_id % V
                row = b * num_beams + beam
                if eos is not None and tok == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(list(rows[row]), sc)
                else:
                    sent.append((sc, tok, row))
                if len(sent) == num_beams:
                    break
            done[b] = done[b] or hyps[b].is_done(max(top_s[b]), cur_len)
            assert len(sent) == num_beams, "Beam should always be full"
            nxt.extend(sent)
        if all(done):
            break
        host_scores = [x[0] for x in nxt]
        beam_scores = torch.tensor(host_scores, dtype=torch.float32, device=dev)
        beam_idx = [x[2] for x in nxt]
        rows = [rows[j] + [x[1]] for j, x in zip(beam_idx, nxt)]
        last = torch.tensor([x[1] for x in nxt], dtype=torch.long, device=dev)
        beam_idx_dev = torch.tensor(beam_idx, dtype=torch.long, device=dev)
        cur_len += 1
    for b in range(B):
        if done[b]:
            continue
        for beam in range(num_beams):
            row = b * num_beams + beam
            hyps[b].add(list(rows[row]), host_scores[row])
    best = [sorted(h.items, key=lambda x: x[0])[-1][1] for h in hyps]
    lens = [len(t) for t in best]
    if min(lens) != max(lens):
        L = min(max(lens) + 1, max_length)
        out = torch.full((B, L), pad, dtype=torch.long)
        for i, t in enumerate(best):
            out[i, :lens[i]] = torch.tensor(t, dtype=torch.long)
            if lens[i] < max_length:
                out[i, lens[i]] = eos
    else:
        out = torch.tensor(best, dtype=torch.long)
    return out.to(dev)
