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
* the step's tail -- forced BOS/EOS, log-softmax, min-length and n-gram bans, + beam scores, top 2*num_beams of every
  business -- is mmsum_beam_topk (two launches over the [rows, V] logits; the reference materialises four [rows, V] f32
  tensors per step), and the beam reorder is a gather of a [rows, max_length] ancestor table that
  mmsum_decode_self_attn reads the caches through (the reference index_selects every layer's K and V cache).
The hypothesis bookkeeping stays on the host like the reference's: per step one small host->device block (tokens, parents,
banned n-gram tokens, beam scores) and one device->host read of the 2*num_beams candidates per business.
"""
import numpy as np
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


def _banned_ngram_table(tokens, n, cur_len, out):
    """calc_banned_ngram_tokens (generation_utils.py:57-98) for every hypothesis at once: `tokens` [rows, >= cur_len] (numpy, the
    first cur_len columns are the hypotheses so far); a next token t is banned for a row when the row's last n-1 tokens followed
    by t already occur in it.  Fills `out` [rows, nban] (int32, -1 padded) and returns it."""
    out.fill(-1)
    if cur_len + 1 < n:
        return out
    seq = tokens[:, :cur_len]
    npos = cur_len - n + 1                                   # start positions of complete n-grams
    if npos <= 0:
        return out
    match = np.ones((seq.shape[0], npos), dtype=bool)
    for k in range(n - 1):                                   # the (n-1)-gram starting at i equals the row's last n-1 tokens
        match &= seq[:, k:k + npos] == seq[:, cur_len - (n - 1) + k][:, None]
    r, c = np.nonzero(match)
    if r.size:
        rank = (np.cumsum(match, axis=1) - 1)[r, c]
        out[r, rank] = seq[r, c + n - 1]
    return out


def _bad_word_bans(tokens, cur_len, bad_words, out, col0):
    """calc_banned_bad_words_ids (generation_utils.py:871-904) for every hypothesis: the last token of a bad word is banned when the row
    ends with the word's other tokens (a one-token word always).  The reference's length guard compares the word with the NUMBER OF
    ROWS (`len(prev_input_ids)`), not with the row's length: kept.  Appends to the rows' ban lists of `out` starting at their first -1
    at or after column col0 (lists are filled from the front)."""
    rows = tokens.shape[0]
    fill = (out[:, :] >= 0).sum(axis=1)
    for seq in bad_words:
        head = seq[:-1]
        if len(head) == 0:
            hit = np.ones(rows, dtype=bool)
        elif len(head) > rows or len(head) > cur_len:
            # (a word longer than the rows so far cannot match: the reference's list comparison of unequal lengths is False)
            hit = np.zeros(rows, dtype=bool)
        else:
            hit = (tokens[:, cur_len - len(head):cur_len] == np.asarray(head, dtype=tokens.dtype)[None, :]).all(axis=1)
        r = np.nonzero(hit)[0]
        out[r, fill[r]] = seq[-1]
        fill[r] += 1
    return out


def _distinct_tokens(tokens, cur_len, out):
    """A row's distinct previous tokens, -1 padded (the set() of enforce_repetition_penalty_, generation_utils.py:47-55)."""
    out.fill(-1)
    for r in range(tokens.shape[0]):
        u = np.unique(tokens[r, :cur_len])
        out[r, :u.size] = u
    return out


class DecodeSession:
    """KV-cached single-token decoder over an engine's weights; rows = B * num_beams hypotheses.

    Every buffer is static (memory matrix, masks, cross K/V of all layers, self-attention caches + ancestor tables, the
    step's host-filled inputs, logits, candidate outputs), so a decode step at position t is the same launch sequence on
    the same addresses in every generate() call: on the GPU it is captured once per position into a HIP graph (first use)
    and replayed afterwards -- a step is ~250 launches of a few microseconds each, i.e. purely launch-bound when issued
    from Python.  Sessions are cached on the engine per (memory layout, beams, lengths, n-gram size).

    Per step the host sends ONE int32 block (token and parent hypothesis of every row, the banned n-gram tokens) and the
    beam scores, and reads back the 2*num_beams (score, index) candidates of every business: log-softmax, forced BOS/EOS,
    bans and the top-k run in mmsum_beam_topk, and the beam reorder is a gather of the [rows, max_length] ancestor table
    (mmsum_decode_self_attn reads the caches through it) instead of a gather of every layer's K/V cache."""

    def __init__(self, engine, layout, num_beams, max_length, has_rating, min_length=0, ngram=0, bad_words=None, penalty=1.0, greedy=False, ncand=0):
        e, cfg = engine, engine.cfg
        if max_length > 256:
            raise ValueError("max_length > 256 exceeds the decode self-attention kernel's cache walk (mmsum_decode_self_attn: Tmax <= 256)")
        if num_beams > 8:
            raise ValueError("num_beams > 8: the candidate kernel keeps 2*num_beams <= 16 entries per thread")
        self.e, self.L, self.qpb, self.Tmax = e, layout, num_beams, max_length
        self.min_length, self.ngram = min_length, ngram
        self.bad_words = [list(map(int, w)) for w in bad_words] if bad_words else []
        self.penalty, self.greedy = float(penalty), bool(greedy)
        # ncand > 0 (sampling): the step returns that many candidates of every row (its top_k best) and forces no token -- the reference
        # skips adjust_logits_during_generation when it samples (:1811)
        self.ncand = int(ncand)
        if self.ncand and (num_beams != 1 or not greedy or self.ncand > 64):
            raise ValueError("candidate lists (sampling) are built for one hypothesis per business and at most 64 candidates")
        D, R = cfg.d_model, layout.B * num_beams
        self.rows = R
        dev = e.device
        self.mem = e.empty(layout.rows, D)
        self.pads = [torch.empty(layout.B, N, S, dtype=torch.uint8, device=dev) for (N, S) in layout.mods]
        self.nulls = [e.empty(layout.B * N, dtype=torch.uint8) for (N, S) in layout.mods]
        self.no_table = self.nulls[1] if e.multimodal else None
        self.no_img = e.empty(layout.B, dtype=torch.uint8) if e.multimodal else None
        self.rd = e.empty(R, dtype=torch.float32) if has_rating else None
        self.kv = [e.empty(layout.rows, 2 * D) for _ in range(cfg.decoder_layers)]
        # self-attention caches [rows * max_length, D]: position t of physical row r is written once, by the hypothesis that sits
        # in row r at step t; later hypotheses that descend from it find it through the ancestor table (two tables, ping-pong)
        self.kc = [e.zeros(R * max_length, D) for _ in range(cfg.decoder_layers)]
        self.vc = [e.zeros(R * max_length, D) for _ in range(cfg.decoder_layers)]
        self.anc = [torch.zeros(R, max_length, dtype=torch.int32, device=dev) for _ in range(2)]
        self.arange = torch.arange(R, dtype=torch.int32, device=dev)
        self.mean, self.rstd = e.empty(R, dtype=torch.float32), e.empty(R, dtype=torch.float32)
        # host-filled inputs of a step: [tokens R | parent rows R | banned tokens R * nban] int32, beam scores f32
        self.nban = (max_length if ngram > 0 else 0) + len(self.bad_words)
        self.npen = max_length if self.penalty != 1.0 else 0       # the row's distinct previous tokens (repetition penalty)
        pin = dev.type == "cuda"
        self.h_int = torch.zeros(R * (2 + self.nban + self.npen), dtype=torch.int32, pin_memory=pin)
        self.h_sc = torch.zeros(R, dtype=torch.float32, pin_memory=pin)
        self.h_int_np, self.h_sc_np = self.h_int.numpy(), self.h_sc.numpy()       # host views the bookkeeping writes with numpy
        self.d_int = torch.zeros(R * (2 + self.nban + self.npen), dtype=torch.int32, device=dev)
        self.beam_scores = torch.zeros(R, dtype=torch.float32, device=dev)
        self.tokens = torch.zeros(R, 1, dtype=torch.long, device=dev)
        # the quantity that is ranked stays f32 in either compute mode: f32 logits [rows, Vpad] (6 MB), and in bf16 mode the last
        # LayerNorm also leaves its un-rounded f32 result for the LM head (mmsum_gemm's MMSUM_GEMM_A_F32 form: bf16 weights, the
        # f32 activations multiplied as bf16 hi + lo parts, f32 accumulation and f32 store)
        self.logits = e.empty(R, e.Vpad, dtype=torch.float32)
        self.x32 = e.empty(R, D, dtype=torch.float32) if (e.dtype == torch.bfloat16 and D % 128 == 0) else None
        K = self.ncand or 2 * num_beams
        self.out_scores = torch.zeros(layout.B, K, dtype=torch.float32, device=dev)
        self.out_ids = torch.zeros(layout.B, K, dtype=torch.int64, device=dev)
        self.h_out_scores = torch.zeros(layout.B, K, dtype=torch.float32, pin_memory=pin)
        self.h_out_ids = torch.zeros(layout.B, K, dtype=torch.int64, pin_memory=pin)
        # bf16 on the device: the decode step's own kernels (round 4) -- every product through mmsum_dec_gemm (the reduction split over
        # one-wave workgroups so that every CU pulls weights; the slices meet in the product's last arriver) and the three modalities'
        # cross-attention + entity mean over the cached K / V in ONE launch of one workgroup per (entity, head)
        # (mmsum_decode_cross_attn).  f32 (the parity mode) and shapes outside those kernels keep the generic path below.
        nm = len(layout.mods)
        self.fast = (dev.type == "cuda" and e.dtype in (torch.bfloat16, torch.float32) and nm * R <= 96 and num_beams <= 8 and D % 256 == 0
                     and all(S <= 224 and N <= 32 for (N, S) in layout.mods) and __import__("os").environ.get("MMSUM_DECODE_FAST") != "0")
        if self.fast:
            Fd = cfg.decoder_ffn_dim
            shapes = [(R, 3 * D, D), (R, D, D), (nm * R, D, D), (R, D, 2 * D), (R, Fd, D), (R, D, Fd), (R, cfg.vocab_size, D)]
            nbytes = max(kn.lib.mmsum_dec_gemm_workspace(M_, N_, K_) for (M_, N_, K_) in shapes if K_ % 256 == 0)
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)         # (bf16 only: the f32 mode's products all take mmsum_gemm's f32 weight-streaming kernel)
            n_ent = sum(layout.B * N for (N, S) in layout.mods)
            self.xws = kn.decode_cross_attn_workspace(n_ent, cfg.heads, num_beams, layout.B, nm, dev)
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

    def step(self, tokens, parents, scores, history, t):
        """One decode step at position t.  tokens / parents: numpy [rows] (token at position t of every hypothesis and the row of
        the previous step it continues; parents None at t = 0), scores: numpy [rows] beam scores, history: numpy [rows, >= t+1]
        tokens so far (for the n-gram bans).  -> (scores [B, 2*beams], ids [B, 2*beams]) numpy, best first, id = beam * V + token."""
        R, nb = self.rows, self.nban
        hi = self.h_int_np
        hi[:R] = tokens
        hi[R:2 * R] = np.arange(R, dtype=np.int32) if parents is None else parents
        if nb:
            table = hi[2 * R:2 * R + R * nb].reshape(R, nb)
            table.fill(-1)
            if self.ngram > 0:
                _banned_ngram_table(history, self.ngram, t + 1, table)
            if self.bad_words:
                _bad_word_bans(history, t + 1, self.bad_words, table, 0)
        if self.npen:
            _distinct_tokens(history, t + 1, hi[2 * R + R * nb:].reshape(R, self.npen))
        self.h_sc_np[:] = scores
        self.d_int.copy_(self.h_int, non_blocking=True)
        self.beam_scores.copy_(self.h_sc, non_blocking=True)
        if not self.use_graphs:
            self._step(t)
        else:
            if not self.warm:                                    # one eager pass first: lazy kernel attributes, allocator warm-up
                self._step(t)
                self.warm = True
            g = self.graphs.get(t)
            if g is None:
                import gc
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # no garbage collection while the stream captures: collecting a CUDAGraph of an EARLIER session (a model that went out
                # of scope) destroys it through the HIP API, which is not permitted during capture and aborts the process
                gc_was = gc.isenabled()
                gc.collect()
                gc.disable()
                try:
                    with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                        self._step(t)
                finally:
                    if gc_was:
                        gc.enable()
                if self.pool is None:
                    self.pool = g.pool()
                self.graphs[t] = g
            g.replay()
        self.h_out_scores.copy_(self.out_scores, non_blocking=True)
        self.h_out_ids.copy_(self.out_ids, non_blocking=True)
        if self.out_ids.is_cuda:
            torch.cuda.current_stream().synchronize()             # the step's one host<->device round trip
        return self.h_out_scores.numpy(), self.h_out_ids.numpy()

    def _mm(self, x, w, out, bias=None, epi=kn.EPI_NONE, x2=None):
        """One product of the fast step: mmsum_dec_gemm where its shape rules hold (K, and the split point of a two-tensor x, multiples
        of 256), the general entry point otherwise (tiny test configurations)."""
        K = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
        # Measured per launch inside a graph (tools/decode_kernels_bench.py, 32 rows): the cross-workgroup split wins where the reduction is
        # long -- fc2, K = 4096: 12.0 against 14.5 us -- and loses 1 .. 2 us where K <= 2048 (qkv 9.1 / 6.9, out 7.6 / 6.7, alpha 10.4 / 8.4,
        # fc1 9.0 / 7.2): a product of a few MB sits on a ~7 us floor of launch + one memory round trip + epilogue either way, and the
        # hand-off adds to it.  So: K >= 4096 here, everything else on mmsum_gemm's weight-streaming kernels.
        if x.dtype == torch.bfloat16 and K >= 4096 and K % 256 == 0 and x2 is None and x.shape[0] <= 96 and w.shape[0] <= 8192:
            return kn.dec_gemm(x, w, out, self.ws, bias=bias, epi=epi, x2=x2)
        return kn.gemm(x, w, out, bias=bias, epi=epi, a2=x2)

    def _lm_head(self, x):
        """logits (f32) = x . shared^T + final_logits_bias.  bf16 mode: the un-rounded f32 LayerNorm output against the bf16 weights
        (MMSUM_GEMM_A_F32), which only the weight-streaming kernel serves (at most 64 rows per call, K a multiple of 128): more
        hypothesis rows go through it in row chunks of 64; a width it cannot take falls back to the rounded bf16 rows (f32 output)."""
        e = self.e
        V = e.cfg.vocab_size
        w, bias = e.arena.w(e.bp + "model.shared.weight"), e.buffers[e.bp + "final_logits_bias"].reshape(-1)
        if self.x32 is None:
            return kn.gemm(x, w, self.logits[:, :V], bias=bias)
        for r0 in range(0, self.rows, 64):
            r1 = min(self.rows, r0 + 64)
            kn.gemm(self.x32[r0:r1], w, self.logits[r0:r1, :V], bias=bias)

    def _step_fast(self, t):
        """The decode step with its own kernels (bf16, and f32 -- the parity mode -- on the f32 forms of the same kernels: mmsum_gemm's f32
        weight-streaming kernel, mmsum_decode_cross_attn / mmsum_decode_self_attn in f32): per layer the weight-streaming products (the long-K one with its reduction split
        over workgroups), the cache-walking self-attention, ONE cross-attention launch over the cached K / V of every modality (one
        workgroup per entity and head), the gate and three LayerNorms: 14 launches (16 before)."""
        e, cfg, a = self.e, self.e.cfg, self.e.arena
        D, H, R, Tm = cfg.d_model, cfg.heads, self.rows, self.Tmax
        b = e.bp + "model.decoder."
        scale = 64 ** -0.5
        cur, prev = t & 1, (t - 1) & 1
        self.tokens.copy_(self.d_int[:R].view(R, 1))
        anc = self.anc[cur]
        if t > 0:
            torch.index_select(self.anc[prev], 0, self.d_int[R:2 * R], out=anc)
        anc[:, t] = self.arange
        x = e.empty(R, D)
        kn.embed_ln_fwd(self.tokens, a.w(e.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), self.rd,
                        a.w(b + "rating_embeddings") if self.rd is not None else None, a.f32(b + "layernorm_embedding.weight"),
                        a.f32(b + "layernorm_embedding.bias"), x, self.mean, self.rstd, R, 1, cfg.extra_pos_embeddings + t, 1e-5, 0.0, 0)
        L = self.L
        nm = len(L.mods)
        for i in range(cfg.decoder_layers):
            lb = b + "layers.%d." % i
            q, k, v = e._attn_names(lb, "self_attn")
            qkv = e.empty(R, 3 * D)
            self._mm(x, a.wspan(q + ".weight", v + ".weight", (3 * D, D)), qkv, bias=a.span(a.data, q + ".bias", v + ".bias", (3 * D,)))
            att = e.empty(R, D)
            kn.decode_self_attn(qkv[:, :D], self.kc[i], self.vc[i], anc, att, H, t + 1, Tm, scale, k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
            o = e.empty(R, D)
            self._mm(att, a.w(lb + "self_attn.out_proj.weight"), o, bias=a.f32(lb + "self_attn.out_proj.bias"))
            y = e.empty(R, D)
            kn.add_ln_fwd(o, x, a.f32(lb + "self_attn_layer_norm.weight"), a.f32(lb + "self_attn_layer_norm.bias"), y, self.mean, self.rstd,
                          1e-5, 0.0, 0)
            x = y
            # ---- cross-attention over the cached K / V of every modality + entity mean: one launch (:711-750, :794-869)
            q, _, _ = e._attn_names(lb, "encoder_attn")
            pre = lb + "encoder_attn."
            cq = e.empty(R, D)
            self._mm(x, a.w(q + ".weight"), cq, bias=a.f32(q + ".bias"))
            heads = e.empty(nm * R, D)
            mods = []
            for m, ((N, S), pad) in enumerate(zip(L.mods, self.pads)):
                rows = slice(L.offs[m], L.offs[m] + L.B * N * S)
                mods.append((self.kv[i][rows, :D], self.kv[i][rows, D:], pad, self.nulls[m], N, S))
            kn.decode_cross_attn(cq, mods, heads, self.xws, L.B, self.qpb, H, scale)
            yy = e.empty(nm * R, D)
            self._mm(heads, a.w(pre + "out_proj.weight"), yy, bias=a.f32(pre + "out_proj.bias"))
            if e.multimodal:
                yt, ytab, yimg = yy[:R], yy[R:2 * R], yy[2 * R:]
                pa, pb = e.empty(R, D), e.empty(R, D)
                if R <= 64 and D % 256 == 0 and D <= 4096 and e.dtype == torch.bfloat16:          # alpha and beta: two independent products, one launch
                    kn.gemm_pair([yt, yt], [ytab, yimg], [a.w(pre + "alpha_proj.weight"), a.w(pre + "beta_proj.weight")], [pa, pb],
                                 [a.f32(pre + "alpha_proj.bias"), a.f32(pre + "beta_proj.bias")])
                else:
                    self._mm(yt, a.w(pre + "alpha_proj.weight"), pa, bias=a.f32(pre + "alpha_proj.bias"), x2=ytab)
                    self._mm(yt, a.w(pre + "beta_proj.weight"), pb, bias=a.f32(pre + "beta_proj.bias"), x2=yimg)
                y = e.empty(R, D)                     # gate + residual + LayerNorm in one launch
                kn.gate_add_ln_fwd(pa, pb, yt, ytab, yimg, self.no_table, self.no_img, x, a.f32(lb + "encoder_attn_layer_norm.weight"),
                                   a.f32(lb + "encoder_attn_layer_norm.bias"), y, self.qpb, 1e-5)
            else:
                y = e.empty(R, D)
                kn.add_ln_fwd(yy, x, a.f32(lb + "encoder_attn_layer_norm.weight"), a.f32(lb + "encoder_attn_layer_norm.bias"), y, self.mean,
                              self.rstd, 1e-5, 0.0, 0)
            x = y
            Fd = a.shapes[lb + "fc1.weight"][0]
            h = e.empty(R, Fd)
            self._mm(x, a.w(lb + "fc1.weight"), h, bias=a.f32(lb + "fc1.bias"), epi=kn.EPI_GELU)
            f = e.empty(R, D)
            self._mm(h, a.w(lb + "fc2.weight"), f, bias=a.f32(lb + "fc2.bias"))
            y = e.empty(R, D)
            last = i == cfg.decoder_layers - 1
            kn.add_ln_fwd(f, x, a.f32(lb + "final_layer_norm.weight"), a.f32(lb + "final_layer_norm.bias"), y, self.mean, self.rstd, 1e-5,
                          0.0, 0, y_f32=self.x32 if last else None)
            x = y
        V = cfg.vocab_size
        self._lm_head(x)
        cur_len = t + 1
        eos = cfg.eos_token_id
        force = cfg.bos_token_id if cur_len == 1 else (eos if (cur_len == Tm - 1 and eos is not None) else -1)
        if self.ncand:
            force = -1
        ban = eos if (eos is not None and cur_len < self.min_length) else -1
        banned = self.d_int[2 * R:2 * R + R * self.nban].view(R, self.nban) if self.nban else None
        pen = self.d_int[2 * R + R * self.nban:].view(R, self.npen) if self.npen else None
        kn.beam_topk(self.logits, V, self.beam_scores, banned, force, ban, self.qpb, self.out_scores, self.out_ids, penalized=pen,
                     penalty=self.penalty, penalty_on_logits=self.greedy, ncand=self.ncand)

    def _step(self, t):
        if self.fast:
            return self._step_fast(t)
        e, cfg, a = self.e, self.e.cfg, self.e.arena
        D, H, R, Tm = cfg.d_model, cfg.heads, self.rows, self.Tmax
        b = e.bp + "model.decoder."
        scale = 64 ** -0.5
        cur, prev = t & 1, (t - 1) & 1
        self.tokens.copy_(self.d_int[:R].view(R, 1))
        anc = self.anc[cur]
        if t > 0:                                                # beam reorder (:2996-3003, _reorder_cache) = gather of the ancestor table
            torch.index_select(self.anc[prev], 0, self.d_int[R:2 * R], out=anc)
        anc[:, t] = self.arange
        x = e.empty(R, D)
        kn.embed_ln_fwd(self.tokens, a.w(e.bp + "model.shared.weight"), a.w(b + "embed_positions.weight"), self.rd,
                        a.w(b + "rating_embeddings") if self.rd is not None else None, a.f32(b + "layernorm_embedding.weight"),
                        a.f32(b + "layernorm_embedding.bias"), x, self.mean, self.rstd, R, 1, cfg.extra_pos_embeddings + t, 1e-5, 0.0, 0)
        nm = len(self.L.mods)
        for i in range(cfg.decoder_layers):
            lb = b + "layers.%d." % i
            # ---- self-attention over the cache (:442-461 with layer_state, :776-815)
            q, k, v = e._attn_names(lb, "self_attn")
            qkv = e.empty(R, 3 * D)
            kn.gemm(x, a.wspan(q + ".weight", v + ".weight", (3 * D, D)), qkv, bias=a.span(a.data, q + ".bias", v + ".bias", (3 * D,)))
            att = e.empty(R, D)           # the kernel appends this step's K / V to the caches itself (position t of every row)
            kn.decode_self_attn(qkv[:, :D], self.kc[i], self.vc[i], anc, att, H, t + 1, Tm, scale, k_new=qkv[:, D:2 * D], v_new=qkv[:, 2 * D:])
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
            last = i == cfg.decoder_layers - 1
            kn.add_ln_fwd(f, x, a.f32(lb + "final_layer_norm.weight"), a.f32(lb + "final_layer_norm.bias"), y, self.mean, self.rstd, 1e-5,
                          0.0, 0, y_f32=self.x32 if last else None)
            x = y
        V = cfg.vocab_size
        self._lm_head(x)                                                                                     # :2281
        # ---- tail: forced BOS / EOS, log-softmax, min-length and n-gram bans, + beam scores, top 2*beams per business
        cur_len = t + 1
        eos = cfg.eos_token_id
        force = cfg.bos_token_id if cur_len == 1 else (eos if (cur_len == Tm - 1 and eos is not None) else -1)      # :3084-3089
        if self.ncand:
            force = -1
        ban = eos if (eos is not None and cur_len < self.min_length) else -1
        banned = self.d_int[2 * R:2 * R + R * self.nban].view(R, self.nban) if self.nban else None
        pen = self.d_int[2 * R + R * self.nban:].view(R, self.npen) if self.npen else None
        kn.beam_topk(self.logits, V, self.beam_scores, banned, force, ban, self.qpb, self.out_scores, self.out_ids, penalized=pen,
                     penalty=self.penalty, penalty_on_logits=self.greedy, ncand=self.ncand)


def _session(engine, layout, num_beams, max_length, has_rating, min_length, ngram, bad_words=None, penalty=1.0, greedy=False, ncand=0):
    bw = tuple(tuple(int(t) for t in w) for w in bad_words) if bad_words else ()
    key = (tuple(layout.mods), layout.B, num_beams, max_length, has_rating, min_length, ngram, bw, float(penalty), bool(greedy), int(ncand))
    cache = engine.__dict__.setdefault("_decode_sessions", {})
    if key not in cache:
        if len(cache) >= 4:                          # static buffers + graphs per shape: keep only a few
            cache.pop(next(iter(cache)))
        cache[key] = DecodeSession(engine, layout, num_beams, max_length, has_rating, min_length, ngram, bw, penalty, greedy, ncand)
    return cache[key]


@torch.no_grad()
def greedy_search(engine, hiddens, layout, pads, rating_diff, max_length, min_length, no_repeat_ngram_size, decoder_start_token_id,
                  bad_words_ids=None, repetition_penalty=1.0):
    """Greedy decoding (_generate_no_beam_search with do_sample = False, modeling_multimodalsum.py:2767-2868 / :1767-1868) on the decode
    session with ONE hypothesis per business: the step's tail (forced BOS / EOS, repetition penalty and bans on the LOGITS -- the
    reference post-processes the tensor its argmax reads -- then the best candidate) is mmsum_beam_topk with num_beams = 1; a finished
    row keeps being fed and is padded with pad_token_id; the loop ends when every row has produced EOS.  Returns LongTensor [B, L]."""
    cfg = engine.cfg
    pad, eos, V = cfg.pad_token_id, cfg.eos_token_id, cfg.vocab_size
    B = layout.B
    sess = _session(engine, layout, 1, max_length, rating_diff is not None, min_length if eos is not None else 0, no_repeat_ngram_size,
                    bad_words_ids, repetition_penalty, greedy=True)
    sess.begin(hiddens, pads, rating_diff)
    hist = np.full((B, max_length), pad, dtype=np.int64)
    hist[:, 0] = decoder_start_token_id
    last = hist[:, 0].astype(np.int32)
    unfinished = np.ones(B, dtype=bool)
    zeros = np.zeros(B, dtype=np.float32)
    cur_len = 1
    while cur_len < max_length:
        _, top_i = sess.step(last, None, zeros, hist, cur_len - 1)
        tok = (top_i[:, 0] % V).astype(np.int64)
        add = np.where(unfinished, tok, pad) if eos is not None else tok
        hist[:, cur_len] = add
        last = add.astype(np.int32)
        cur_len += 1
        if eos is not None:
            unfinished &= add != eos
        if not unfinished.any():
            break
    return torch.from_numpy(hist[:, :cur_len].copy()).to(engine.device)


def sample_from_candidates(scores, tokens, u, temperature=1.0, top_k=50, top_p=1.0):
    """One draw per row from the step's candidate lists (host; a few dozen numbers per row).  scores [B, K] = the K best post-processed
    log-probabilities of a row, best first, tokens [B, K] their ids; K > top_k: the extra candidates tell whether the top_k-th value
    is tied (the reference keeps every token whose logit is not BELOW the top_k-th largest, generation_utils.py:923-927); a tie that
    runs to the end of the list cannot be resolved and raises.
    / temperature, the top-p cut on the sorted probabilities (:929-944: everything up to and including the first token whose
    cumulative probability exceeds top_p), softmax over what is left, then the pinned sampling rule of the oracle and the fixtures
    (oracle/generate_oracle.inverse_cdf_draw: the first token, in vocabulary order, whose cumulative probability exceeds u * total)."""
    B, K = scores.shape
    out = np.zeros(B, dtype=np.int64)
    for b in range(B):
        s = scores[b].astype(np.float64)
        keep = np.isfinite(s)
        keep[top_k:] &= s[top_k:] == s[top_k - 1]                # beyond the top_k only exact ties with the top_k-th value stay
        if keep[top_k:].all() and K > top_k and np.isfinite(s[K - 1]) and s[K - 1] == s[top_k - 1]:
            raise RuntimeError("sampling: more ties at the top_k-th logit than the candidate list holds")
        s, t = s[keep] / temperature, tokens[b][keep]
        if top_p < 1.0:
            p = np.exp(s - s.max())
            p /= p.sum()
            remove = np.cumsum(p) > top_p
            remove[1:] = remove[:-1].copy()
            remove[0] = False
            s, t = s[~remove], t[~remove]
        p = np.exp(s - s.max())
        order = np.argsort(t, kind="stable")                      # vocabulary order
        cdf = np.cumsum(p[order])
        out[b] = t[order][int(np.argmax(cdf > float(u[b]) * cdf[-1]))]
    return out


@torch.no_grad()
def sample_search(engine, hiddens, layout, pads, rating_diff, max_length, min_length, no_repeat_ngram_size, decoder_start_token_id,
                  bad_words_ids=None, repetition_penalty=1.0, temperature=1.0, top_k=50, top_p=1.0, draws=None):
    """Sampling (_generate_no_beam_search with do_sample = True, modeling_multimodalsum.py:1767-1868): the greedy session with
    candidate lists -- no forced BOS / EOS (the reference skips them when it samples), repetition penalty and bans on the logits, the
    top_k + 4 (at most 64) best of every row from mmsum_beam_topk -- and one draw per row on the host (sample_from_candidates).
    draws: callable(step, B) -> B uniforms in [0, 1) (tests and fixtures pass recorded ones); default torch.rand on the host, i.e.
    torch.manual_seed reproduces a run of THIS implementation; torch.multinomial's stream of the reference is not reproducible across
    implementations.  top_k must be 1 .. 63 (the reference's default is 50).  Returns LongTensor [B, L]."""
    cfg = engine.cfg
    pad, eos, V = cfg.pad_token_id, cfg.eos_token_id, cfg.vocab_size
    if not (1 <= int(top_k) <= 63):
        raise NotImplementedError("sampling is built for 1 <= top_k <= 63 (the candidate kernel returns at most 64 of a row); top_k = 0 "
                                  "(the whole vocabulary) is not")
    if not (temperature > 0.0):
        raise ValueError("temperature must be positive")
    top_k = min(int(top_k), V - 1)
    B = layout.B
    sess = _session(engine, layout, 1, max_length, rating_diff is not None, min_length if eos is not None else 0, no_repeat_ngram_size,
                    bad_words_ids, repetition_penalty, greedy=True, ncand=min(64, top_k + 4))      # up to three ties with the top_k-th value are followed
    sess.begin(hiddens, pads, rating_diff)
    hist = np.full((B, max_length), pad, dtype=np.int64)
    hist[:, 0] = decoder_start_token_id
    last = hist[:, 0].astype(np.int32)
    unfinished = np.ones(B, dtype=bool)
    zeros = np.zeros(B, dtype=np.float32)
    cur_len, step = 1, 0
    while cur_len < max_length:
        top_s, top_i = sess.step(last, None, zeros, hist, cur_len - 1)
        u = draws(step, B) if draws is not None else torch.rand(B, dtype=torch.float64).numpy()
        tok = sample_from_candidates(top_s, top_i % V, np.asarray(u, dtype=np.float64), temperature, top_k, top_p)
        step += 1
        add = np.where(unfinished, tok, pad) if eos is not None else tok
        hist[:, cur_len] = add
        last = add.astype(np.int32)
        cur_len += 1
        if eos is not None:
            unfinished &= add != eos
        if not unfinished.any():
            break
    return torch.from_numpy(hist[:, :cur_len].copy()).to(engine.device)


@torch.no_grad()
def beam_search(engine, hiddens, layout, pads, rating_diff, num_beams, max_length, min_length, no_repeat_ngram_size, early_stopping,
                length_penalty, decoder_start_token_id, trace=None, bad_words_ids=None, repetition_penalty=1.0):
    """Greedy beam search (_generate_beam_search :2803-3067).  Returns LongTensor [B, L] on the engine's device.
    trace (a list, tests only): receives one dict per decode step -- the hypotheses the step scored, their beam scores, which
    businesses were still open and the 2 * num_beams candidates the device returned -- so that a CPU oracle can re-score the
    very same hypotheses (tests/test_timed_path_gpu.py holds the bf16 decode path to the fp32 oracle step by step)."""
    cfg = engine.cfg
    pad, bos, eos, V = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id, cfg.vocab_size
    dev = engine.device
    B = layout.B
    R = B * num_beams
    sess = _session(engine, layout, num_beams, max_length, rating_diff is not None, min_length if eos is not None else 0, no_repeat_ngram_size,
                    bad_words_ids, repetition_penalty)
    sess.begin(hiddens, pads, rating_diff)
    hist = np.full((R, max_length), pad, dtype=np.int64)                   # host copy of input_ids
    hist[:, 0] = decoder_start_token_id
    last = hist[:, 0].astype(np.int32)
    parents = None
    hyps = [_Hypotheses(num_beams, max_length, length_penalty, early_stopping) for _ in range(B)]
    host_scores = np.where(np.arange(R) % num_beams == 0, 0.0, -1e9).astype(np.float32)    # :2836-2840: only the first beam of a business is live
    done = [False] * B
    cur_len = 1
    while cur_len < max_length:
        # one decode step + its tail on the device: forced BOS / EOS (:3084-3089), log_softmax (:2874), min-length and n-gram
        # bans, + beam scores, top 2*num_beams of every business (:2925)
        top_s, top_i = sess.step(last, parents, host_scores, hist, cur_len - 1)
        if trace is not None:
            trace.append({"cur_len": cur_len, "prefixes": hist[:, :cur_len].copy(), "beam_scores": host_scores.copy(), "open": [not d for d in done],
                          "top_scores": top_s.copy(), "top_ids": top_i.copy()})
        top_beam, top_tok = top_i // V, top_i % V
        nxt_s = np.zeros(R, dtype=np.float32)
        nxt_t = np.full(R, pad, dtype=np.int32)
        nxt_p = np.zeros(R, dtype=np.int32)
        # the common case of a business -- no EOS among its first candidates -- is the first num_beams candidates as they come
        plain = np.ones(B, dtype=bool) if eos is None else ~(top_tok[:, :num_beams + 1] == eos).any(axis=1)
        # (checked one past num_beams: if none of the first num_beams is EOS the loop below stops there and never looks further)
        if plain.any():
            rows_ = (np.arange(B)[:, None] * num_beams + np.arange(num_beams)[None, :])
            sel = plain[:, None] & ~np.asarray(done, dtype=bool)[:, None]
            nxt_s.reshape(B, num_beams)[:] = np.where(sel, top_s[:, :num_beams], 0.0)
            nxt_t.reshape(B, num_beams)[:] = np.where(sel, top_tok[:, :num_beams], pad)
            nxt_p.reshape(B, num_beams)[:] = np.where(sel, np.arange(B)[:, None] * num_beams + top_beam[:, :num_beams], 0)
        for b in range(B):
            if done[b]:
                continue                                                   # padded out: score 0, pad token, parent row 0 (:2938-2941)
            if plain[b]:
                done[b] = hyps[b].is_done(float(top_s[b].max()), cur_len)
                continue
            n_sent = 0
            for rank in range(2 * num_beams):
                tok, sc = int(top_tok[b, rank]), float(top_s[b, rank])
                row = b * num_beams + int(top_beam[b, rank])
                if eos is not None and tok == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(hist[row, :cur_len].tolist(), sc)
                else:
                    j = b * num_beams + n_sent
                    nxt_s[j], nxt_t[j], nxt_p[j] = sc, tok, row
                    n_sent += 1
                if n_sent == num_beams:
                    break
            done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
            assert n_sent == num_beams, "Beam should always be full"
        if all(done):
            break
        host_scores, parents, last = nxt_s, nxt_p, nxt_t
        hist = hist[parents]
        hist[:, cur_len] = last
        cur_len += 1
    for b in range(B):
        if done[b]:
            continue
        for beam in range(num_beams):
            row = b * num_beams + beam
            hyps[b].add(hist[row, :cur_len].tolist(), float(host_scores[row]))
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
