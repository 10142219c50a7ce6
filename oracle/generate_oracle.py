"""CPU restatement of the reference's beam-search generation (TEST INFRASTRUCTURE ONLY).

Follows BartForMultiEncConditionalGeneration.generate / _generate_beam_search
(/root/reference/src/transformer/modeling_multimodalsum.py:2295-2693, 2803-3067), the score post-processing
and n-gram ban of /root/reference/src/transformer/generation_utils.py:57-98, 848-868, BeamHypotheses
(:948-993) and the logits adjustment (:3084-3102), for the greedy (do_sample=False) beam-search branch that
`src/test.py:156-158` uses.  The reference decodes one token at a time with cached keys/values; the cached step is
algebraically the last row of a causal decoder pass over the whole prefix, which is what this restatement runs
(bart_oracle.bart_decoder), so no cache logic is needed here.

Pinned by tests/golden/g1_beam.npz (token ids produced by the reference itself, oracle/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import torch
import torch.nn.functional as F

from . import bart_oracle as bo


class Hypotheses:
    """n-best list of finished hypotheses of one batch entry (generation_utils.py:948-993)."""

    def __init__(self, num_beams, max_length, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.max_length = max_length - 1
        self.items = []                      # (score, tokens)
        self.worst = 1e9

    def __len__(self):
        return len(self.items)

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


def banned_ngram_tokens(rows, n, cur_len):
    """Tokens that would complete an n-gram already present in the hypothesis (generation_utils.py:848-868)."""
    if cur_len + 1 < n:
        return [[] for _ in rows]
    out = []
    for toks in rows:
        seen = {}
        for i in range(len(toks) - n + 1):
            seen.setdefault(tuple(toks[i:i + n - 1]), []).append(toks[i + n - 1])
        out.append(seen.get(tuple(toks[cur_len + 1 - n:cur_len]), []))
    return out


def next_token_logits(sd, cfg, input_ids, hiddens, masks, rating_diff, multimodal, prefix):
    """Logits of the next token for every hypothesis row: last row of a causal pass over the prefix."""
    T = input_ids.shape[1]
    causal = torch.triu(torch.full((T, T), float("-inf")), 1)
    h = bo.bart_decoder(sd, cfg, input_ids, hiddens, masks, None, causal, rating_diff, multimodal, False, prefix)
    return F.linear(h[:, -1, :], sd[prefix + "model.shared.weight"])


def banned_bad_words(rows, bad_words_ids):
    """calc_banned_bad_words_ids (generation_utils.py:871-904), literally -- including its quirk: a bad word longer than the NUMBER OF
    ROWS (`len(prev_input_ids)`, the batch dimension, not the sequence length) never matches."""
    out = []
    for toks in rows:
        banned = []
        for seq in bad_words_ids:
            assert len(seq) > 0
            head = list(seq[:-1])
            if len(head) == 0:
                ok = True
            elif len(head) > len(rows):
                ok = False
            else:
                ok = toks[-len(head):] == head
            if ok:
                banned.append(seq[-1])
        out.append(banned)
    return out


def repetition_penalty_(scores, rows, penalty):
    """enforce_repetition_penalty_ (generation_utils.py:47-55): every token already in the row: score < 0 -> * penalty, else / penalty."""
    for i, toks in enumerate(rows):
        for t in set(toks):
            if scores[i, t] < 0:
                scores[i, t] *= penalty
            else:
                scores[i, t] /= penalty


def postprocess_(scores, input_ids, cfg, cur_len, min_length, no_repeat_ngram_size, bad_words_ids, repetition_penalty):
    """postprocess_next_token_scores (generation_utils.py:57-98), in its order: repetition penalty, min-length EOS ban, n-gram bans, bad words."""
    rows = input_ids.tolist()
    eos = cfg.eos_token_id
    if repetition_penalty != 1.0:
        repetition_penalty_(scores, rows, repetition_penalty)
    if eos is not None and cur_len < min_length:
        scores[:, eos] = float("-inf")
    if no_repeat_ngram_size > 0:
        for i, banned in enumerate(banned_ngram_tokens(rows, no_repeat_ngram_size, cur_len)):
            scores[i, banned] = float("-inf")
    if bad_words_ids is not None:
        for i, banned in enumerate(banned_bad_words(rows, bad_words_ids)):
            scores[i, banned] = float("-inf")
    return scores


def _forced(logits, cfg, cur_len, max_length):
    """adjust_logits_during_generation (:3084-3102): BOS forced at length 1, EOS at max_length - 1."""
    bos, eos = cfg.bos_token_id, cfg.eos_token_id
    if cur_len == 1:
        keep = logits[:, bos].clone()
        logits.fill_(float("-inf"))
        logits[:, bos] = keep
    if cur_len == max_length - 1 and eos is not None:
        keep = logits[:, eos].clone()
        logits.fill_(float("-inf"))
        logits[:, eos] = keep
    return logits


def greedy_search(sd, cfg, hiddens, masks, rating_diff, multimodal, max_length, min_length=0, no_repeat_ngram_size=0, bad_words_ids=None,
                  repetition_penalty=1.0, decoder_start_token_id=None, prefix=""):
    """_generate_no_beam_search with do_sample = False (:2767-2868 / :1767-1868): argmax of the post-processed LOGITS (the post-processing
    works in place on the tensor the argmax reads: penalty and bans apply to raw logits, there is no log_softmax), finished rows are
    fed and padded with pad_token_id, the loop ends when every row has produced EOS.  Returns LongTensor [B, L]."""
    pad, bos, eos = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id
    start = bos if decoder_start_token_id is None else decoder_start_token_id
    first = hiddens[0] if multimodal else hiddens
    B = first.shape[0]
    input_ids = torch.full((B, 1), start, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    cur_len = 1
    memo_store = {}
    while cur_len < max_length:
        prev, bo.KV_MEMO = bo.KV_MEMO, memo_store
        try:
            logits = next_token_logits(sd, cfg, input_ids, hiddens, masks, rating_diff, multimodal, prefix)
        finally:
            bo.KV_MEMO = prev
        logits = _forced(logits, cfg, cur_len, max_length)
        postprocess_(logits, input_ids, cfg, cur_len, min_length, no_repeat_ngram_size, bad_words_ids, repetition_penalty)
        nxt = torch.argmax(logits, dim=-1)
        add = nxt * unfinished + pad * (1 - unfinished) if eos is not None else nxt
        input_ids = torch.cat([input_ids, add[:, None]], dim=1)
        cur_len += 1
        if eos is not None:
            unfinished = unfinished * (add != eos).long()
        if unfinished.max() == 0:
            break
    return input_ids


def top_k_top_p_filtering_(logits, top_k=0, top_p=1.0, min_tokens_to_keep=1):
    """generation_utils.py:907-945, in place on logits [rows, V]: keep the tokens whose logit is not below the top_k-th largest (ties
    with it stay), then, of the sorted probabilities, everything up to and including the first token whose cumulative probability
    exceeds top_p."""
    if top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        logits[logits < torch.topk(logits, top_k)[0][..., -1, None]] = float("-inf")
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        remove = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1) > top_p
        if min_tokens_to_keep > 1:
            remove[..., :min_tokens_to_keep] = False
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        logits[remove.scatter(1, sorted_indices, remove)] = float("-inf")
    return logits


def inverse_cdf_draw(probs, u):
    """The sampling rule the fixtures and tests pin torch.multinomial to (its own random stream differs between devices and versions):
    token = the first index, in VOCABULARY order, whose cumulative probability (float64) exceeds u * total.  probs [rows, V], u [rows]."""
    cdf = torch.cumsum(probs.double(), dim=-1)
    return (cdf > (u.double() * cdf[:, -1])[:, None]).float().argmax(-1)


def sample_search(sd, cfg, hiddens, masks, rating_diff, multimodal, max_length, draws, min_length=0, no_repeat_ngram_size=0, bad_words_ids=None,
                  repetition_penalty=1.0, temperature=1.0, top_k=50, top_p=1.0, decoder_start_token_id=None, prefix=""):
    """_generate_no_beam_search with do_sample = True (:1767-1868): NO forced BOS / EOS (adjust_logits_during_generation is skipped when
    sampling, :1811), post-processing on the logits, / temperature, top_k_top_p_filtering, softmax, one draw per row -- draws [steps, B]
    uniforms through inverse_cdf_draw in the place of torch.multinomial.  Returns LongTensor [B, L]."""
    pad, bos, eos = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id
    start = bos if decoder_start_token_id is None else decoder_start_token_id
    first = hiddens[0] if multimodal else hiddens
    B = first.shape[0]
    input_ids = torch.full((B, 1), start, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    cur_len, step = 1, 0
    memo_store = {}
    while cur_len < max_length:
        prev, bo.KV_MEMO = bo.KV_MEMO, memo_store
        try:
            logits = next_token_logits(sd, cfg, input_ids, hiddens, masks, rating_diff, multimodal, prefix)
        finally:
            bo.KV_MEMO = prev
        postprocess_(logits, input_ids, cfg, cur_len, min_length, no_repeat_ngram_size, bad_words_ids, repetition_penalty)
        if temperature != 1.0:
            logits = logits / temperature
        top_k_top_p_filtering_(logits, top_k=top_k, top_p=top_p)
        nxt = inverse_cdf_draw(F.softmax(logits, dim=-1), draws[step])
        step += 1
        add = nxt * unfinished + pad * (1 - unfinished) if eos is not None else nxt
        input_ids = torch.cat([input_ids, add[:, None]], dim=1)
        cur_len += 1
        if eos is not None:
            unfinished = unfinished * (add != eos).long()
        if unfinished.max() == 0:
            break
    return input_ids


def step_scores(sd, cfg, input_ids, hiddens, masks, rating_diff, multimodal, max_length, min_length=0, no_repeat_ngram_size=0, prefix="",
                bad_words_ids=None, repetition_penalty=1.0):
    """Log-probabilities of the next token for every hypothesis row `input_ids` [rows, cur_len], with the reference's adjustments:
    BOS forced at length 1 and EOS at max_length - 1 on the logits (:3084-3102), log_softmax (:2874), EOS banned below min_length
    (:2877-2879) and the n-gram ban (:2890-2899) on the scores.  hiddens / masks / rating_diff: one entry per row."""
    bos, eos = cfg.bos_token_id, cfg.eos_token_id
    cur_len = input_ids.shape[1]
    logits = next_token_logits(sd, cfg, input_ids, hiddens, masks, rating_diff, multimodal, prefix)
    if cur_len == 1:
        keep = logits[:, bos].clone()
        logits.fill_(float("-inf"))
        logits[:, bos] = keep
    if cur_len == max_length - 1 and eos is not None:
        keep = logits[:, eos].clone()
        logits.fill_(float("-inf"))
        logits[:, eos] = keep
    scores = F.log_softmax(logits, dim=-1)
    return postprocess_(scores, input_ids, cfg, cur_len, min_length, no_repeat_ngram_size, bad_words_ids, repetition_penalty)


def beam_search(sd, cfg, hiddens, masks, rating_diff, multimodal, num_beams, max_length, min_length=0,
                no_repeat_ngram_size=0, early_stopping=False, length_penalty=1.0, decoder_start_token_id=None, prefix="",
                return_scores=False, margins=None, guide=None, return_all=False, bad_words_ids=None, repetition_penalty=1.0):
    """hiddens/masks: list of [B,N,S,D] / [B,N,S] (multimodal) or single tensors.  Returns LongTensor [B, L] (return_scores: and the
    best hypothesis' score per business, which the reference does not return; tests check sequence_score against it).
    margins (a list, tests only): receives per decode step the smallest gap between two consecutive candidates among the best
    2 * num_beams + 1 of any open business -- how far the step's ranking is from a tie (an id-exact comparison with another
    implementation is only meaningful where this is well above that implementation's rounding).
    guide (tests only): guide(step, input_ids, beam_scores, cand [B, num_beams * V]) -> (scores [B, 2 * num_beams], ids [B, 2 * num_beams]) or
    None; when it returns a pair, that pair takes the place of torch.topk's (:2925) for the step.  torch.topk leaves the order of
    candidates with (nearly) equal scores unspecified; a test that holds another implementation to this search passes a guide that
    checks the other implementation's choice against `cand` and, where it is a valid top 2 * num_beams up to a stated tolerance,
    adopts its ORDER with this search's own scores -- every later step then sees the same hypotheses and the final ids must agree.
    return_all: also return, per business, the finished hypotheses [(score, tokens)] the final choice was made from."""
    pad, bos, eos, V = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id, cfg.vocab_size
    start = bos if decoder_start_token_id is None else decoder_start_token_id
    first = hiddens[0] if multimodal else hiddens
    B = first.shape[0]
    rep = lambda t: None if t is None else t.repeat_interleave(num_beams, dim=0)      # noqa: E731  (:2599-2627)
    if multimodal:
        hid, msk = [rep(h) for h in hiddens], [rep(m) for m in masks]
    else:
        hid, msk = rep(hiddens), rep(masks)
    rd = rep(rating_diff)
    input_ids = torch.full((B * num_beams, 1), start, dtype=torch.long)
    hyps = [Hypotheses(num_beams, max_length, length_penalty, early_stopping) for _ in range(B)]
    beam_scores = torch.zeros(B, num_beams)
    beam_scores[:, 1:] = -1e9                                                          # (:2848-2850)
    beam_scores = beam_scores.view(-1)
    done = [False] * B
    cur_len = 1
    next_scores = next_tokens = None

    def _scores(ids):
        # cross-attention K / V of the (fixed) memory: projected once per generate() call like the reference's cache (:804-815)
        prev, bo.KV_MEMO = bo.KV_MEMO, memo_store
        try:
            return step_scores(sd, cfg, ids, hid, msk, rd, multimodal, max_length, min_length, no_repeat_ngram_size, prefix,
                               bad_words_ids, repetition_penalty)
        finally:
            bo.KV_MEMO = prev
    memo_store = {}
    while cur_len < max_length:
        scores = _scores(input_ids)
        cand = (scores + beam_scores[:, None]).view(B, num_beams * V)
        next_scores, next_tokens = torch.topk(cand, 2 * num_beams, dim=1, largest=True, sorted=True)
        if guide is not None:
            g = guide(cur_len - 1, input_ids, beam_scores, cand)
            if g is not None:
                next_scores, next_tokens = g
        if margins is not None:
            top = torch.topk(cand, 2 * num_beams + 1, dim=1, largest=True, sorted=True)[0]
            gaps = (top[:, :-1] - top[:, 1:])
            gaps = torch.where(torch.isfinite(gaps) & (top[:, 1:] > -1e8), gaps, torch.full_like(gaps, float("inf")))   # (dead beams carry -1e9)
            open_rows = [b for b in range(B) if not done[b]]
            margins.append(float(gaps[open_rows].min()) if open_rows else float("inf"))
        nxt = []
        for b in range(B):
            if done[b]:
                nxt.extend([(0.0, pad, 0)] * num_beams)
                continue
            sent = []
            for rank, (tok_id, sc) in enumerate(zip(next_tokens[b].tolist(), next_scores[b].tolist())):
                beam, tok = tok_id // V, tok_id % V
                row = b * num_beams + beam
                if eos is not None and tok == eos:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(input_ids[row].clone(), sc)
                else:
                    sent.append((sc, tok, row))
                if len(sent) == num_beams:
                    break
            done[b] = done[b] or hyps[b].is_done(next_scores[b].max().item(), cur_len)
            assert len(sent) == num_beams
            nxt.extend(sent)
        if all(done):
            break
        beam_scores = torch.tensor([x[0] for x in nxt], dtype=torch.float32)
        beam_tokens = torch.tensor([x[1] for x in nxt], dtype=torch.long)
        beam_idx = torch.tensor([x[2] for x in nxt], dtype=torch.long)
        input_ids = torch.cat([input_ids[beam_idx], beam_tokens[:, None]], dim=1)
        cur_len += 1
        # every hypothesis row of a batch entry shares its encoder tensors: reordering them is a no-op
    for b in range(B):
        if done[b]:
            continue
        for beam in range(num_beams):
            row = b * num_beams + beam
            hyps[b].add(input_ids[row], beam_scores[row].item())
    best = [sorted(h.items, key=lambda x: x[0])[-1][1] for h in hyps]
    best_scores = [sorted(h.items, key=lambda x: x[0])[-1][0] for h in hyps]
    lens = [len(t) for t in best]
    if min(lens) != max(lens):
        L = min(max(lens) + 1, max_length)
        out = torch.full((B, L), pad, dtype=torch.long)
        for i, t in enumerate(best):
            out[i, :lens[i]] = t
            if lens[i] < max_length:
                out[i, lens[i]] = eos
    else:
        out = torch.stack(best).long()
    if return_all:
        return out, [[(float(sc), [int(t) for t in toks]) for sc, toks in h.items] for h in hyps]
    return (out, best_scores) if return_scores else out


def sequence_score(sd, cfg, seq, hiddens, masks, rating_diff, multimodal, max_length, min_length=0, length_penalty=1.0, prefix=""):
    """The score _generate_beam_search assigns to the finished hypothesis that `seq` (one returned row: start token, tokens,
    optionally the appended EOS and pads, modeling_multimodalsum.py:3040-3067) stands for: the sum of the log-probabilities of
    its tokens and of the closing EOS under the same logits adjustment (:3084-3102: BOS forced at length 1, EOS forced at
    max_length - 1, EOS banned below min_length), divided by len(tokens) ** length_penalty (BeamHypotheses.add, :959-960).
    Teacher-forced: one causal decoder pass over the hypothesis.  hiddens / masks hold ONE business."""
    pad, bos, eos = cfg.pad_token_id, cfg.bos_token_id, cfg.eos_token_id
    toks = [int(t) for t in seq.tolist()]
    while len(toks) > 1 and toks[-1] == pad:
        toks.pop()
    if len(toks) > 1 and toks[-1] == eos:
        toks.pop()
    ids = torch.tensor([toks], dtype=torch.long)
    T = ids.shape[1]
    causal = torch.triu(torch.full((T, T), float("-inf")), 1)
    h = bo.bart_decoder(sd, cfg, ids, hiddens, masks, None, causal, rating_diff, multimodal, False, prefix)
    logits = F.linear(h[0], sd[prefix + "model.shared.weight"])               # row t: the distribution of token t + 1
    total = 0.0
    for t in range(T):
        cur_len = t + 1
        row = logits[t].clone()
        nxt = toks[t + 1] if t + 1 < T else eos
        if cur_len == 1:
            keep = row[bos].clone()
            row.fill_(float("-inf"))
            row[bos] = keep
        if cur_len == max_length - 1:
            keep = row[eos].clone()
            row.fill_(float("-inf"))
            row[eos] = keep
        lp = F.log_softmax(row, dim=-1)
        if cur_len < min_length:
            lp[eos] = float("-inf")
        if t + 1 == T and cur_len >= max_length:
            break                                                            # an unfinished hypothesis of full length: no closing EOS
        total += float(lp[nxt])
    return total / (len(toks) ** length_penalty)
