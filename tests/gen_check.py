"""Holding a beam search to the CPU oracle's over many steps (shared by the GPU generation tests and the CPU host-logic test).

A beam-search step ranks num_beams * V candidates by f32 scores whose magnitude grows with the length, and models with random weights
rank many candidates within 1e-4 of each other (tools/gen_margin_probe.py: the oracle's OWN 127-step run at BART-large width meets gaps of
0 .. 6e-5 between consecutive candidates of its best 2 * num_beams + 1; torch.topk, :2925, breaks such ties in an unspecified order).  Two
correct implementations that sum in different orders differ by ~1e-5 on a logit, so over 100+ steps "the same ids as an independent run"
is not a property either of them has.  The property that IS checkable, and that a wrong ancestor table, a missed n-gram ban, a stale
cache row or a mis-scored beam breaks by many nats:

  the oracle's search, run with a guide that at every step (a) requires its hypotheses to equal the hypotheses the other search holds,
  (b) re-scores the step ON THE OTHER SEARCH'S running beam scores (oracle log-probabilities of the step + the beam scores the other
  search carried into it: the step is judged, not the rounding it has accumulated so far) and requires every (score, beam, token)
  candidate the other search returned to carry that score within `tie`, -inf candidates (forced tokens, bans) to agree on being
  -inf, and no candidate the other search passed over to beat its worst pick by more than 2 * tie, and then (c) adopts the other
  search's ORDER with the oracle's own running scores, ends on the same hypotheses; and the ids the other search returned are, per
  business, the oracle's best finished hypothesis or one whose score is within `tie` per token of it.
"""
import torch

from oracle import generate_oracle as go


def guided_check(out_ids, trace, sd, ocfg, hiddens, masks, rd, multimodal, kw, tie, start_token):
    """out_ids [B, L] (cpu) and trace: what generation.beam_search returned / recorded.  hiddens / masks / rd: the oracle's CPU
    inputs.  -> dict(worst_score, worst_rank, steps)."""
    beams, max_length = kw["num_beams"], kw["max_length"]
    V = ocfg.vocab_size
    B = out_ids.shape[0]
    stat = {"worst_score": 0.0, "worst_rank": 0.0, "steps": 0}

    def guide(i, input_ids, beam_scores, cand):
        assert i < len(trace), "the oracle's search runs longer than the search under test (step %d)" % i
        st = trace[i]
        stat["steps"] += 1
        assert st["cur_len"] == input_ids.shape[1]
        pre = torch.from_numpy(st["prefixes"]).long()
        top_s, top_i = torch.topk(cand, 2 * beams, dim=1, largest=True, sorted=True)
        # the step's log-probabilities on the other search's running scores
        step_lp = cand.view(B * beams, V) - beam_scores[:, None]
        cand_h = (step_lp + torch.from_numpy(st["beam_scores"]).float()[:, None]).view(B, beams * V)
        for b in range(B):
            if not st["open"][b]:
                continue                                  # finished business: padded rows on both sides, the step's pick is never used
            rows = slice(b * beams, (b + 1) * beams)
            assert torch.equal(pre[rows], input_ids[rows]), ("hypotheses differ at step %d, business %d" % (i, b), pre[rows], input_ids[rows])
            got_ids = torch.from_numpy(st["top_ids"][b]).long()
            got_sc = torch.from_numpy(st["top_scores"][b]).float()
            want = cand_h[b, got_ids]
            assert len(set(got_ids.tolist())) == got_ids.numel() or not bool(torch.isfinite(want).all()), ("duplicate candidates", i, b, got_ids)
            fin_w, fin_g = want > -1e8, got_sc > -1e8
            assert bool((fin_w == fin_g).all()), ("forced / banned / dead candidates disagree at step %d, business %d" % (i, b), got_sc, want)
            if fin_w.any():
                dev = float((got_sc[fin_w] - want[fin_w]).abs().max())
                assert dev <= tie, ("a candidate's score is not the oracle's: step %d business %d, off by %.3e" % (i, b, dev), got_sc, want)
                rest = cand_h[b].clone()
                rest[got_ids] = float("-inf")
                over = float(rest.max() - want[fin_w].min())
                assert over <= 2 * tie, ("a better candidate was passed over: step %d business %d, by %.3e" % (i, b, over))
                stat["worst_score"], stat["worst_rank"] = max(stat["worst_score"], dev), max(stat["worst_rank"], over)
            top_s[b], top_i[b] = cand[b, got_ids], got_ids   # the other search's order, the oracle's own running scores
        return top_s, top_i

    ref, hyps = go.beam_search(sd, ocfg, hiddens, masks, rd, multimodal, decoder_start_token_id=start_token, guide=guide, return_all=True, **kw)
    assert stat["steps"] == len(trace), ("the search under test ran %d steps, the oracle %d" % (len(trace), stat["steps"]))
    pad, eos = ocfg.pad_token_id, ocfg.eos_token_id
    for b in range(B):
        row = out_ids[b].tolist()
        if torch.equal(out_ids[b, :ref.shape[1]], ref[b]) and all(t == pad for t in row[ref.shape[1]:]):
            continue
        # not the oracle's pick: admissible only as another finished hypothesis whose score ties with the best (per-token tolerance)
        best = max(s for s, _ in hyps[b])
        ok = False
        for s, toks in hyps[b]:
            cand_row = toks + ([eos] if len(toks) < max_length else [])
            if row[:len(cand_row)] == cand_row and all(t == pad for t in row[len(cand_row):]) and best - s <= tie * len(toks):
                ok = True
        assert ok, ("business %d: the returned ids are not the oracle's best hypothesis nor one that ties with it" % b, row[:16], ref[b][:16])
    return stat
