"""GPU: beam-search generation (BASELINE config 5, test.py:137-164) held to the CPU restatement of the reference's
_generate_beam_search (oracle/generate_oracle.py, pinned to the reference's own generate() by tests/golden/g1_beam.npz)
past the sizes tests/test_modules_gpu.py and tests/test_bench_shapes_gpu.py cover:

  * max_length 128 at BART-large width (decode positions 24 .. 127: the ancestor table over 126 steps, n-gram bans on long
    prefixes, the caches' Tmax walk), f32 compute mode;
  * the full depth (12 + 12 layers) in f32;
  * the mode `bench.py --workload generate` times (bf16 kernels; since round 4 the last LayerNorm's output and the logits stay
    f32): how many of the generated tokens equal the oracle's, beside the step-by-step score check of tests/test_timed_path_gpu.py;
  * max_length 256, the decode self-attention kernel's limit (mmsum_decode_self_attn: Tmax <= 256).

What "token ids equal" can mean over 127 steps, and the check that replaces it where ties exist: tests/gen_check.py (the oracle's
search, guided through the hypotheses the HIP search visited: same hypotheses at every step, every returned candidate carries the oracle's
score within TIE, nothing better was passed over, the final ids are the oracle's).  Beside it each test reports how many leading tokens
equal the oracle's INDEPENDENT run (reported), and demands all of them when the oracle's own run met no near-tie at all.
"""
import os

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from multimodalsum_amd import synthetic as syn
from multimodalsum_amd.config import BartConfig
from multimodalsum_amd.formula_init import formula_state_dict, formula_tensor
from oracle import bart_oracle as bo
from oracle import generate_oracle as go

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bart_large(layers):
    cfg = BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    cfg.dropout = 0.0
    cfg.encoder_layers = cfg.decoder_layers = layers
    return cfg


def _setup(layers, Bz, dtype, std=0.06, S=128):
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    cfg = _bart_large(layers)
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=layers, decoder_layers=layers,
                      heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.0)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=std)
    model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=dtype, deterministic=dtype == torch.float32)
    model.load_state_dict(sd)
    model.eval()
    N = 8
    ids = syn.token_batch(Bz * N, S, cfg.vocab_size, seed=21, mean_len=75.0 * S / 128, std_len=20.0 * S / 128, min_len=S // 4).view(Bz, N, S)
    text_m = ids.ne(1).clone()
    table_h = formula_tensor("g.table_h", (Bz, 1, 47, cfg.d_model), std=1.0)
    img_h = formula_tensor("g.img_h", (Bz, 2, 196, cfg.d_model), std=1.0)
    table_m = torch.ones(Bz, 1, 47, dtype=torch.bool)
    img_m = torch.ones(Bz, 2, 196, dtype=torch.bool)
    img_m[Bz - 1, 1] = False
    return cfg, ocfg, sd, model, ids, text_m, table_h, table_m, img_h, img_m


def _run_and_check(layers, Bz, max_length, dtype, tie, enc_tol, independent=True, S=128):
    """-> (leading tokens of the HIP output equal to the oracle's independent run per business, generated length, guided-check statistics)."""
    from tests.gen_check import guided_check
    cfg, ocfg, sd, model, ids, text_m, table_h, table_m, img_h, img_m = _setup(layers, Bz, dtype, S=S)
    N, beams = 8, 4
    kw = dict(num_beams=beams, max_length=max_length, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0)
    rd = torch.zeros(Bz, 1)
    cast = (lambda t: t.to(DEV).to(dtype))
    trace = []
    with torch.no_grad():
        enc = model.model.encoder(input_ids=ids.view(-1, S).to(DEV), attention_mask=text_m.view(-1, S).to(DEV))[0].view(Bz, N, S, -1)
        oenc = bo.bart_encoder(sd, ocfg, ids.view(-1, S), text_m.view(-1, S)).view(Bz, N, S, -1)
        valid = text_m.view(Bz, N, S, 1).float()
        assert float(((enc.float().cpu() - oenc) * valid).abs().max()) <= enc_tol * float(oenc.abs().max())
        out = model.generate(enc, text_m.to(DEV), cast(table_h), table_m.to(DEV), cast(img_h), img_m.to(DEV), rating_diff=rd.to(DEV),
                             decoder_start_token_id=cfg.bos_token_id, trace=trace, **kw).cpu()
        assert out.shape[0] == Bz and len(trace) == max_length - 1, (out.shape, len(trace))       # random weights never end early
        hid, msk = [oenc, table_h, img_h], [text_m, table_m, img_m]
        st = guided_check(out, trace, sd, ocfg, hid, msk, rd, True, kw, tie=tie, start_token=cfg.bos_token_id)
        same, L = None, out.shape[1]
        if independent:
            margins = []
            ref = go.beam_search(sd, ocfg, hid, msk, rd, True, decoder_start_token_id=cfg.bos_token_id, margins=margins, **kw)
            L = min(out.shape[1], ref.shape[1])
            same = [int((out[b, :L] == ref[b, :L]).long().cumprod(0).sum()) for b in range(Bz)]
            clear = next((i for i, m in enumerate(margins) if m < 4 * tie), len(margins))
            if clear == len(margins):           # the oracle's own ranking never came near a tie: the independent runs must agree outright
                assert min(same) == L, ("ids differ from the oracle's independent run although it met no near-tie", same, L)
            # (a near-tie at ANY step can change which beam survives, and with it earlier tokens of the final hypothesis: a prefix rule would
            # be unsound; where near-ties occur the guided check above is what holds the search, and the fixture of
            # test_generation_ids_equal_the_reference_at_config5_size is the case chosen to have none)
            st.update(clear=clear, oracle_steps=len(margins))
    print("generation %d+%d layers, max_length %d, %s: guided check over %d steps: worst candidate-score deviation %.2e nats, worst passed-over "
          "margin %.2e; ids equal to the oracle's independent run for %s of %d tokens (its first margin below %.0e at step %s)"
          % (layers, layers, max_length, str(dtype).replace("torch.", ""), st["steps"], st["worst_score"], st["worst_rank"], same, L, 4 * tie,
             st.get("clear")))
    return same, L, st


def test_generation_f32_max_length_128_at_bart_large_width():
    """BASELINE config 5's lengths: 8 reviews x 128 tokens + table + images, num_beams 4, no_repeat_ngram_size 3, max_length 128, at
    BART-large width (D 1024, H 16, F 4096, V 50265; 2 + 2 layers), f32 compute mode.  TIE = 5e-4 nats (measured deviation 6e-5: f32 logits
    of two summation orders accumulated into running scores over 127 steps)."""
    same, L, st = _run_and_check(2, 2, 128, torch.float32, tie=5e-4, enc_tol=1e-3)
    assert L >= 100 and st["steps"] == 127 and min(same) >= 4


def test_generation_f32_on_158_token_reviews():
    """test.py's own Yelp inputs (src/test.py:56-60: max_length 160 -> 158 tokens per review after the strip at data_utils.py:48-52): the
    encoder over [16, 158] (two query blocks per sequence over 158 keys), the decode step's cross-attention over 158-key text entities +
    table + images, 4 beams, max_length 64, f32 compute mode.  Same checks as the 128-token case above."""
    same, L, st = _run_and_check(2, 2, 64, torch.float32, tie=5e-4, enc_tol=1e-3, S=158)
    assert L >= 48 and st["steps"] == 63 and min(same) >= 4


def test_generation_f32_at_full_depth():
    """The 12 + 12-layer model (cfg/bart-large.json as it is), one business, max_length 32, f32 compute mode: the decode path at the
    depth test.py runs (24 self-attention caches, 12 cross-attention K / V sets, the cache walk through every layer).  TIE = 1e-2 (measured 1.1e-3):
    the post-LN stack amplifies f32 rounding ~1e4 times at this depth -- measured: scores of -50 that differ from
    the oracle's by a common 5e-3 (1e-4 relative), i.e. the running beam score, while the candidates' order is the oracle's."""
    same, L, st = _run_and_check(12, 1, 32, torch.float32, tie=1e-2, enc_tol=2e-3)
    assert L >= 24 and st["steps"] == 31


def test_generation_bf16_guided_against_the_oracle():
    """The timed mode (bf16 kernels, f32 final LayerNorm output and f32 logits) is NOT held to token-id equality: its candidate scores are
    ~0.25 nats from the fp32 oracle's (measured), more than most ranking gaps.  What is held: the step-by-step guided rule of
    tests/test_timed_path_gpu.py at TIE 0.35 nats over max_length 64 (same hypotheses at every step, every returned candidate within TIE of the
    oracle's score, nothing better than TIE passed over), and the first four tokens, which are far from any tie; the count of leading tokens
    equal to the oracle's independent run is reported.  Token-id equality is the f32 mode's property
    (test_generation_ids_equal_the_reference_at_config5_size)."""
    same, L, st = _run_and_check(2, 2, 64, torch.bfloat16, tie=0.35, enc_tol=5e-2)
    assert min(same) >= 4 and st["steps"] == 63


def test_generation_max_length_256():
    """mmsum_decode_self_attn walks up to Tmax = 256 cache positions (four per lane): a 255-step search on a small model (D 256, 2 + 2
    layers, three reviews of 8 tokens + table + images; 2 beams, EOS banned until length 250 so that it cannot end early) passes the
    guided check, so positions 224 .. 255 -- the fourth key per lane, the ancestor table's last columns, bans over 250-token prefixes
    -- are compared too.  f32 mode.  (tests/test_host_logic_cpu.py runs the same search through the kernel emulator.)"""
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    from tests.gen_check import guided_check
    from tests.test_host_logic_cpu import tiny_cfg, oracle_cfg
    cfg = tiny_cfg(vocab=400, d=256, ffn=128, layers=2, heads=4, maxpos=300)
    ocfg = oracle_cfg(cfg)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=0.08)
    model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=torch.float32, deterministic=True)
    model.load_state_dict(sd)
    model.eval()
    Bz, N, S = 2, 3, 8
    ids = syn.token_batch(Bz * N, S, cfg.vocab_size, seed=11, min_len=3).view(Bz, N, S)
    text_m = ids.ne(1).clone()
    table_h = formula_tensor("t.table_h", (Bz, 1, 6, cfg.d_model), std=1.0)
    img_h = formula_tensor("t.img_h", (Bz, 2, 4, cfg.d_model), std=1.0)
    table_m = torch.ones(Bz, 1, 6, dtype=torch.bool)
    img_m = torch.ones(Bz, 2, 4, dtype=torch.bool)
    img_m[1, 1] = False
    kw = dict(num_beams=2, max_length=256, min_length=250, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0)
    rd = torch.zeros(Bz, 1)
    trace = []
    with torch.no_grad():
        enc = model.model.encoder(input_ids=ids.view(-1, S).to(DEV), attention_mask=text_m.view(-1, S).to(DEV))[0].view(Bz, N, S, -1)
        oenc = bo.bart_encoder(sd, ocfg, ids.view(-1, S), text_m.view(-1, S)).view(Bz, N, S, -1)
        out = model.generate(enc, text_m.to(DEV), table_h.to(DEV), table_m.to(DEV), img_h.to(DEV), img_m.to(DEV), rating_diff=rd.to(DEV),
                             decoder_start_token_id=cfg.bos_token_id, trace=trace, **kw).cpu()
        assert len(trace) >= 249, len(trace)                       # the search really reached the last cache positions
        st = guided_check(out, trace, sd, ocfg, [oenc, table_h, img_h], [text_m, table_m, img_m], rd, True, kw, tie=1e-3, start_token=cfg.bos_token_id)
    print("max_length 256: guided check over %d steps, worst candidate-score deviation %.2e nats" % (st["steps"], st["worst_score"]))


def test_generation_ids_equal_the_reference_at_config5_size(golden_dir):
    """VERDICT r4 item 3a.  BASELINE config 5 at its real size -- cfg/bart-large.json (12 + 12 layers), 4 beams, max_length 128,
    no_repeat_ngram_size 3, two businesses x (8 reviews x 128 + 47 table + 4 x 196 image) memory rows -- against the token ids the REFERENCE's
    own generate() returned (tests/golden/g2_generate_full.npz, oracle/make_golden_r5.py; modeling_multimodalsum.py:2803-3067,
    test.py:156-158): `torch.equal` in the f32 compute mode.  Weights: the formula init at the reference's init_std 0.02 -- at the 0.06 of
    the tests above the post-LN stack is chaotic in f32 (the oracle's own f32 and f64 log-probabilities differ by nats), at 0.02 they
    agree to 3e-6 nats while the fixture's smallest gap at a decision that changes the result is `decision_margin` (~5e-4 nats; the input
    seed with the widest one of those tried).  The timed bf16 mode on the same inputs: the count of equal leading tokens is reported
    (measured [4, 24] of 127: bf16 scores are ~0.2 nats from the f32 ones, more than most ranking gaps) and at least the first 4 must agree."""
    from multimodalsum_amd.config import BartConfig
    from multimodalsum_amd.modules import BartForMultiEncConditionalGeneration
    from oracle.gen_fixture import G2, g2_inputs, g2_kwargs
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(golden_dir, "g2_generate_full.npz")).items()}
    cfg = BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=cfg.encoder_layers,
                      decoder_layers=cfg.decoder_layers, heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.0)
    sd = formula_state_dict(bo.bart_param_shapes(ocfg, True, prefix=""), std=G2["std"])
    text_h, text_m, table_h, table_m, img_h, img_m = g2_inputs(cfg, int(g["seed"]))
    assert float(g["decision_margin"]) >= 1e-4
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        model = BartForMultiEncConditionalGeneration(cfg, device=DEV, dtype=dtype, deterministic=(dtype == torch.float32))
        model.load_state_dict(sd)
        model.eval()
        cast = (lambda t: t.to(DEV).to(dtype))
        with torch.no_grad():
            out = model.generate(cast(text_h), text_m.to(DEV), cast(table_h), table_m.to(DEV), cast(img_h), img_m.to(DEV),
                                 rating_diff=torch.zeros(G2["B"], 1, device=DEV), decoder_start_token_id=cfg.decoder_start_token_id, **g2_kwargs()).cpu()
        res[dtype] = out
        del model
        torch.cuda.empty_cache()
    ref = g["ids"]
    f32, b16 = res[torch.float32], res[torch.bfloat16]
    L = min(b16.shape[1], ref.shape[1])
    same16 = [int((b16[b, :L] == ref[b, :L]).long().cumprod(0).sum()) for b in range(ref.shape[0])]
    print("config 5 at its real size: f32 ids equal to the reference's: %s (%d tokens per business); bf16: %s of %d leading tokens equal"
          % (bool(torch.equal(f32, ref)), ref.shape[1], same16, L))
    assert torch.equal(f32, ref), (f32[:, :16], ref[:, :16], [int((f32[b] == ref[b]).long().cumprod(0).sum()) for b in range(ref.shape[0])])
    assert min(same16) >= 4, same16


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_generation_with_more_than_64_hypothesis_rows(dtype):
    """17 businesses x 4 beams = 68 hypothesis rows -- past the 32 rows of the decode step's own kernels (the general step runs) and
    past the 64 rows one call of the f32-activation LM head takes (bf16 mode: mmsum_gemm's MMSUM_GEMM_A_F32 form goes through in row chunks;
    before round 5 this raised MMSUM_ERR_BAD_DTYPE).  The businesses alternate between two inputs: every copy of an input must come out with
    the same ids (hypothesis rows do not interact), equal to the ids a two-business call returns for it on the same path-independent f32 mode;
    in bf16 the two-business call takes the fast step, whose rounding differs: compared over the first four tokens only."""
    cfg, ocfg, sd, model, ids, text_m, table_h, table_m, img_h, img_m = _setup(2, 2, dtype)
    N, S, Bz = 8, 128, 17
    kw = dict(num_beams=4, max_length=12, no_repeat_ngram_size=3, early_stopping=True, length_penalty=1.0)
    pick = torch.arange(Bz) % 2
    cast = (lambda t: t.to(DEV).to(dtype))
    with torch.no_grad():
        enc2 = model.model.encoder(input_ids=ids.view(-1, S).to(DEV), attention_mask=text_m.view(-1, S).to(DEV))[0].view(2, N, S, -1)
        small = model.generate(enc2, text_m.to(DEV), cast(table_h), table_m.to(DEV), cast(img_h), img_m.to(DEV), rating_diff=torch.zeros(2, 1, device=DEV),
                               decoder_start_token_id=cfg.bos_token_id, **kw).cpu()
        big = model.generate(enc2[pick.to(DEV)].contiguous(), text_m[pick].to(DEV), cast(table_h[pick]), table_m[pick].to(DEV), cast(img_h[pick]), img_m[pick].to(DEV),
                             rating_diff=torch.zeros(Bz, 1, device=DEV), decoder_start_token_id=cfg.bos_token_id, **kw).cpu()
    assert big.shape[0] == Bz
    for b in range(Bz):
        assert torch.equal(big[b], big[int(pick[b])]), (b, big[b], big[int(pick[b])])              # copies of one input agree among themselves
    L = min(big.shape[1], small.shape[1])
    n = L if dtype == torch.float32 else 4
    assert torch.equal(big[:2, :n], small[:, :n]), (big[:2, :L], small[:, :L])

