"""CPU (no GPU): the decode path beyond argmax — SURVEY §8(f)3.

1. the oracle (oracle/decode_ref.py) is PINNED to tests/golden/decode.npz, produced by the reference's own ``generate`` and
   logits-processor classes (tests/golden/make_golden.py decode);
2. the product's host logic (cleantransformer_amd/generation) is driven through the torch-CPU emulation of the kernel contracts
   and must reproduce the same token ids bit-exactly and the same filtered scores.
The kernels themselves (ctmi_row_lse / ctmi_group_topk / ctmi_scores_filter) are checked on the GPU box (tests/test_gpu_decode.py).
"""
import os

import numpy as np
import pytest
import torch

import cpu_kernel_emulation as emu
from oracle import bloom_ref as R
from oracle import decode_ref as D
from oracle import gpt_ref as GR

G = os.path.join(os.path.dirname(__file__), "golden")
DEC = np.load(os.path.join(G, "decode.npz"))
BLOOM = (211, 64, 2, 8)
GPTS = (173, 64, 2, 4, 64)


def T(a):
    return torch.from_numpy(np.asarray(a))


def same(a, b):
    """bit-exact including the -inf pattern"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.array_equal(a, b), float(np.nanmax(np.abs(np.where(np.isfinite(a) & np.isfinite(b), a - b, 0))))


# ------------------------------------------------------------------------------------------------ 1. oracle pinned to the reference
def test_oracle_logits_processors_match_reference():
    sc, hist = T(DEC["lp_scores"]), T(DEC["lp_hist"])
    for n in (2, 3):
        same(D.no_repeat_ngram(hist, sc, n), DEC[f"lp_ngram{n}"])
    assert np.isinf(DEC["lp_ngram2"]).sum() > 0                         # the fixture does ban something
    same(D.temperature(sc, 0.7), DEC["lp_temp"])
    same(D.temperature(sc, 0.0), DEC["lp_temp_floor"])
    for k in (1, 10, 500):
        same(D.top_k(sc, k), DEC[f"lp_topk{k}"])
    for p in (0.3, 0.8, 1.0, 0.0):
        same(D.top_p(sc, p), DEC[f"lp_topp{p}"])


def _bloom_step():
    sh = R.BloomShape(*BLOOM)
    p = R.det_init(sh)

    def step(ids, mask, pasts):
        _, logits, _, pasts = R.bloom_forward(p, sh, ids, mask, None, pasts)
        return logits, pasts
    return step


def _gpt_step():
    s = GR.GPTShape(*GPTS, version="gpt2")
    p = GR.det_init(s)

    def step(ids, mask, pasts, position_ids=None, segment_ids=None):
        _, logits, _, pasts = GR.gpt_forward(p, s, ids, mask, None, pasts, position_ids=position_ids, segment_ids=segment_ids)
        return logits, pasts
    return step


def test_oracle_bloom_beam_search_matches_reference():
    step = _bloom_step()
    ids, am = T(DEC["bloom_prompt"]), T(DEC["bloom_mask"])
    ends = [int(e) for e in DEC["bloom_ends"]]
    same(D.beam_search(step, 2, ids, am, 3, 6, [BLOOM[0] + 5], pad_id=3), DEC["bloom_beam3_free"])
    for es in (True, False):
        same(D.beam_search(step, 2, ids, am, 3, 6, ends, pad_id=3, early_stop=es), DEC[f"bloom_beam3_ends_es{int(es)}"])
    same(D.beam_search(step, 2, ids, am, 2, 8, [BLOOM[0] + 5], pad_id=3, no_repeat_ngram_size=2), DEC["bloom_beam2_ngram2"])
    rep = T(DEC["bloom_rep_prompt"])
    for n in (0, 2):
        same(D.greedy_ngram(step, rep, torch.ones_like(rep), 8, n), DEC[f"bloom_greedy_ngram{n}"])


def test_oracle_gpt_beam_search_matches_reference():
    step = _gpt_step()
    ids = T(DEC["gpt_prompt"])
    am = torch.ones_like(ids)
    same(D.beam_search(step, 2, ids, am, 4, 6, [GPTS[0] + 1], pad_id=3), DEC["gpt_beam4_free"])
    ends = [int(e) for e in DEC["gpt_ends"]]
    for es in (True, False):
        same(D.beam_search(step, 2, ids, am, 4, 6, ends, pad_id=3, early_stop=es), DEC[f"gpt_beam4_ends_es{int(es)}"])
    # explicit position / segment ids are carried and extended by both searches (generation_util.py:98-99, :268-271)
    pos, seg = T(DEC["gpt_pos"]), T(DEC["gpt_seg"])
    same(D.greedy_ngram(step, ids, am, 6, 0, position_ids=pos, segment_ids=seg), DEC["gpt_greedy_posseg"])
    same(D.beam_search(step, 2, ids, am, 3, 6, [GPTS[0] + 1], pad_id=3, position_ids=pos, segment_ids=seg), DEC["gpt_beam3_posseg"])
    assert not np.array_equal(DEC["gpt_greedy_posseg"], np.load(os.path.join(G, "tiny_gpt.npz"))["greedy_out"])   # the ids do matter


# ------------------------------------------------------------------------------------------------ 2. product host logic
def _bloom_model():
    from test_host_logic_cpu import build
    return build(*BLOOM).eval()


def _gpt_model():
    from cleantransformer_amd.models.modeling_gpt import GPTConfig, GPTLMHeadModel
    V, H, L, nh, P = GPTS
    m = GPTLMHeadModel(GPTConfig(vocab_size=V, n_embd=H, n_positions=P, n_layer=L, n_head=nh, n_ctx=P, embd_pdrop=0.0, attn_pdrop=0.0,
                                 resid_pdrop=0.0), version="gpt2")
    sd = dict(GR.det_init(GR.GPTShape(*GPTS, version="gpt2")))
    sd["lm_head.weight"] = sd["gpt.tokens_embed.weight"]
    m.load_state_dict(sd, strict=False)
    m._tie_weights()
    return m.eval()


def test_logits_processors_host_logic_bit_exact(monkeypatch):
    emu.install(monkeypatch)
    from CleanTransformer.generation.logits_processor import (NoRepeatNGramLogitsProcessor, TemperatureLogitsWrapper,
                                                              TopKLogitsWrapper, TopPLogitsWrapper)
    sc, hist = T(DEC["lp_scores"]), T(DEC["lp_hist"])
    for n in (2, 3):
        same(NoRepeatNGramLogitsProcessor(n)(hist, sc.clone()), DEC[f"lp_ngram{n}"])
    same(NoRepeatNGramLogitsProcessor(1)(hist, sc.clone()), DEC["lp_scores"])     # n = 1 never bans (the reference's lookup key)
    same(TemperatureLogitsWrapper(0.7)(hist, sc.clone()), DEC["lp_temp"])
    same(TemperatureLogitsWrapper(0.0)(hist, sc.clone()), DEC["lp_temp_floor"])
    for k in (1, 10, 500):
        same(TopKLogitsWrapper(k)(hist, sc.clone()), DEC[f"lp_topk{k}"])
    for p in (0.3, 0.8, 1.0, 0.0):
        same(TopPLogitsWrapper(p)(hist, sc.clone()), DEC[f"lp_topp{p}"])


def test_bloom_beam_search_host_logic_bit_exact(monkeypatch):
    emu.install(monkeypatch)
    m = _bloom_model()
    ids, am = T(DEC["bloom_prompt"]), T(DEC["bloom_mask"])
    ends = [int(e) for e in DEC["bloom_ends"]]
    gen = lambda **kw: m.generate(ids, attention_mask=am, generation_configs=dict(do_sample=False, pad_id=3, **kw)).numpy()   # noqa: E731
    same(gen(beam_size=3, max_gen_len=6, end_ids=[BLOOM[0] + 5]), DEC["bloom_beam3_free"])
    for es in (True, False):
        same(gen(beam_size=3, max_gen_len=6, end_ids=ends, early_stop=es), DEC[f"bloom_beam3_ends_es{int(es)}"])
    same(gen(beam_size=2, max_gen_len=8, end_ids=[BLOOM[0] + 5], no_repeat_ngram_size=2), DEC["bloom_beam2_ngram2"])
    rep = T(DEC["bloom_rep_prompt"])
    for n in (0, 2):
        out = m.generate(rep, attention_mask=torch.ones_like(rep),
                         generation_configs=dict(beam_size=1, max_gen_len=8, do_sample=False, end_ids=None, pad_id=3, no_repeat_ngram_size=n))
        same(out.numpy(), DEC[f"bloom_greedy_ngram{n}"])
    with pytest.raises(TypeError):                                             # the reference's behaviour for end_ids=None
        m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=3, do_sample=False))


def test_gpt_beam_search_host_logic_bit_exact(monkeypatch):
    emu.install(monkeypatch)
    m = _gpt_model()
    ids = T(DEC["gpt_prompt"])
    am = torch.ones_like(ids)
    gen = lambda **kw: m.generate(ids, attention_mask=am, generation_configs=dict(do_sample=False, pad_id=3, beam_size=4, max_gen_len=6, **kw)).numpy()   # noqa: E731
    same(gen(end_ids=[GPTS[0] + 1]), DEC["gpt_beam4_free"])
    ends = [int(e) for e in DEC["gpt_ends"]]
    for es in (True, False):
        same(gen(end_ids=ends, early_stop=es), DEC[f"gpt_beam4_ends_es{int(es)}"])
    pos, seg = T(DEC["gpt_pos"]), T(DEC["gpt_seg"])
    out = m.generate(ids, attention_mask=am, position_ids=pos, segment_ids=seg,
                     generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    same(out.numpy(), DEC["gpt_greedy_posseg"])
    out = m.generate(ids, attention_mask=am, position_ids=pos, segment_ids=seg,
                     generation_configs=dict(beam_size=3, max_gen_len=6, do_sample=False, end_ids=[GPTS[0] + 1], pad_id=3))
    same(out.numpy(), DEC["gpt_beam3_posseg"])


def test_sampling_paths_host_logic(monkeypatch):
    """Sampled ids depend on the RNG stream; what is checkable: top_k = 1 leaves one candidate, so sampling must equal argmax
    decoding (greedy) and the sampled beam search must emit tokens the deterministic beam search ranks first; streamers stop
    the loop; outputs have the reference's shapes."""
    emu.install(monkeypatch)
    m = _bloom_model()
    ids, am = T(DEC["bloom_prompt"]), T(DEC["bloom_mask"])
    torch.manual_seed(0)
    greedy = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, pad_id=3))
    sampled = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=True, top_k=1, top_p=1.0,
                                                                         temperature=0.7, pad_id=3))
    same(sampled.numpy(), greedy.numpy())
    free = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=2, max_gen_len=4, do_sample=True, top_k=5, top_p=0.9,
                                                                      temperature=1.3, end_ids=[BLOOM[0] + 5], pad_id=3))
    assert free.shape == (3, 2, 6 + 4 + 2) and int(free.max()) < BLOOM[0] and int(free.min()) >= 0
    assert np.array_equal(free[:, :, :6].numpy(), np.repeat(DEC["bloom_prompt"][:, None, :], 2, axis=1))
    seen = []
    out = m.generate(ids, attention_mask=am, steamers=lambda x: (seen.append(tuple(x.shape)), len(seen) >= 2)[1],
                     generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, pad_id=3))
    assert seen == [(3, 1, 7), (3, 1, 8)] and out.shape == (3, 1, 8)
