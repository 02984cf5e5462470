"""GPU: the decode path beyond argmax through the C ABI.

* kernels: ctmi_row_lse / ctmi_group_topk / ctmi_scores_filter against torch fp32 on seeded inputs, including exact ties,
  -inf entries, bf16 inputs, strided logits views and vocabulary-sized rows (V = 250 880);
* beam search / n-gram greedy / samplers of ``generate`` on cuda:0 against the golden token ids the reference's own ``generate``
  produced (tests/golden/decode.npz) — bit-exact — and against the oracle (oracle/decode_ref.py) on a larger vocabulary.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import bloom_ref as R  # noqa: E402
from oracle import decode_ref as D  # noqa: E402
from oracle import gpt_ref as GR  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DEC = np.load(os.path.join(HERE, "golden", "decode.npz"))
BLOOM = (211, 64, 2, 8)
GPTS = (173, 64, 2, 4, 64)


def T(a):
    return torch.from_numpy(np.asarray(a))


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.array_equal(a, b)


def lex_topk(scores2d, k):
    """(value desc, index asc) reference: stable descending sort of the fp32 scores."""
    v, i = torch.sort(scores2d, dim=-1, descending=True, stable=True)
    return v[:, :k], i[:, :k]


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(6, 211), (4, 250880), (3, 1)])
def test_row_lse(rows, cols, dtype):
    from cleantransformer_amd import ops
    x = (torch.randn(rows, cols, generator=torch.Generator().manual_seed(3)) * 4).to(dtype).to(DEV)
    st = ops.row_lse(x)
    ref = torch.logsumexp(x.double(), dim=-1)
    assert torch.equal(st[:, 0], x.float().max(-1).values)
    assert torch.allclose(st[:, 0].double() + st[:, 1].double(), ref, rtol=1e-6, atol=2e-6)
    x3 = torch.stack([torch.zeros_like(x), x], dim=1)                      # [rows, 2, cols]: the last-position view, ld = 2*cols
    assert torch.equal(ops.row_lse(x3[:, -1, :]), st)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_group_topk_plain_ties_and_inf(dtype):
    from cleantransformer_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(5, 1000, generator=g) * 2).to(dtype)
    x[0, 10] = x[0, 700] = x[0].max() + 1                                  # an exact tie for first place
    x[1, :] = 0.5                                                           # a constant row: pure index order
    x[2, 3:] = -float("inf")                                                # fewer finite entries than k
    x[3, ::2] = -float("inf")
    xd = x.to(DEV)
    for k in (1, 7, 64):
        val, idx = ops.group_topk(xd, 1, k)
        rv, ri = lex_topk(x.float(), k)
        assert torch.equal(idx.cpu(), ri), k
        assert torch.equal(val.cpu(), rv), k
    # grouped: 2 groups of... 5 rows do not split evenly -> use 4 rows as 2 groups of 2
    val, idx = ops.group_topk(xd[:4], 2, 9)
    rv, ri = lex_topk(x[:4].float().reshape(2, -1), 9)
    assert torch.equal(idx.cpu(), ri) and torch.equal(val.cpu(), rv)


def test_group_topk_beam_scores_vocab_sized():
    """The fused form generation_util.py:199-217 needs: ((x - max) - logsum) + beam score, over beam * V candidates."""
    from cleantransformer_amd import ops
    beam, bsz, V = 4, 3, 250880
    g = torch.Generator().manual_seed(11)
    logits3 = (torch.randn(bsz * beam, 2, V, generator=g) * 3).to(DEV)       # [rows, S_new, V]: the kernel sees the last-position view
    last = logits3[:, -1, :]
    assert last.stride(0) == 2 * V
    beam_scores = (torch.randn(bsz, beam, generator=g) * 2).to(DEV)
    stats = ops.row_lse(last)
    val, idx = ops.group_topk(last, beam, 2 * beam, stats=stats, add=beam_scores.reshape(-1))
    ref = ((last - stats[:, 0:1]) - stats[:, 1:2]) + beam_scores.reshape(-1, 1)     # same fp32 operation order
    rv, ri = lex_topk(ref.view(bsz, -1).cpu(), 2 * beam)
    assert torch.equal(idx.cpu(), ri) and torch.equal(val.cpu(), rv)
    # and against torch.log_softmax itself: same candidates, scores within fp32 rounding
    tv, ti = (torch.log_softmax(last, -1) + beam_scores.reshape(-1, 1)).view(bsz, -1).topk(2 * beam)
    assert torch.equal(ti.cpu(), idx.cpu())
    assert torch.allclose(tv.cpu(), val.cpu(), rtol=0, atol=4e-6)


def test_scores_filter_and_processors_bit_exact():
    """The four processors on cuda:0 against the reference's outputs (decode.npz): bit-exact, -inf pattern included."""
    from cleantransformer_amd.generation.logits_processor import (NoRepeatNGramLogitsProcessor, TemperatureLogitsWrapper,
                                                                  TopKLogitsWrapper, TopPLogitsWrapper)
    sc, hist = T(DEC["lp_scores"]).to(DEV), T(DEC["lp_hist"]).to(DEV)
    for n in (2, 3):
        same(NoRepeatNGramLogitsProcessor(n)(hist, sc.clone()).cpu(), DEC[f"lp_ngram{n}"])
    same(TemperatureLogitsWrapper(0.7)(hist, sc.clone()).cpu(), DEC["lp_temp"])
    same(TemperatureLogitsWrapper(0.0)(hist, sc.clone()).cpu(), DEC["lp_temp_floor"])
    for k in (1, 10, 500):
        same(TopKLogitsWrapper(k)(hist, sc.clone()).cpu(), DEC[f"lp_topk{k}"])
    for p in (0.3, 0.8, 1.0, 0.0):
        same(TopPLogitsWrapper(p)(hist, sc.clone()).cpu(), DEC[f"lp_topp{p}"])
    # bf16 logits are widened exactly before filtering
    b = sc.to(torch.bfloat16)
    out = TopKLogitsWrapper(10)(hist, b)
    assert out.dtype == torch.float32
    same(out.cpu(), D.top_k(b.float().cpu(), 10))


# ------------------------------------------------------------------------------------------------ generate()
def _bloom(V=BLOOM[0], params=None):
    from test_gpu_bloom import build
    return build(V, *BLOOM[1:], params=params).eval()


def _gpt():
    from test_gpu_gpt import build
    return build(GR.GPTShape(*GPTS, version="gpt2")).eval()


def test_bloom_beam_search_bit_exact_vs_reference_golden():
    m = _bloom()
    ids, am = T(DEC["bloom_prompt"]).to(DEV), T(DEC["bloom_mask"]).to(DEV)
    ends = [int(e) for e in DEC["bloom_ends"]]
    gen = lambda **kw: m.generate(ids, attention_mask=am, generation_configs=dict(do_sample=False, pad_id=3, **kw)).cpu().numpy()   # noqa: E731
    same(gen(beam_size=3, max_gen_len=6, end_ids=[BLOOM[0] + 5]), DEC["bloom_beam3_free"])
    for es in (True, False):
        same(gen(beam_size=3, max_gen_len=6, end_ids=ends, early_stop=es), DEC[f"bloom_beam3_ends_es{int(es)}"])
    same(gen(beam_size=2, max_gen_len=8, end_ids=[BLOOM[0] + 5], no_repeat_ngram_size=2), DEC["bloom_beam2_ngram2"])
    rep = T(DEC["bloom_rep_prompt"]).to(DEV)
    for n in (0, 2):
        out = m.generate(rep, attention_mask=torch.ones_like(rep),
                         generation_configs=dict(beam_size=1, max_gen_len=8, do_sample=False, end_ids=None, pad_id=3, no_repeat_ngram_size=n))
        same(out.cpu().numpy(), DEC[f"bloom_greedy_ngram{n}"])


def test_gpt_beam_search_bit_exact_vs_reference_golden():
    m = _gpt()
    ids = T(DEC["gpt_prompt"]).to(DEV)
    am = torch.ones_like(ids)
    gen = lambda **kw: m.generate(ids, attention_mask=am, generation_configs=dict(do_sample=False, pad_id=3, beam_size=4, max_gen_len=6, **kw)).cpu().numpy()   # noqa: E731
    same(gen(end_ids=[GPTS[0] + 1]), DEC["gpt_beam4_free"])
    ends = [int(e) for e in DEC["gpt_ends"]]
    for es in (True, False):
        same(gen(end_ids=ends, early_stop=es), DEC[f"gpt_beam4_ends_es{int(es)}"])
    pos, seg = T(DEC["gpt_pos"]).to(DEV), T(DEC["gpt_seg"]).to(DEV)            # explicit position / segment ids (GPT only)
    out = m.generate(ids, attention_mask=am, position_ids=pos, segment_ids=seg,
                     generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    same(out.cpu().numpy(), DEC["gpt_greedy_posseg"])
    out = m.generate(ids, attention_mask=am, position_ids=pos, segment_ids=seg,
                     generation_configs=dict(beam_size=3, max_gen_len=6, do_sample=False, end_ids=[GPTS[0] + 1], pad_id=3))
    same(out.cpu().numpy(), DEC["gpt_beam3_posseg"])


def test_bloom_beam_search_vs_oracle_larger_vocab():
    """A vocabulary the golden file does not cover (V = 5003, beam 5, ragged left padding), oracle on the CPU beside it."""
    V = 5003
    sh = R.BloomShape(V, *BLOOM[1:])
    p = R.det_init(sh)
    m = _bloom(V, params=p)
    g = torch.Generator().manual_seed(21)
    ids = torch.randint(0, V, (4, 9), generator=g)
    am = torch.ones(4, 9, dtype=torch.long)
    am[2, :3] = 0
    free = D.beam_search(lambda i, a, ps: (lambda o: (o[1], o[3]))(R.bloom_forward(p, sh, i, a, None, ps)), 2, ids, am, 5, 5, [V + 1], pad_id=1)
    ends = sorted(set(int(t) for t in free[:, 0, 10:12].reshape(-1)))
    for e, es in (([V + 1], True), (ends, True), (ends, False)):
        ref = D.beam_search(lambda i, a, ps: (lambda o: (o[1], o[3]))(R.bloom_forward(p, sh, i, a, None, ps)), 2, ids, am, 5, 5, e, pad_id=1,
                            early_stop=es)
        out = m.generate(ids.to(DEV), attention_mask=am.to(DEV),
                         generation_configs=dict(beam_size=5, max_gen_len=5, do_sample=False, end_ids=e, pad_id=1, early_stop=es))
        same(out.cpu().numpy(), ref.numpy())


def test_sampling_paths_on_gpu():
    m = _bloom()
    ids, am = T(DEC["bloom_prompt"]).to(DEV), T(DEC["bloom_mask"]).to(DEV)
    greedy = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, pad_id=3))
    sampled = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=True, top_k=1, top_p=1.0,
                                                                         temperature=0.7, pad_id=3))
    same(sampled.cpu().numpy(), greedy.cpu().numpy())              # one surviving candidate: sampling == argmax
    free = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=2, max_gen_len=4, do_sample=True, top_k=5, top_p=0.9,
                                                                      temperature=1.3, end_ids=[BLOOM[0] + 5], pad_id=3))
    assert free.shape == (3, 2, 12) and int(free.max()) < BLOOM[0] and int(free.min()) >= 0
    # with top_k = 3 every sampled token must be one of the 3 most likely continuations of its own prefix
    torch.manual_seed(1)
    s3 = m.generate(ids, attention_mask=am, generation_configs=dict(beam_size=1, max_gen_len=3, do_sample=True, top_k=3, top_p=1.0, pad_id=3))
    seq = s3[:, 0, :]
    full_mask = torch.cat([am, am[:, -1:].expand(-1, seq.shape[1] - am.shape[1])], dim=1)
    (logits, _), _ = m(seq, attention_mask=full_mask)
    for t in range(ids.shape[1], seq.shape[1]):
        top3 = logits[:, t - 1, :].float().topk(3).indices
        assert bool((top3 == seq[:, t:t + 1]).any(dim=1).all()), t
