"""CPU property tests (hypothesis) of small host-side rules the decode / trainer widening added: they are restated from the
reference line by line in oracle/decode_ref.py (pinned to the reference's outputs) or from the formulas the reference delegates
to, and checked here on random inputs rather than on a handful of fixtures."""
import os

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import cpu_kernel_emulation as emu
from oracle import decode_ref as D


class _Patch:
    def __init__(self):
        self.undo = []

    def setattr(self, obj, name, val):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, val)

    def restore(self):
        for obj, name, val in reversed(self.undo):
            setattr(obj, name, val)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 5), min_size=0, max_size=14), st.integers(1, 5))
def test_ngram_ban_equals_the_reference_restatement(tokens, n):
    """NoRepeatNGramLogitsProcessor.banned_tokens == the dictionary walk of logits_processor.py:18-30 (oracle), any length / n."""
    from cleantransformer_amd.generation.logits_processor import NoRepeatNGramLogitsProcessor
    got = sorted(set(NoRepeatNGramLogitsProcessor.banned_tokens(list(tokens), n)))
    if len(tokens) == 0:
        assert got == []
        return
    ids = torch.tensor([tokens])
    ref = D.no_repeat_ngram(ids, torch.zeros(1, 6), n)[0]
    assert got == sorted(int(i) for i in torch.isinf(ref).nonzero().flatten())


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 40), st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 3, 17]))
def test_topk_and_temperature_wrappers_equal_the_reference_restatement(k, seed, rows):
    p = _Patch()
    emu.install(p)
    try:
        from cleantransformer_amd.generation.logits_processor import TemperatureLogitsWrapper, TopKLogitsWrapper
        g = torch.Generator().manual_seed(seed)
        sc = torch.randn(rows, 33, generator=g) * 2
        sc[:, 5] = sc[:, 9]                                                  # ties at arbitrary rank
        assert torch.equal(TopKLogitsWrapper(k)(None, sc.clone()), D.top_k(sc, k))
        t = float(torch.rand(1, generator=g)) * 2
        assert torch.equal(TemperatureLogitsWrapper(t)(None, sc.clone()), D.temperature(sc, t))
    finally:
        p.restore()


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 50), st.integers(0, 20), st.sampled_from(["linear", "constant"]))
def test_schedule_matches_the_closed_form(total, warmup, kind):
    """transformers' get_linear_schedule_with_warmup / get_constant_schedule_with_warmup lambdas (what trainer.py:854-865 builds)."""
    from cleantransformer_amd.trainer.trainer import _Schedule

    class Opt:
        lr = 2e-3
    o = Opt()
    sch = _Schedule(o, kind, warmup, total)
    for s in range(total + 3):
        if s < warmup:
            f = s / max(1, warmup)
        elif kind == "constant":
            f = 1.0
        else:
            f = max(0.0, (total - s) / max(1, total - warmup))
        assert abs(sch.get_last_lr()[0] - 2e-3 * f) < 1e-15, (s, kind)
        sch.step()
    sd = sch.state_dict()
    o2 = Opt()
    s2 = _Schedule(o2, kind, warmup, total)
    s2.load_state_dict(sd)
    assert s2.get_last_lr() == sch.get_last_lr()


@settings(max_examples=40, deadline=None)
@given(steps=st.lists(st.integers(1, 500), min_size=1, max_size=9, unique=True), limit=st.integers(1, 4), protect_best=st.booleans())
def test_checkpoint_rotation_keeps_newest_and_best(steps, limit, protect_best):
    """_rotate_checkpoints (trainer.py:1465-1511): numeric order (not lexical), newest `limit` kept, the best model never deleted."""
    from cleantransformer_amd.trainer.trainer import Trainer, TrainingArguments
    import pathlib
    import tempfile
    root = pathlib.Path(tempfile.mkdtemp(prefix="ctmi_rot_"))
    for s in steps:
        os.makedirs(root / f"checkpoint-{s}")
    (root / "checkpoint-notanumber").mkdir()
    tr = Trainer.__new__(Trainer)
    tr.args = TrainingArguments(output_dir=str(root), save_total_limit=limit)
    from cleantransformer_amd.trainer.trainer import TrainerState
    tr.state = TrainerState()
    best = min(steps)
    if protect_best:
        tr.state.best_model_checkpoint = str(root / f"checkpoint-{best}")
    tr._rotate_checkpoints(use_mtime=False, output_dir=str(root))
    left = sorted(int(d.split("-")[1]) for d in os.listdir(root) if d.split("-")[1].isdigit())
    want = sorted(steps)[-limit:]
    if protect_best and best not in want:
        if limit == 1:
            want = [best] + want                       # save_total_limit = 1 with a best model elsewhere keeps two (:1477-1483)
        else:
            want = [best] + want[1:]                   # the best one takes the place of the oldest survivor (:1504-1510)
    assert left == sorted(want), (steps, limit, protect_best, left)
    assert os.path.isdir(root / "checkpoint-notanumber")
    import shutil
    shutil.rmtree(root, ignore_errors=True)


def test_beam_book_matches_reference_candidate_rules():
    """_BeamBook.add == generation_util.py:148-157 on a fixed sequence: worst score tracking, eviction of the lowest, ties by age."""
    from cleantransformer_amd.generation.generation_util import _BeamBook
    b = _BeamBook()
    cands, worst = [], 1e9
    seq = [np.float32(x) for x in (-1.5, -0.5, -0.5, -2.0, -0.25, -0.5, -3.0)]
    for i, sc in enumerate(seq):
        b.add(i, sc, beam=3)
        cands.append({"ids": i, "score": sc})                       # the reference's list-of-dicts walk
        if len(cands) > 3:
            ranked = sorted((c["score"], j) for j, c in enumerate(cands))
            del cands[ranked[0][1]]
            worst = ranked[1][0]
        else:
            worst = min(sc, worst)
        assert [ids for _, ids in b.finished] == [c["ids"] for c in cands] and b.worst == worst, i
