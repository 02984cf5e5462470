"""CPU restatement (plain torch CPU ops, python loops) of the reference's decode path beyond argmax:
beam search (CleanTransformer/generation/generation_util.py:121-290) and the logits processors (logits_processor.py).

TEST INFRASTRUCTURE — the oracle for ``cleantransformer_amd/generation``; never imported by the product package.

Parity status: PINNED by ``tests/test_decode_cpu.py::test_oracle_*`` against ``tests/golden/decode.npz``, produced in the build
container by the reference's own ``generate`` / processor classes (``tests/golden/make_golden.py decode``).

The model is abstracted as ``step_fn(ids_new, attention_mask, pasts[, position_ids=, segment_ids=]) -> (logits [rows, S_new, V],
pasts)`` so that the same search runs over ``oracle.bloom_ref`` and ``oracle.gpt_ref``.  Only the deterministic configuration (``do_sample=False``) is
restated: sampled ids depend on the RNG stream.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------ logits_processor.py
def no_repeat_ngram(input_ids: Tensor, scores: Tensor, n: int) -> Tensor:
    """logits_processor.py:15-32: ban every token that would repeat an n-gram already in the row's history."""
    scores = scores.clone()
    for i in range(input_ids.shape[0]):
        toks = input_ids[i].tolist()
        seen = {}
        for j in range(len(toks) - n + 1):
            seen.setdefault(tuple(toks[j:j + n - 1]), []).append(toks[j + n - 1])
        for t in seen.get(tuple(toks[-n + 1:]), []):
            scores[i, t] = -float("inf")
    return scores


def temperature(scores: Tensor, t: float) -> Tensor:
    """logits_processor.py:35-41 (floor 1e-2)."""
    return scores / max(t, 1e-2)


def top_k(scores: Tensor, k: int, fill: float = -float("inf")) -> Tensor:
    """logits_processor.py:44-56: keep the k largest (and everything tied with the k-th)."""
    k = min(int(max(k, 1)), scores.shape[-1])
    kth = scores.topk(k, dim=-1).values[..., -1, None]
    return scores.masked_fill(scores < kth, fill)


def top_p(scores: Tensor, p: float, fill: float = -float("inf"), min_keep: int = 1) -> Tensor:
    """logits_processor.py:59-79: drop the ascending-sorted tail whose cumulative probability is <= 1 - p."""
    p = max(min(p, 1.0), 0)
    srt, idx = torch.sort(scores, descending=False)
    rm = srt.softmax(-1).cumsum(-1) <= (1 - p)
    rm[..., -max(1, min_keep):] = False
    return scores.masked_fill(rm.scatter(1, idx, rm), fill)


# ------------------------------------------------------------------------------------------------ generation_util.py:121-290
def _extra(position_ids, segment_ids, step):
    kw = {}
    if position_ids is not None:
        kw["position_ids"] = position_ids[:, step:]
    if segment_ids is not None:
        kw["segment_ids"] = segment_ids[:, step:]
    return kw


def beam_search(step_fn: Callable, n_layer: int, input_ids: Tensor, attention_mask: Tensor, beam: int, max_gen_len: int,
                end_ids: Sequence[int], pad_id: int = 0, early_stop: bool = True, no_repeat_ngram_size: int = 0,
                length_penalty: float = 1.0, position_ids: Optional[Tensor] = None, segment_ids: Optional[Tensor] = None) -> Tensor:
    bsz = input_ids.shape[0]
    max_len = max_gen_len + input_ids.shape[-1]
    ids = input_ids.repeat_interleave(beam, dim=0)
    mask = attention_mask.repeat_interleave(beam, dim=0)
    pos = None if position_ids is None else position_ids.repeat_interleave(beam, dim=0)
    seg = None if segment_ids is None else segment_ids.repeat_interleave(beam, dim=0)
    probs = torch.zeros(bsz, beam)
    probs[:, 1:] = -1e9                                                          # :238-239
    infos = [dict(done=False, worst=1e9, cands=[]) for _ in range(bsz)]         # :242
    pasts, step = None, 0
    while True:
        with torch.no_grad():
            logits, pasts = step_fn(ids[:, step:], mask, pasts, **_extra(pos, seg, step))
        last = logits[:, -1, :]
        if no_repeat_ngram_size > 1:
            last = no_repeat_ngram(ids, last, no_repeat_ngram_size)             # :253-256
        V = last.shape[-1]
        scores = (torch.log_softmax(last, dim=-1) + probs.view(-1, 1)).view(bsz, -1)   # :200-207
        cval, cflat = scores.topk(2 * beam, dim=1, largest=True, sorted=True)          # :217
        csrc, ctok = torch.div(cflat, V, rounding_mode="floor"), cflat % V
        nsrc = torch.zeros(bsz, beam, dtype=torch.long)
        ntok = torch.zeros(bsz, beam, dtype=ids.dtype)
        nval = torch.zeros(bsz, beam)
        for b in range(bsz):                                                      # :135-195
            info = infos[b]
            if info["done"]:
                ntok[b, :] = pad_id
                continue
            filled = 0
            for c in range(beam):
                if int(ctok[b, c]) in end_ids:
                    score = cval[b, c] / (ids.shape[-1] ** length_penalty)
                    info["cands"].append(dict(ids=ids[beam * b + int(csrc[b, c])], score=score))
                    if len(info["cands"]) > beam:
                        ranked = sorted((cd["score"], j) for j, cd in enumerate(info["cands"]))
                        del info["cands"][ranked[0][1]]
                        info["worst"] = ranked[1][0]
                    else:
                        info["worst"] = min(score, info["worst"])
                else:
                    nsrc[b, filled], ntok[b, filled], nval[b, filled] = csrc[b, c], ctok[b, c], cval[b, c]
                    filled += 1
                if filled >= beam:
                    break
            if len(info["cands"]) >= beam:
                if early_stop:
                    info["done"] = True
                elif info["worst"] > cval[b].max().item() / ((ids.shape[-1] + 1) ** length_penalty):
                    info["done"] = True
        rows = (nsrc + torch.arange(bsz)[:, None] * beam).view(-1)               # :262-265, :276-277
        ids = torch.cat([ids[rows], ntok.view(-1, 1)], dim=-1)
        mask = mask[rows]
        mask = torch.cat([mask, mask[:, -1:]], dim=-1)
        if pos is not None:                                                       # :268-269
            pos = pos[rows]
            pos = torch.cat([pos, pos[:, -1:] + 1], dim=-1)
        if seg is not None:                                                       # :270-271
            seg = seg[rows]
            seg = torch.cat([seg, seg[:, -1:]], dim=-1)
        pasts = [tuple(s.index_select(0, rows) for s in layer) for layer in pasts]   # :278-282
        probs = nval
        step = ids.shape[1] - 1
        if step > max_len:                                                        # :286-288
            break
    return ids.view(bsz, beam, -1)


def greedy_ngram(step_fn: Callable, input_ids: Tensor, attention_mask: Tensor, max_gen_len: int, n: int,
                 position_ids: Optional[Tensor] = None, segment_ids: Optional[Tensor] = None) -> Tensor:
    """generation_util.py:57-119 with do_sample=False, end_ids=None and the n-gram ban (:71-74)."""
    ids, mask = input_ids.clone(), attention_mask.clone()
    pos, seg = position_ids, segment_ids
    max_len = max_gen_len + ids.shape[-1]
    pasts, step = None, 0
    while True:
        with torch.no_grad():
            logits, pasts = step_fn(ids[:, step:], mask, pasts, **_extra(pos, seg, step))
        last = logits[:, -1, :]
        if n > 1:
            last = no_repeat_ngram(ids, last, n)
        ids = torch.cat([ids, torch.argmax(last, dim=-1)[:, None]], dim=-1)
        if pos is not None:
            pos = torch.cat([pos, (pos.max(dim=-1).values + 1).view(-1, 1)], dim=-1)     # :98
        if seg is not None:
            seg = torch.cat([seg, seg[:, -1:]], dim=-1)                                  # :99
        mask = torch.cat([mask, mask[:, -1:]], dim=-1)
        step = ids.shape[1] - 1
        if step > max_len:
            break
    return ids.view(ids.shape[0], 1, -1)
