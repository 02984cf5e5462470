"""Logits processors of the decode path — the reference's ``CleanTransformer/generation/logits_processor.py`` surface
(same class names, constructor arguments and call signature) over this package's kernels.

* :class:`NoRepeatNGramLogitsProcessor` (logits_processor.py:11-32): token ids are tiny host-side state; the ban list is computed
  on the host from one ``tolist()`` and applied with one indexed write.
* :class:`TemperatureLogitsWrapper` (:35-41) and :class:`TopKLogitsWrapper` (:44-56): ``ctmi_scores_filter`` (a true division /
  a per-row threshold fill) with the k-th largest value from ``ctmi_group_topk``.
* :class:`TopPLogitsWrapper` (:59-79): the nucleus filter needs a full-vocabulary sort; it only runs on the random-sampling
  path (whose token ids depend on the device RNG and are not parity-checkable), so it uses torch's device sort / cumsum.

Wrappers take and return fp32 scores (bf16 logits are widened exactly before they are filtered).
"""
from __future__ import annotations

import torch

from .. import ops


def _f32(scores):
    return scores if scores.dtype == torch.float32 else scores.float()


class NoRepeatNGramLogitsProcessor():
    def __init__(self, ngram_size):
        self.ngram_size = ngram_size

    @staticmethod
    def banned_tokens(tokens, n):
        """tokens that would complete an n-gram already present in `tokens` (logits_processor.py:18-30)."""
        if n <= 1 or len(tokens) + 1 < n:       # n == 1: the reference looks up the whole history as the prefix -> never found
            return []
        prefix = tokens[len(tokens) - (n - 1):]
        return [tokens[j + n - 1] for j in range(len(tokens) - n + 1) if tokens[j:j + n - 1] == prefix]

    def __call__(self, input_ids, scores):
        rows, cols = [], []
        for i, toks in enumerate(input_ids.tolist()):
            for t in self.banned_tokens(toks, self.ngram_size):
                rows.append(i)
                cols.append(t)
        if rows:
            dev = scores.device
            scores[torch.tensor(rows, device=dev), torch.tensor(cols, device=dev)] = -float("inf")   # in place, as the reference
        return scores


class TemperatureLogitsWrapper():
    def __init__(self, temperature):
        self.temperature = max(temperature, 1e-2)

    def __call__(self, input_ids, scores, *args, **kwargs):
        return ops.scores_filter(_f32(scores), divisor=self.temperature)


class TopKLogitsWrapper():
    def __init__(self, top_k, filter_value=-float('Inf'), min_tokens_to_keep=1):
        self.top_k = int(max(top_k, min_tokens_to_keep, 1))
        self.filter_value = filter_value

    def __call__(self, input_ids, scores, *args, **kwargs):
        scores = _f32(scores)
        top_k = min(self.top_k, scores.size(-1))
        kth, _ = ops.group_topk(scores, 1, top_k)                       # [rows, top_k] descending
        return ops.scores_filter(scores, thr=kth[:, top_k - 1], fill=self.filter_value)


class TopPLogitsWrapper():
    def __init__(self, top_p, filter_value=-float('Inf'), min_tokens_to_keep=1):
        self.top_p = max(min(top_p, 1.0), 0)
        self.filter_value = filter_value
        self.min_tokens_to_keep = max(1, min_tokens_to_keep)

    def __call__(self, input_ids, scores, *args, **kwargs):
        scores = _f32(scores)
        ordered, order = torch.sort(scores, descending=False)
        tail_mass = ordered.softmax(dim=-1).cumsum(dim=-1)
        drop = tail_mass <= (1 - self.top_p)                             # the low-probability tail holding <= 1-p of the mass
        drop[..., -self.min_tokens_to_keep:] = False
        return scores.masked_fill(torch.zeros_like(drop).scatter(1, order, drop), self.filter_value)
