"""Greedy (argmax) KV-cached decoding with the reference's GenerationMixin surface
(CleanTransformer/generation/generation_util.py:13-119).  Only the deterministic path the hot-path contract names
("bit-exact argmax decode") is built: ``beam_size == 1`` and ``do_sample == False``.  Sampling / beam search /
logits processors are listed as "next" in SURVEY.md §8(f) and raise NotImplementedError.
"""
from __future__ import annotations

import torch

from .. import ops


class GenerationMixin():
    def generate(self, input_ids, attention_mask=None, position_ids=None, segment_ids=None, generation_configs={}, steamers=None):
        beam_size = generation_configs.get('beam_size', 1)
        max_gen_len = generation_configs.get('max_gen_len', 100)
        end_ids = generation_configs.get('end_ids', None)
        pad_id = generation_configs.get('pad_id', 0)
        no_repeat_ngram_size = generation_configs.get('no_repeat_ngram_size', 0)
        self.do_sample = generation_configs.get('do_sample', True)
        if beam_size != 1 or self.do_sample or no_repeat_ngram_size > 1:
            raise NotImplementedError("only greedy decoding (beam_size=1, do_sample=False, no n-gram penalty) is built; "
                                      "sampling / beam search are SURVEY §8(f) 'next'")
        if isinstance(end_ids, int):
            end_ids = [end_ids]
        end_ids_tensor = torch.tensor(list(end_ids)).to(input_ids.device) if end_ids is not None else None
        self.steamers = steamers
        return self._greedy_search(input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor,
                                   max_gen_len=max_gen_len, pad_id=pad_id)

    @torch.no_grad()
    def _greedy_search(self, input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor, max_gen_len, pad_id):
        """generation_util.py:57-119 with do_sample=False.  Keeps the reference's exit test (``step > max_len``), which
        emits max_gen_len + 2 tokens, so that decoded ids are bit-identical."""
        bsz = input_ids.size(0)
        max_len = max_gen_len + input_ids.size(-1)
        k_v_pasts = [None for _ in range(self.config.n_layer)]
        step = 0
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        unfinished = torch.ones(bsz, dtype=torch.long, device=input_ids.device)
        while True:
            outputs, k_v_pasts = self(input_ids[:, step:], attention_mask=attention_mask, k_v_pasts=k_v_pasts)
            logits = outputs[0]
            step_output = ops.argmax_lastdim(logits[:, -1, :])                       # generation_util.py:86
            step_output = step_output * unfinished + pad_id * (1 - unfinished)
            if end_ids_tensor is not None:
                unfinished = unfinished.mul(
                    step_output.tile(end_ids_tensor.shape[0], 1).ne(end_ids_tensor.unsqueeze(1)).prod(dim=0))
            input_ids = torch.concat([input_ids, step_output[:, None]], dim=-1)
            attention_mask = torch.concat([attention_mask, attention_mask[:, -1:]], dim=-1)
            finish = False
            if self.steamers is not None:
                self.steamers = self.steamers if isinstance(self.steamers, list) else [self.steamers, ]
                for steamer in self.steamers:
                    if callable(steamer):
                        finish = steamer(input_ids.view(bsz, 1, -1)) or finish
            if finish:
                break
            step = input_ids.shape[1] - 1
            if unfinished.max() == 0 or step > max_len:
                break
        return input_ids.view(bsz, 1, -1)
