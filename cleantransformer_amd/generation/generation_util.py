"""KV-cached decoding with the reference's GenerationMixin surface (CleanTransformer/generation/generation_util.py:13-290):
``generate(input_ids, attention_mask, position_ids, segment_ids, generation_configs, steamers)`` with the same config keys and
defaults, greedy / sampled search (``beam_size == 1``) and beam search.

What runs where
  * per-step scoring on the GPU through the C ABI: ``ctmi_argmax`` (greedy pick), ``ctmi_row_lse`` + ``ctmi_group_topk``
    (the ``log_softmax + beam score -> topk(2*beam)`` of generation_util.py:199-217 without materialising the [bsz, beam*V]
    score matrix), ``ctmi_scores_filter`` (temperature / top-k);
  * beam bookkeeping on the host, from ONE small device->host copy per step (2*beam candidates per batch element) — the
    reference's ``_update_beam_infos`` reads them one ``.item()`` at a time;
  * random sampling (``do_sample=True``) draws with ``torch.multinomial`` on the device: token ids then depend on the device RNG
    stream, so parity is defined on the filtered distributions (tests) and on the deterministic paths (bit-exact ids).

Reference behaviours kept because they decide the emitted ids:
  * loops exit on ``step > max_len`` -> max_gen_len + 2 new tokens (:115-116, :286-288);
  * beam search only inspects the first ``beam`` of the 2*beam candidates (:146); slots it cannot fill keep token 0, source beam 0
    and score 0 (:131-133); a finished batch element emits ``pad_id`` from the NEXT step on (:137-139);
  * finished hypotheses are collected (score = sum-logprob / length) only to decide ``is_done``; ``generate`` returns the live
    beams ``[bsz, beam, len]`` (:290);
  * ``end_ids=None`` with ``beam_size > 1`` is a TypeError in the reference (membership test on None, :144) and is one here.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .logits_processor import NoRepeatNGramLogitsProcessor, TemperatureLogitsWrapper, TopKLogitsWrapper, TopPLogitsWrapper


class _BeamBook:
    """Finished hypotheses of one batch element (the reference's ``generated_beam_infos[i]`` dict, :234)."""

    def __init__(self):
        self.done = False
        self.worst = np.float32(1e9)
        self.finished = []                                  # [(score fp32, ids tensor)]

    def add(self, ids_row, score, beam):
        self.finished.append((score, ids_row))
        if len(self.finished) > beam:                        # drop the lowest score (ties: the older one), :152-155
            order = sorted(range(len(self.finished)), key=lambda j: (self.finished[j][0], j))
            self.worst = self.finished[order[1]][0]
            del self.finished[order[0]]
        else:
            self.worst = min(score, self.worst)


class GenerationMixin():
    def generate(self, input_ids, attention_mask=None, position_ids=None, segment_ids=None, generation_configs={}, steamers=None):
        beam_size = generation_configs.get('beam_size', 1)
        max_gen_len = generation_configs.get('max_gen_len', 100)
        end_ids = generation_configs.get('end_ids', None)
        pad_id = generation_configs.get('pad_id', 0)
        no_repeat_ngram_size = generation_configs.get('no_repeat_ngram_size', 0)
        self.do_sample = generation_configs.get('do_sample', True)
        temperature = generation_configs.get('temperature', 1.0)
        top_k = generation_configs.get('top_k', 10)
        top_p = generation_configs.get('top_p', 0.8)
        early_stop = generation_configs.get('early_stop', True)

        if isinstance(end_ids, int):
            end_ids = [end_ids]
        end_ids_tensor = torch.tensor(list(end_ids)).to(input_ids.device) if end_ids is not None else None

        self.logits_processors = []
        if no_repeat_ngram_size > 1:
            self.logits_processors.append(NoRepeatNGramLogitsProcessor(no_repeat_ngram_size))
        self.logits_wrapper = []
        self.temperature = temperature
        if self.do_sample and temperature != 1.0:
            self.logits_wrapper.append(TemperatureLogitsWrapper(temperature))
        if self.do_sample and top_k > 0:
            self.logits_wrapper.append(TopKLogitsWrapper(top_k, min_tokens_to_keep=1))
        if self.do_sample and top_p < 1.0:
            self.logits_wrapper.append(TopPLogitsWrapper(top_p, min_tokens_to_keep=1))
        self.steamers = steamers

        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if beam_size == 1:
            return self._greedy_search(input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor,
                                       max_gen_len=max_gen_len, pad_id=pad_id)
        return self._beam_search(input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor, max_gen_len=max_gen_len,
                                 pad_id=pad_id, beam_size=beam_size, early_stop=early_stop)

    # ------------------------------------------------------------------------------------------------ shared pieces
    def _decode_step(self, input_ids, step, attention_mask, position_ids, segment_ids, k_v_pasts):
        """One KV-cached forward over the not-yet-consumed tokens; returns the last position's logits [rows, V] (a view)."""
        extra = {}
        if position_ids is not None:
            extra["position_ids"] = position_ids[:, step:]
        if segment_ids is not None:
            extra["segment_ids"] = segment_ids[:, step:]
        outputs, k_v_pasts = self(input_ids[:, step:], attention_mask=attention_mask, k_v_pasts=k_v_pasts, **extra)
        last = outputs[0][:, -1, :]
        for proc in getattr(self, "logits_processors", ()):
            last = proc(input_ids, last)
        return last, k_v_pasts

    def _stream(self, ids3d):
        finish = False
        if self.steamers is not None:
            self.steamers = self.steamers if isinstance(self.steamers, list) else [self.steamers, ]
            for steamer in self.steamers:
                if callable(steamer):
                    finish = steamer(ids3d) or finish
        return finish

    # ------------------------------------------------------------------------------------------------ beam_size == 1
    @torch.no_grad()
    def _greedy_search(self, input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor, max_gen_len, pad_id):
        """generation_util.py:57-119.  Keeps the reference's exit test (``step > max_len``), which emits max_gen_len + 2 tokens
        (SURVEY Q16), so that decoded ids are bit-identical."""
        bsz = input_ids.size(0)
        max_len = max_gen_len + input_ids.size(-1)
        k_v_pasts = [None for _ in range(self.config.n_layer)]
        step = 0
        unfinished = torch.ones(bsz, dtype=torch.long, device=input_ids.device)
        while True:
            last, k_v_pasts = self._decode_step(input_ids, step, attention_mask, position_ids, segment_ids, k_v_pasts)
            if self.do_sample:
                for wrap in self.logits_wrapper:
                    last = wrap(input_ids, last)
                step_output = torch.multinomial(torch.softmax(last.float(), dim=-1), num_samples=1).squeeze(1)   # :83-84
            else:
                step_output = ops.argmax_lastdim(last)                                   # :86
            step_output = step_output * unfinished + pad_id * (1 - unfinished)
            if end_ids_tensor is not None:
                unfinished = unfinished.mul(
                    step_output.tile(end_ids_tensor.shape[0], 1).ne(end_ids_tensor.unsqueeze(1)).prod(dim=0))
            input_ids = torch.concat([input_ids, step_output[:, None]], dim=-1)
            if position_ids is not None:
                position_ids = torch.concat([position_ids, (position_ids.max(dim=-1).values + 1).view(-1, 1)], dim=-1)
            if segment_ids is not None:
                segment_ids = torch.concat([segment_ids, segment_ids[:, -1:]], dim=-1)
            attention_mask = torch.concat([attention_mask, attention_mask[:, -1:]], dim=-1)
            if self._stream(input_ids.view(bsz, 1, -1)):
                break
            step = input_ids.shape[1] - 1
            if unfinished.max() == 0 or step > max_len:
                break
        return input_ids.view(bsz, 1, -1)

    # ------------------------------------------------------------------------------------------------ beam_size > 1
    def _beam_topk(self, x_ids, bsz, beam_size, last, beam_scores):
        """generation_util.py:199-224 -> (source beam, token, score) of the 2*beam best continuations per batch element."""
        vocab = last.shape[-1]
        stats = ops.row_lse(last)
        if not self.do_sample:
            val, flat = ops.group_topk(last, beam_size, 2 * beam_size, stats=stats, add=beam_scores.reshape(-1))
        else:
            scores = (last.float() - stats[:, 0:1]) - stats[:, 1:2]
            scores = (scores + beam_scores.reshape(-1, 1) * self.temperature).view(bsz, -1)
            for wrap in self.logits_wrapper:
                scores = wrap(x_ids, scores)
            flat = torch.multinomial(torch.softmax(scores, dim=-1), num_samples=2 * beam_size)
            val, order = torch.sort(torch.gather(scores, -1, flat), descending=True, dim=1)
            flat = torch.gather(flat, -1, order)
        return torch.div(flat, vocab, rounding_mode="floor"), flat % vocab, val

    def _update_beam_infos(self, beam, books, input_ids, token_indices, next_tokens, probs, end_ids, pad_token_id,
                           length_penalty=1.0, early_stop=True):
        """generation_util.py:121-197 on host copies of the 2*beam candidates; returns the host lists of the next beams."""
        bsz = len(books)
        cur_len = input_ids.shape[-1]
        src_l, tok_l, val_l = token_indices.tolist(), next_tokens.tolist(), probs.float().cpu().numpy()
        new_src = [[0] * beam for _ in range(bsz)]
        new_tok = [[0] * beam for _ in range(bsz)]
        new_val = np.zeros((bsz, beam), dtype=np.float32)
        norm = np.float32(cur_len ** length_penalty)
        for b, book in enumerate(books):
            if book.done:
                new_tok[b] = [pad_token_id] * beam
                continue
            filled = 0
            for c in range(beam):                                   # only the first `beam` of the 2*beam candidates (:141)
                if tok_l[b][c] in end_ids:
                    book.add(input_ids[beam * b + src_l[b][c]], val_l[b, c] / norm, beam)
                else:
                    new_src[b][filled], new_tok[b][filled], new_val[b, filled] = src_l[b][c], tok_l[b][c], val_l[b, c]
                    filled += 1
                if filled >= beam:
                    break
            if len(book.finished) >= beam:
                if early_stop:
                    book.done = True
                else:                                               # no future hypothesis can beat the worst kept one (:189-193)
                    best_next = np.float32(float(val_l[b].max()) / ((cur_len + 1) ** length_penalty))
                    book.done = bool(book.worst > best_next)
        return new_src, new_tok, new_val

    @torch.no_grad()
    def _beam_search(self, input_ids, attention_mask, position_ids, segment_ids, end_ids_tensor, max_gen_len, pad_id, beam_size,
                     early_stop):
        if end_ids_tensor is None:
            raise TypeError("beam search needs end_ids (the reference tests membership in end_ids, generation_util.py:144)")
        end_ids = set(end_ids_tensor.tolist())
        dev = input_ids.device
        bsz = input_ids.size(0)
        max_len = max_gen_len + input_ids.size(-1)
        k_v_pasts = [None for _ in range(self.config.n_layer)]
        step = 0
        rep = lambda t: None if t is None else t.repeat_interleave(beam_size, dim=0)     # noqa: E731   (:232-235)
        input_ids, position_ids, attention_mask, segment_ids = rep(input_ids), rep(position_ids), rep(attention_mask), rep(segment_ids)
        beam_scores = torch.zeros((bsz, beam_size), device=dev)
        beam_scores[:, 1:] = -1e9                       # all beams start identical: the first expansion uses beam 0 only (:238-239)
        books = [_BeamBook() for _ in range(bsz)]
        base = (torch.arange(bsz, device=dev) * beam_size)[:, None]

        while True:
            last, k_v_pasts = self._decode_step(input_ids, step, attention_mask, position_ids, segment_ids, k_v_pasts)
            src, tok, val = self._beam_topk(input_ids, bsz, beam_size, last, beam_scores)
            new_src, new_tok, new_val = self._update_beam_infos(beam_size, books, input_ids, src, tok, val, end_ids, pad_id,
                                                                early_stop=early_stop)
            rows = (torch.tensor(new_src, device=dev, dtype=torch.long) + base).view(-1)          # flat source rows
            step_output = torch.tensor(new_tok, device=dev, dtype=input_ids.dtype).view(-1)
            beam_scores = torch.from_numpy(new_val).to(dev)

            take = lambda t: None if t is None else t.index_select(0, rows)              # noqa: E731
            input_ids = torch.concat([take(input_ids), step_output[:, None]], dim=-1)
            if position_ids is not None:
                position_ids = take(position_ids)
                position_ids = torch.concat([position_ids, position_ids[:, -1:] + 1], dim=-1)
            attention_mask = take(attention_mask)
            attention_mask = torch.concat([attention_mask, attention_mask[:, -1:]], dim=-1)
            if segment_ids is not None:
                segment_ids = take(segment_ids)
                segment_ids = torch.concat([segment_ids, segment_ids[:, -1:]], dim=-1)
            k_v_pasts = [tuple(s.index_select(0, rows) for s in layer) for layer in k_v_pasts]      # caches follow their beams

            if self._stream(input_ids.view(bsz, beam_size, -1)):
                break
            step = input_ids.shape[1] - 1
            if step > max_len:
                break
        return input_ids.view(bsz, beam_size, -1)
