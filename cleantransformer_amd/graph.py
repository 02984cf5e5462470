"""The whole SFT step as ONE hipGraph launch.

The step of ft_bloom.py:79-90 — ``model(**batch)`` -> ``optimizer.zero_grad()`` -> ``loss.backward()`` -> ``optimizer.step()`` — is ~460 kernel
launches on one stream, each depending on the one before.  Between two dependent launches of a stream the GPU idles: measured on an MI355X
(profiles/r06_launch_floor.txt) a chain of LayerNorm launches pays ~4.5 us per launch beside its work, ~2.5 us when the same chain is replayed
from a hipGraph (a launch that does nothing: 4.5 vs 1.6 us) — about a millisecond of a 36 ms step.  ``GraphedStep`` captures the reference
loop's step once, with ``torch.cuda.graph`` (stream capture: every kernel of this package is a plain launch on the current stream, nothing in
the hot path synchronises with the host), and replays it for every later batch of the same shape.

    step = GraphedStep(model, optimizer)
    for batch in loader:
        loss = step(batch["input_ids"], batch["attention_mask"], batch["labels"])      # == outputs[0] of the reference loop

What is inside the graph is exactly what the eager loop launches, in the same order — the results agree to the last bit except where the eager step
itself is not bit-reproducible (the tied table's embedding gradient is accumulated with fp32 atomics).  What cannot be inside a graph is anything
that changes from step to step without being data: the optimizer's bias corrections and learning rate.  ``AdamW.prepare_graph_step()`` writes
them into a 48-byte device record before each replay (optimizer.py), so schedulers and ``grad_scale`` keep working.

The first ``warmup`` calls run EAGERLY on their own batches (they are real training steps: lazily created optimizer state, compute-dtype weight
copies and the grouped weight-gradient work lists all appear there, outside any capture); the next call captures and replays.  A batch of another
shape, or ``model.eval()``, falls back to an eager step (and re-captures if the new shape persists).  Single-GPU: under DistributedDataParallel the
step stays eager — its tied-gradient row exchange agrees on a capacity through pinned host memory inside backward.
"""
from __future__ import annotations

from typing import Optional

import torch


class GraphedStep:
    def __init__(self, model: torch.nn.Module, optimizer, warmup: int = 2, enabled: bool = True):
        if not hasattr(optimizer, "enable_graph_mode"):
            raise TypeError("GraphedStep needs this package's fused optimizer (cleantransformer_amd.optimizer.AdamW): its per-step numbers must "
                            "live in device memory for a replay to see them")
        self.model, self.optimizer = model, optimizer
        self.warmup, self.enabled = int(warmup), bool(enabled)
        self.calls = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._shape = None
        self._static = None
        self._loss: Optional[torch.Tensor] = None
        self.replays = 0
        self.fallback_reason: Optional[str] = None

    # ------------------------------------------------------------------------------------------------ the reference loop's step
    def _eager(self, input_ids, attention_mask, labels):
        outputs, _ = self.model(input_ids=input_ids, attention_mask=attention_mask, labels=labels)
        loss = outputs[0]
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _capture(self, input_ids, attention_mask, labels) -> None:
        dev = input_ids.device
        self._static = tuple(torch.empty_like(t) for t in (input_ids, attention_mask, labels))
        for s, t in zip(self._static, (input_ids, attention_mask, labels)):
            s.copy_(t)
        self.optimizer.enable_graph_mode(dev)
        g = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize(dev)                               # everything the eager warm-up left on side streams is complete: nothing to wait for inside
            with torch.cuda.graph(g):
                outputs, _ = self.model(input_ids=self._static[0], attention_mask=self._static[1], labels=self._static[2])
                loss = outputs[0]
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
                self._loss = loss.detach()
        except Exception as e:                                       # noqa: BLE001
            self.optimizer.disable_graph_mode()
            self.graph, self._static, self._loss = None, None, None
            self.enabled = False
            self.fallback_reason = f"capture failed: {type(e).__name__}: {e}"[:300]
            torch.cuda.synchronize(dev)
            raise
        self.graph = g
        self._shape = (tuple(input_ids.shape), input_ids.dtype, attention_mask.dtype, labels.dtype)
        # The replays address memory by pointer.  Everything the captured launches touch that lives OUTSIDE the graph's own pool is pinned here, so a
        # cache that later replaces its buffer (a larger backward scratch for another model, a weight copy in another compute dtype) frees nothing
        # the graph still writes to: the process-wide scratch buffers, the compute-dtype weight copies, the optimizer state.
        from . import ops
        keep = list(ops._BLOCK_WS.values()) + list(ops._SPLITK_WS.values())
        for p in self.model.parameters():
            keep += [getattr(p, a) for a in ("_ct_shadow", "_ct_shadow_pad", "_ct_wt") if torch.is_tensor(getattr(p, a, None))]
        keep += [t for t in list(self.optimizer.momentum_buffer) + list(self.optimizer.rmsp_buffer) if torch.is_tensor(t)]
        self._keep = keep
        self._opt_epoch = getattr(self.optimizer, "_state_epoch", 0)

    def __call__(self, input_ids, attention_mask, labels):
        self.calls += 1
        shape = (tuple(input_ids.shape), input_ids.dtype, attention_mask.dtype, labels.dtype)
        usable = self.enabled and self.model.training and input_ids.is_cuda and torch.is_grad_enabled()
        if not usable or self.calls <= self.warmup:
            if self.graph is not None:
                self._drop()
            return self._eager(input_ids, attention_mask, labels)
        if self.graph is not None and getattr(self.optimizer, "_state_epoch", 0) != self._opt_epoch:
            self._drop()                                              # optimizer.load_state_dict() replaced the moment buffers the graph writes
            return self._eager(input_ids, attention_mask, labels)
        if self.graph is not None and shape != self._shape:
            self._drop()                                              # another shape: eager now, capture again when it persists
            return self._eager(input_ids, attention_mask, labels)
        if self.graph is None:
            try:
                self._capture(input_ids, attention_mask, labels)      # records launches, executes nothing
            except Exception:                                         # noqa: BLE001  (fallback_reason holds it; the step itself must still happen)
                return self._eager(input_ids, attention_mask, labels)
        else:
            for s, t in zip(self._static, (input_ids, attention_mask, labels)):
                s.copy_(t, non_blocking=True)
        self.optimizer.prepare_graph_step()
        self.graph.replay()
        self.replays += 1
        return self._loss.clone()

    def reset(self) -> None:
        """Forget the captured graph (the next calls warm up and capture again).  Needed after anything that REPLACES tensors the step uses instead
        of writing into them — e.g. swapping a parameter object; model.load_state_dict() and optimizer.load_state_dict() are handled."""
        self._drop()

    def _drop(self) -> None:
        self.optimizer.disable_graph_mode()
        self.graph, self._static, self._loss, self._shape, self._keep = None, None, None, None, None
        self.calls = 0                                                # the next shape warms up again (its gradients are new allocations)
