"""Loss scaling for the mixed-precision SFT loop — ``GradScaler`` with the surface of ``torch.cuda.amp.GradScaler`` as the
reference's DDP script drives it (examples/ft_bloom_DDP.py:107-128):

    scaler = GradScaler()
    scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()

MI355X-first: the record {scale, growth tracker, found_inf} lives in ONE device tensor; ``scale()`` multiplies by it on the
device, ``unscale_`` is a multi-tensor in-place kernel (``ctmi_amp_unscale``: one launch per <= 24 gradients, writes
``found_inf``), ``update()`` a one-thread kernel (``ctmi_amp_update``) — the scale itself is never read on the host.  The only
host round-trip is the 4-byte ``found_inf`` read in ``step()``, the same one torch's scaler makes before deciding to call
``optimizer.step()`` (so that a skipped step does not advance Adam's bias-correction count).

With the bf16 compute policy of this package (fp32 master weights and gradients, bf16 operands) scaling is not needed for range;
with a power-of-two scale in fp32/bf16 it is exact (scaled gradients are bit-identical after unscaling unless something overflowed).
In fp16 (``autocast(dtype=torch.float16)`` or ``config.compute_dtype = "fp16"``: the reference's autocast precision, round 5) it does its
real job: activation gradients travel as IEEE half, overflow to inf is what ``found_inf`` catches, a skipped step halves the scale.
"""
from __future__ import annotations

import contextlib

import torch

from . import ops


_warned_default = False


@contextlib.contextmanager
def autocast(device_type=None, dtype=None, enabled=True, cache_enabled=None):
    """``torch.cuda.amp.autocast()`` / ``torch.autocast(...)`` for the reference's loop (ft_bloom_DDP.py:122).  The fused kernels do not dispatch
    through torch's autocast; the context selects the COMPUTE DTYPE of the model forwards run inside it (``ops.effective_compute_dtype``):
      * ``autocast(dtype=torch.float16)`` — the reference's published DDP launch (scripts/ft_bloom_DDP.sh:11 ``--use_torch_amp``; torch's default
        autocast dtype on a GPU): activations and operand copies of the weights in IEEE half, fp32 accumulation, softmax / LayerNorm / loss
        statistics in fp32 (a superset of the fp16 -> fp32 score upcast at modeling_bloom.py:106-107), fp32 master weights and gradients, on fp16
        twins of the bf16 kernel families; a ``GradScaler`` is needed here, as in the reference;
      * ``autocast(dtype=torch.bfloat16)`` — the measured path (same as ``config.compute_dtype = "bf16"``);
      * ``autocast(enabled=False)`` / ``autocast(dtype=torch.float32)`` — no override inside (also when nested in an enabled context): the model's own dtype;
      * ``autocast()`` with no dtype (what ft_bloom_DDP.py literally writes) keeps the model's own ``config.compute_dtype`` and says so once: this
        package's default mixed precision is bf16, not fp16 (no loss scaling needed for range) — pass
        ``dtype=torch.float16`` (``examples.ft_bloom_DDP.train(amp_dtype=torch.float16)``) to reproduce the reference's precision."""
    global _warned_default
    if isinstance(device_type, bool):        # torch.cuda.amp.autocast's first positional argument is `enabled` (torch.autocast's is device_type):
        enabled, device_type = device_type, None     # autocast(False) must mean "off", not device_type=False (round-3 advisor)
    old = ops.get_autocast_dtype()
    if not enabled:
        # torch semantics: a disabled region nested in an enabled one runs WITHOUT autocast — the forwards inside compute in the model's own
        # config.compute_dtype again (round-5 advisor; rounds 3-5 left the outer override in force)
        ops.set_autocast_dtype(None)
    else:
        if dtype not in (None, torch.float16, torch.bfloat16, torch.float32):
            raise NotImplementedError(f"autocast(dtype={dtype}) is not supported: compute dtypes are fp16, bf16 and fp32")
        if dtype is None and not _warned_default:
            _warned_default = True
            import warnings
            warnings.warn("cleantransformer_amd.amp.autocast(): no dtype given — the forward runs in the model's config.compute_dtype (bf16 or fp32; "
                          "fp32 statistics and master weights).  torch's default autocast dtype on a GPU is fp16: pass dtype=torch.float16 for it",
                          stacklevel=3)
        if dtype == torch.float32:
            # torch.autocast(dtype=torch.float32) on a GPU warns and DISABLES autocast; here too: no override, not a forced fp32 forward of a
            # bf16 / fp16 model (which is what rounds 3-5 did, silently)
            ops.set_autocast_dtype(None)
        elif dtype is not None:
            ops.set_autocast_dtype(dtype)
    try:
        yield
    finally:
        ops.set_autocast_dtype(old)


class GradScaler():
    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        if enabled:
            assert growth_factor > 1.0 and backoff_factor < 1.0
        self._enabled = enabled
        self._init_scale, self._growth, self._backoff, self._interval = float(init_scale), growth_factor, backoff_factor, int(growth_interval)
        self._state = None                                   # device float[3]: scale, growth tracker, found_inf (lazily placed)
        self._unscaled = set()                                # ids of optimizers already unscaled this step
        if enabled and torch.cuda.is_available():
            # register the device scale NOW, not at the first scale(): the first forward of a new scaler already folds it into dlogits —
            # in fp16 that is a matter of correctness, not only of a saved pass (an unscaled half dlogits underflows; round-5 advisor)
            self._ensure(torch.device("cuda", torch.cuda.current_device()))

    # ------------------------------------------------------------------------------------------------ state
    def _ensure(self, device):
        device = torch.device(device)
        if self._state is None:
            self._place(torch.tensor([self._init_scale, 0.0, 0.0], dtype=torch.float32, device=device))
        elif device.type == "cuda" and self._state.is_cuda and device.index is not None and self._state.device != device:
            # constructed (eagerly) on another GPU than the one the loss lives on: the state follows the loss and registers again
            self._place(self._state.to(device))
        return self._state

    def _place(self, state):
        self._state = state
        if state.is_cuda:
            # from the next forward on, the fused loss folds this scale into dlogits (ops.set_expected_loss_grad): the backward of
            # scaler.scale(loss) then finds its upstream gradient already applied and skips the rescale pass over [T,V]
            from . import ops
            ops.set_expected_loss_grad(scale=state[0:1], owner=self)

    def is_enabled(self):
        return self._enabled

    def get_scale(self):
        if not self._enabled:
            return 1.0
        return self._init_scale if self._state is None else float(self._state[0])

    def get_growth_factor(self):
        return self._growth

    def get_backoff_factor(self):
        return self._backoff

    def get_growth_interval(self):
        return self._interval

    def state_dict(self):
        if not self._enabled:
            return {}
        tracker = 0 if self._state is None else int(self._state[1])
        return {"scale": self.get_scale(), "growth_factor": self._growth, "backoff_factor": self._backoff,
                "growth_interval": self._interval, "_growth_tracker": tracker}

    def load_state_dict(self, sd):
        if not self._enabled:
            return
        if len(sd) == 0:
            raise RuntimeError("The source state dict is empty, possibly because it was saved from a disabled instance of GradScaler.")
        self._init_scale, self._growth, self._backoff = float(sd["scale"]), sd["growth_factor"], sd["backoff_factor"]
        self._interval = int(sd["growth_interval"])
        if self._state is not None:
            self._state.copy_(torch.tensor([self._init_scale, float(sd["_growth_tracker"]), 0.0]))
        else:
            self._pending_tracker = float(sd["_growth_tracker"])

    # ------------------------------------------------------------------------------------------------ the three calls
    def scale(self, outputs):
        if not self._enabled:
            return outputs
        if isinstance(outputs, (list, tuple)):
            return type(outputs)(self.scale(o) for o in outputs)
        st = self._ensure(outputs.device)
        if getattr(self, "_pending_tracker", None) is not None:
            st[1] = self._pending_tracker
            self._pending_tracker = None
        return outputs * st[0].to(outputs.dtype)

    @staticmethod
    def _grads(optimizer):
        if hasattr(optimizer, "param_groups"):
            ps = [p for g in optimizer.param_groups for p in g["params"]]
        else:
            ps = list(optimizer.params)
        return [p.grad for p in ps if p.grad is not None]

    def unscale_(self, optimizer):
        if not self._enabled:
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        grads = self._grads(optimizer)
        if grads:
            ops.amp_unscale(grads, self._ensure(grads[0].device))
        self._unscaled.add(id(optimizer))

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        if self._state is not None and float(self._state[2]) != 0.0:      # the one host read of the step
            return None                                                     # skipped: overflow in this step's gradients
        return optimizer.step(*args, **kwargs)

    def update(self, new_scale=None):
        if not self._enabled:
            return
        self._unscaled.clear()
        if self._state is None:
            return
        if new_scale is not None:
            self._state[0] = float(new_scale)
            self._state[2] = 0.0
            return
        ops.amp_update(self._state, self._growth, self._backoff, self._interval)
