"""Seeds for the counter-based dropout masks of the ctmi355 kernels.

The kernels keep element ``i`` of a dropout site iff ``keep_hash(i, seed) >= p * 2**32`` (csrc/common.h: the seed keys both rounds of
the hash, so two seeds give two functions of the counter, not two windows of one sequence); the backward regenerates the mask from the
same 32-bit seed.  A site draws its seed here once per forward call, from torch's default CPU generator — so
``torch.manual_seed(n)`` makes a run reproducible, exactly as it does for the reference's ``torch.nn.Dropout`` modules — and passes
it through SplitMix64, so consecutive draws give unrelated masks."""
from __future__ import annotations

import torch

_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def next_seed() -> int:
    """A fresh 32-bit seed (host side, no device synchronisation)."""
    raw = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
    return _splitmix64(raw) & 0xFFFFFFFF


def hash32(x: torch.Tensor) -> torch.Tensor:
    """ctmi_hash32 on an int64 tensor holding 32-bit values (host-side restatement, used by tests and the CPU emulation)."""
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x21f0aaad) & m
    x = x ^ (x >> 15)
    x = (x * 0x735a2d97) & m
    x = x ^ (x >> 15)
    return x


def keep_hash(counter: torch.Tensor, seed: int) -> torch.Tensor:
    """ctmi_dropout_keep_hash: hash32(hash32(counter ^ seed) + seed * 0x9E3779B1 + 0x7F4A7C15) on an int64 tensor of 32-bit counters."""
    m = 0xFFFFFFFF
    seed = int(seed) & m
    key = (seed * 0x9E3779B1 + 0x7F4A7C15) & m
    return hash32((hash32((counter & m) ^ seed) + key) & m)


def drop_threshold(p: float) -> int:
    """ctmi_dropout_threshold: p reaches the library as a C float, so it is rounded to fp32 first."""
    t = float(torch.tensor(float(p), dtype=torch.float32)) * 4294967296.0
    return 0 if t <= 0.0 else (4294967295 if t >= 4294967295.0 else int(t))
