"""CPU (no GPU): probability-target branch of CrossEntropyLoss (loss.py:43-46).  The oracle's restatement
(oracle.bloom_ref.cross_entropy_repo) is pinned to tests/golden/soft_ce.npz (losses and input gradients produced by the
reference's own class, incl. the inputs of its printed self-check, 3.14231014); the product's autograd node is driven through the
kernel-contract emulation against the same vectors."""
import os

import numpy as np
import pytest
import torch

import cpu_kernel_emulation as emu
from oracle import bloom_ref as R

SC = np.load(os.path.join(os.path.dirname(__file__), "golden", "soft_ce.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_soft_ce_matches_reference():
    x = T(SC["x"])
    for name in ("norm", "raw"):
        t = T(SC[f"t_{name}"])
        for red in ("mean", "sum"):
            xi = x.clone().requires_grad_(True)
            loss = R.cross_entropy_repo(xi, t, red)
            loss.backward()
            assert abs(float(loss) - float(SC[f"loss_{name}_{red}"])) <= 1e-6 * abs(float(loss))
            assert torch.allclose(xi.grad, T(SC[f"dx_{name}_{red}"]), rtol=1e-5, atol=1e-7)
    known = R.cross_entropy_repo(T(SC["known_pred"]), T(SC["known_t"]), "mean")
    assert abs(float(known) - 3.14231014) < 1e-6 and abs(float(known) - float(SC["known_loss"])) < 1e-6


def test_soft_ce_host_logic_matches_reference(monkeypatch):
    emu.install(monkeypatch)
    from CleanTransformer.loss import CrossEntropyLoss
    x = T(SC["x"])
    for name in ("norm", "raw"):
        t = T(SC[f"t_{name}"])
        for red in ("mean", "sum"):
            xi = x.clone().requires_grad_(True)
            loss = CrossEntropyLoss(red)(xi, t)
            (loss * 3.0).backward()                                      # a non-unit upstream gradient
            assert abs(float(loss) - float(SC[f"loss_{name}_{red}"])) <= 2e-6 * abs(float(loss))
            assert torch.allclose(xi.grad, 3.0 * T(SC[f"dx_{name}_{red}"]), rtol=1e-5, atol=1e-6)
    assert abs(float(CrossEntropyLoss('mean')(T(SC["known_pred"]), T(SC["known_t"]))) - 3.14231014) < 1e-6
    with pytest.raises(ValueError):
        CrossEntropyLoss()(x, T(SC["t_norm"])[:, :5])
    with pytest.raises(ValueError):
        CrossEntropyLoss(ignore_index=-100)(x, T(SC["t_norm"]))
