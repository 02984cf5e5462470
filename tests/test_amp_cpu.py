"""CPU (no GPU): loss scaling — SURVEY §8(f)2, the ``use_torch_amp`` branch of ft_bloom_DDP.py:107-128 — driven through the
torch-CPU emulation of the kernel contracts.  A power-of-two scale is exact in fp32, so the scaled loop must land on the SAME
golden trajectory (generated from the reference model + torch.optim.AdamW) as the unscaled one."""

import pytest
import torch

import cpu_kernel_emulation as emu
from test_host_logic_cpu import TINY, T, build, close


def _setup(monkeypatch):
    emu.install(monkeypatch)
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    from cleantransformer_amd.optimizer import AdamW
    m = build(V, H, L, nh)
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    batch = {"input_ids": T(TINY["ids"]), "attention_mask": T(TINY["mask"]), "labels": T(TINY["ids"]).clone()}
    return m, opt, batch


def test_scaled_loop_lands_on_the_reference_trajectory(monkeypatch):
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.examples.ft_bloom import train_step_amp
    m, opt, batch = _setup(monkeypatch)
    scaler = GradScaler()
    for t in range(4):
        opt.zero_grad()
        loss = train_step_amp(m, batch, opt, scaler)
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 1e-6 * TINY["traj"][t, 0], (t, float(loss))
        if t == 0:                                      # after step() the gradients are the TRUE (unscaled) ones, as with torch
            for n, p in m.named_parameters():
                close(p.grad, TINY["g0_" + n], 1e-4, 1e-7)
    for n, p in m.named_parameters():
        close(p, TINY["p4_" + n], 1e-5, 1e-7)
    assert scaler.get_scale() == 65536.0 and scaler.state_dict()["_growth_tracker"] == 4


def test_reference_branch_has_no_zero_grad(monkeypatch):
    """ft_bloom_DDP.py:121-127 never clears gradients on the amp branch: after step t the stored gradient is
    g_t + (previous stored gradient) / scale."""
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.examples.ft_bloom import train_step_amp
    m, opt, batch = _setup(monkeypatch)
    scaler = GradScaler(init_scale=4.0)
    train_step_amp(m, batch, opt, scaler)
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    w1 = {n: p.detach().clone() for n, p in m.named_parameters()}
    train_step_amp(m, batch, opt, scaler)
    m2, opt2, _ = _setup(monkeypatch)                 # fresh model moved to the same weights: its clean gradient is g_2
    with torch.no_grad():
        for n, p in m2.named_parameters():
            p.copy_(w1[n])
    (loss, _, _), _ = m2(**batch)
    loss.backward()
    for n, p in m.named_parameters():
        close(p.grad, dict(m2.named_parameters())[n].grad + g1[n] / 4.0, 1e-5, 1e-9)


def test_overflow_skips_the_step_and_backs_off(monkeypatch):
    from cleantransformer_amd.amp import GradScaler
    m, opt, batch = _setup(monkeypatch)
    scaler = GradScaler(init_scale=1024.0, growth_interval=2)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    (loss, _, _), _ = m(**batch)
    scaler.scale(loss).backward()
    m.bloom.ln_f.bias.grad[3] = float("inf")
    assert scaler.step(opt) is None
    scaler.update()
    assert scaler.get_scale() == 512.0 and opt.steps[0] == 1             # Adam's step count did not advance
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    for t in range(2):                                                     # two clean steps -> growth
        opt.zero_grad()
        (loss, _, _), _ = m(**batch)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)                                               # explicit unscale (e.g. before clipping), then step
        with pytest.raises(RuntimeError):
            scaler.unscale_(opt)
        scaler.step(opt)
        scaler.update()
    assert scaler.get_scale() == 1024.0 and opt.steps[0] == 3
    sd = scaler.state_dict()
    s2 = GradScaler()
    s2.load_state_dict(sd)
    assert s2.get_scale() == 1024.0 and s2.state_dict() == sd
    off = GradScaler(enabled=False)
    assert off.scale(loss) is loss and off.state_dict() == {} and off.get_scale() == 1.0


def test_torch_gradscaler_drives_the_fused_optimizer(monkeypatch):
    """A caller that keeps ``torch.cuda.amp.GradScaler()`` (ft_bloom_DDP.py:109) only needs ``param_groups`` and ``step()``."""
    m, opt, batch = _setup(monkeypatch)
    scaler = torch.amp.GradScaler("cpu", init_scale=256.0)
    for t in range(2):
        opt.zero_grad()
        (loss, _, _), _ = m(**batch)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 1e-6 * TINY["traj"][t, 0]
    assert opt.steps[0] == 3


def test_autocast_selects_the_compute_dtype_of_the_forwards_inside_it():
    """ft_bloom_DDP.py:122 writes `with autocast():` (fp16 on a GPU).  Round 5: the context selects the compute dtype of the model forwards run
    inside it (ops.effective_compute_dtype) — fp16 (the reference's), bf16 or fp32; nested contexts restore; a disabled context (or dtype=float32) means no override inside; the override is per thread;
    the default form keeps the model's own dtype and says so once; other dtypes are refused."""
    import warnings
    from cleantransformer_amd import amp, ops
    assert ops.effective_compute_dtype(torch.bfloat16) is torch.bfloat16
    with amp.autocast(dtype=torch.float16):
        assert ops.effective_compute_dtype(torch.bfloat16) is torch.float16
        with amp.autocast(dtype=torch.bfloat16):
            assert ops.effective_compute_dtype(torch.float32) is torch.bfloat16
        assert ops.effective_compute_dtype(torch.float32) is torch.float16
        with amp.autocast(False, dtype=torch.float32):        # torch.cuda.amp.autocast(enabled, ...): the first positional argument is `enabled`
            assert ops.effective_compute_dtype(torch.bfloat16) is torch.bfloat16      # a disabled region nested in an enabled one: no override inside (torch semantics, round-5 advisor)
        assert ops.effective_compute_dtype(torch.float32) is torch.float16            # ... and the outer context is back afterwards
        with amp.autocast(dtype=torch.float32):                                       # torch disables autocast for float32: the model's own dtype, not a forced fp32 forward
            assert ops.effective_compute_dtype(torch.bfloat16) is torch.bfloat16
    assert ops.effective_compute_dtype(torch.float32) is torch.float32
    # the override is per thread
    import threading
    seen = []
    with amp.autocast(dtype=torch.float16):
        th = threading.Thread(target=lambda: seen.append(ops.effective_compute_dtype(torch.bfloat16)))
        th.start(); th.join()
    assert seen == [torch.bfloat16]
    with pytest.raises(NotImplementedError):
        with amp.autocast("cuda", torch.float64):
            pass
    try:
        with amp.autocast(True, dtype=torch.float16):
            raise KeyError("boom")
    except KeyError:
        pass
    assert ops._AUTOCAST_DTYPE is None                        # restored when the body raises
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with amp.autocast(dtype=torch.bfloat16):
            pass
        with amp.autocast(dtype=torch.float16, enabled=False):
            assert ops.effective_compute_dtype(torch.bfloat16) is torch.bfloat16
    amp._warned_default = False
    with pytest.warns(UserWarning, match="fp16"):
        with amp.autocast():
            assert ops.effective_compute_dtype(torch.bfloat16) is torch.bfloat16
