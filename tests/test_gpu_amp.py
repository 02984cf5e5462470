"""GPU: loss scaling through the C ABI (ctmi_amp_unscale / ctmi_amp_update) — SURVEY §8(f)2, ft_bloom_DDP.py:107-128."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = np.load(os.path.join(HERE, "golden", "tiny_bloom.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_unscale_kernel_multi_tensor_and_found_inf():
    from cleantransformer_amd import ops
    g = torch.Generator().manual_seed(2)
    sizes = [1, 3, 4, 5, 1023, 4096, 70001] + [17] * 30                   # > 24 tensors: several launches; odd tails
    grads = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    base = torch.randn(sum(sizes) + 1, generator=g).to(DEV)
    grads.append(base[1:1 + 4097])                                          # a 4-byte-aligned (not 16-byte) view: scalar path
    ref = [x.clone() for x in grads]
    state = torch.tensor([8.0, 5.0, 0.0], device=DEV)
    ops.amp_unscale(grads, state)
    for x, r in zip(grads, ref):
        assert torch.equal(x, r / 8.0)
    assert state.tolist() == [8.0, 5.0, 0.0]
    for bad in (float("inf"), float("-inf"), float("nan")):
        state[2] = 0.0
        grads[6][70000] = bad                                               # last element of a large tensor
        ops.amp_unscale(grads, state)
        assert float(state[2]) == 1.0
        grads[6][70000] = 0.0
    # overflow produced BY the unscale (tiny scale) is caught as well
    state = torch.tensor([1e-30, 0.0, 0.0], device=DEV)
    big = [torch.full((100,), 1e20, device=DEV)]
    ops.amp_unscale(big, state)
    assert float(state[2]) == 1.0


def test_update_kernel_matches_torch_semantics():
    from cleantransformer_amd import ops
    st = torch.tensor([1024.0, 0.0, 1.0], device=DEV)
    ops.amp_update(st, 2.0, 0.5, 3)
    assert st.tolist() == [512.0, 0.0, 0.0]
    for expect in ([512.0, 1.0, 0.0], [512.0, 2.0, 0.0], [1024.0, 0.0, 0.0]):
        ops.amp_update(st, 2.0, 0.5, 3)
        assert st.tolist() == expect
    st = torch.tensor([3e38, 0.0, 0.0], device=DEV)                         # growth that would overflow keeps the scale
    ops.amp_update(st, 2.0, 0.5, 1)
    assert st.tolist()[0] == pytest.approx(3e38) and st.tolist()[1:] == [0.0, 0.0]


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_scaled_loop_equals_unscaled_loop(cd):
    """Power-of-two scaling is exact in fp32 and in bf16 operands: the scaled run reproduces the unscaled run bit for bit
    (fp32: and therefore the reference's golden trajectory)."""
    from test_gpu_bloom import build
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.examples.ft_bloom import train_step, train_step_amp
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    # bit-equality needs a bit-reproducible step: no token may repeat inside the batch, so that the embedding scatter-add
    # (fp32 atomics) has one contribution per table row
    ids = torch.randperm(V, generator=torch.Generator().manual_seed(5))[:B * S].view(B, S).to(DEV)
    batch = {"input_ids": ids, "attention_mask": T(TINY["mask"]).to(DEV), "labels": ids.clone()}
    ma, mb = build(V, H, L, nh, cd), build(V, H, L, nh, cd)
    oa = AdamW(ma.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    ob = AdamW(mb.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    scaler = GradScaler()
    for t in range(4):
        la = train_step(ma, batch, oa)
        ob.zero_grad()
        lb = train_step_amp(mb, batch, ob, scaler)
        assert float(la) == float(lb), (t, float(la), float(lb))
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), n
    assert scaler.get_scale() == 65536.0
    if cd == "fp32":                                         # and on the golden batch the scaled loop follows the reference trajectory
        gold = {"input_ids": T(TINY["ids"]).to(DEV), "attention_mask": T(TINY["mask"]).to(DEV), "labels": T(TINY["ids"]).clone().to(DEV)}
        mc = build(V, H, L, nh, cd)
        oc = AdamW(mc.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
        for t in range(4):
            oc.zero_grad()
            lc = train_step_amp(mc, gold, oc, scaler)
            assert abs(float(lc) - TINY["traj"][t, 0]) <= 1e-5 * TINY["traj"][t, 0]


def test_overflow_skips_and_torch_scaler_interop():
    from test_gpu_bloom import build
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    batch = {"input_ids": T(TINY["ids"]).to(DEV), "attention_mask": T(TINY["mask"]).to(DEV), "labels": T(TINY["ids"]).clone().to(DEV)}
    m = build(V, H, L, nh)
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    scaler = GradScaler(init_scale=1024.0, growth_interval=2)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    (loss, _, _), _ = m(**batch)
    scaler.scale(loss).backward()
    m.bloom.ln_f.bias.grad[3] = float("inf")
    assert scaler.step(opt) is None
    scaler.update()
    assert scaler.get_scale() == 512.0 and opt.steps[0] == 1
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    # the caller's own torch.cuda.amp.GradScaler() (ft_bloom_DDP.py:109) drives the fused optimizer too
    ts = torch.amp.GradScaler("cuda", init_scale=256.0)
    for t in range(2):
        opt.zero_grad()
        (loss, _, _), _ = m(**batch)
        ts.scale(loss).backward()
        ts.step(opt)
        ts.update()
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 1e-5 * TINY["traj"][t, 0]
    assert opt.steps[0] == 3


def test_scaler_registration_ends_with_the_scaler_and_a_moved_scale_is_noticed():
    """Round-4 advisor: (a) the loss scale a GradScaler registers for the fused loss (ops.set_expected_loss_grad) is held by weak reference:
    after the scaler is deleted or disabled no later loss folds it; (b) the loss node snapshots the scale it folded — when the scaler's live
    scale moves between forward and backward the rescale is NOT skipped: dlogits equal the unscaled gradient x the scale that backward sends."""
    import gc
    from cleantransformer_amd import ops
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.models.modeling_bloom import ShiftedCrossEntropyFn
    ops.set_expected_loss_grad(factor=1.0, scale=False)
    g = torch.Generator().manual_seed(5)
    B, S, V = 2, 8, 4096
    logits0 = torch.randn(B, S, V, generator=g).to(DEV).to(torch.bfloat16)
    labels = torch.randint(0, V, (B, S), generator=g).to(DEV)

    def dlogits(upstream):
        lg = logits0.clone().requires_grad_(True)
        loss = ShiftedCrossEntropyFn.apply(lg, labels)
        return lg, loss

    lg, loss = dlogits(None)
    loss.backward()
    plain = lg.grad.float().clone()

    sc = GradScaler(init_scale=256.0)
    lg, loss = dlogits(None)
    scaled = sc.scale(loss)                                                  # first use: registers the device scale (for the NEXT forward)
    scaled.backward()
    assert torch.equal(lg.grad.float(), plain * 256.0)
    assert ops.current_expected_loss_grad()[1] is not None
    # the next forward folds 256; then the scale moves to 64 BEFORE backward: the node must rescale by 64 / 256, not skip
    lg, loss = dlogits(None)
    sc.update(new_scale=64.0)
    sc.scale(loss).backward()
    assert torch.allclose(lg.grad.float(), plain * 64.0, rtol=2.0 ** -7, atol=0.0)      # (one extra bf16 rounding from the rescale pass)
    # the scaler goes away: no later loss folds its scale
    del sc, scaled
    gc.collect()
    assert ops.current_expected_loss_grad()[1] is None
    lg, loss = dlogits(None)
    loss.backward()
    assert torch.equal(lg.grad.float(), plain)
    off = GradScaler(init_scale=8.0, enabled=False)
    assert off.scale(loss.detach()) is not None and ops.current_expected_loss_grad()[1] is None


# ------------------------------------------------------------------------------------------------ fp16 (round 5)
def _golden_batch():
    return {"input_ids": T(TINY["ids"]).to(DEV), "attention_mask": T(TINY["mask"]).to(DEV), "labels": T(TINY["ids"]).clone().to(DEV)}


@pytest.mark.parametrize("how", ["config", "autocast"])
def test_fp16_tiny_trajectory_with_dynamic_loss_scaling_tracks_the_fp32_golden(how):
    """The reference's published DDP launch trains under torch.autocast (fp16 on a GPU) with a GradScaler (ft_bloom_DDP.py:107-128,
    scripts/ft_bloom_DDP.sh:11).  Here: compute_dtype "fp16" — or an fp32-configured model under amp.autocast(dtype=torch.float16) — with the
    package's GradScaler at its default initial scale, four steps on the golden batch: the loss follows the reference's fp32 trajectory within the
    bf16 bars (fp16 has three more mantissa bits: it lands well inside), gradients are fp32, activations fp16, the scaler never backs off, and the
    gradient norm after unscaling matches the golden one."""
    import math
    from test_gpu_bloom import build
    from cleantransformer_amd.amp import GradScaler, autocast
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    batch = _golden_batch()
    m = build(V, H, L, nh, "fp16" if how == "config" else "fp32")
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    scaler = GradScaler()
    for t in range(4):
        opt.zero_grad()
        if how == "config":
            (loss, logits, _), _ = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        else:
            with autocast(dtype=torch.float16):
                (loss, logits, _), _ = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        assert logits.dtype == torch.float16
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))
        assert all(p.grad.dtype == torch.float32 for p in m.parameters())
        assert scaler.step(opt) is None or True                       # (AdamW.step returns None)
        scaler.update()
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 3e-3 * TINY["traj"][t, 0], (t, float(loss), TINY["traj"][t, 0])
        assert abs(gn - TINY["traj"][t, 1]) <= 3e-2 * TINY["traj"][t, 1], (t, gn, TINY["traj"][t, 1])
    assert scaler.get_scale() == 65536.0 and opt.steps[0] == 5
    if how == "autocast":                                             # outside the context the same model is an fp32 model again
        (loss, logits, _), _ = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        assert logits.dtype == torch.float32


def test_fp16_overflow_is_found_and_skips_the_step_then_training_resumes():
    """An fp16 run whose loss scale is too large overflows IN the half-precision gradients (not by hand-injected infs): found_inf is raised by the
    unscale kernel, the step is skipped, the scale halves, Adam's step count does not advance; a few skipped steps later the scale fits and the
    loop trains (the loss falls)."""
    from test_gpu_bloom import build
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    batch = _golden_batch()
    m = build(V, H, L, nh, "fp16")
    opt = AdamW(m.parameters(), lr=1e-3, weight_decay=0.0, decoupled=True)
    scaler = GradScaler(init_scale=2.0 ** 30, growth_interval=1000)      # dlogits ~ 2^30 / (B (S-1)) overflows 65504
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    skipped, losses = 0, []
    for t in range(40):
        opt.zero_grad()
        (loss, _, _), _ = m(**batch)
        scaler.scale(loss).backward()
        s0 = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        if scaler.get_scale() < s0:
            skipped += 1
            assert opt.steps[0] == 1 + (t + 1 - skipped)
            if skipped == 1:
                for n, p in m.named_parameters():
                    assert torch.equal(p.detach(), before[n]), n          # a skipped step changes nothing
        else:
            losses.append(float(loss))
    assert skipped >= 5 and scaler.get_scale() == 2.0 ** (30 - skipped), (skipped, scaler.get_scale())
    assert len(losses) >= 20 and losses[-1] < losses[0] - 0.05, losses


def test_fp16_c1_config_trajectory_tracks_the_reference():
    """BASELINE configs[0] (Bloom-560M 2-layer slice, B=2 S=128, full vocabulary) in fp16 with dynamic loss scaling against the reference-generated
    fp32 trajectory of tests/golden/c1_bloom.json: loss 3e-3, gradient norm 3e-2 (the bf16 bars), token ids of the first forward equal on
    >= 99 % of the positions (half-precision logits may flip near-ties among 250 880 candidates)."""
    import json
    import math
    from test_gpu_bloom import build
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.optimizer import AdamW
    doc = json.load(open(os.path.join(HERE, "golden", "c1_bloom.json")))
    c = doc["cfg"]
    m = build(c["V"], c["H"], c["L"], c["nh"], "fp16")
    ids = torch.randint(0, c["V"], (c["B"], c["S"]), generator=torch.Generator().manual_seed(7))
    am = torch.ones(c["B"], c["S"], dtype=torch.long)
    am[c["pad_row"], c["pad_from"]:] = 0
    ids, am = ids.to(DEV), am.to(DEV)
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    scaler = GradScaler()
    for t in range(4):
        (loss, logits, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))
        scaler.step(opt)
        scaler.update()
        if t == 0:
            agree = (logits.argmax(-1).cpu() == torch.tensor(doc["argmax"])).float().mean()
            assert float(agree) >= 0.99, float(agree)
        assert abs(float(loss) - doc["traj"][t][0]) <= 3e-3 * doc["traj"][t][0], (t, float(loss), doc["traj"][t])
        assert abs(gn - doc["traj"][t][1]) <= 3e-2 * doc["traj"][t][1], (t, gn, doc["traj"][t])
    assert scaler.get_scale() == 65536.0


@pytest.mark.parametrize("who", ["none", "ours_first_step", "torch_scaler"])
def test_fp16_dlogits_at_full_vocabulary_do_not_underflow_before_the_scale(who):
    """Round-5 advisor: at V = 250 880 the softmax part of dlogits, p / N, is ~5e-10 — below the smallest half subnormal (6e-8).  Whatever
    scaler drives the loop, and on its FIRST step too, the 2^16 loss scale must reach dlogits in fp32 BEFORE the half cast (as the reference's
    fp32 CE under autocast does): (a) no registered device scale (plain backward of a pre-scaled loss / torch.amp.GradScaler) -> the
    two-pass loss; (b) this package's GradScaler registers its scale at construction -> the fused one-pass loss already folds it."""
    import gc
    from cleantransformer_amd import ops
    from cleantransformer_amd.amp import GradScaler
    from cleantransformer_amd.models.modeling_bloom import ShiftedCrossEntropyFn
    gc.collect()
    ops.set_expected_loss_grad(factor=1.0, scale=False)
    g = torch.Generator().manual_seed(11)
    B, S, V = 2, 256, 250880
    logits = (torch.randn(B, S, V, generator=g) * 2.0).to(DEV).to(torch.float16)
    labels = torch.randint(0, V, (B, S), generator=g).to(DEV)
    labels[0, 5:9] = -100
    SCALE = 65536.0
    lg = logits.clone().requires_grad_(True)
    sc = None
    if who == "ours_first_step":
        sc = GradScaler(init_scale=SCALE)
        assert ops.current_expected_loss_grad()[1] is not None             # registered before any scale() call
        loss = ShiftedCrossEntropyFn.apply(lg, labels)
        sc.scale(loss).backward()
    elif who == "torch_scaler":
        sc = torch.amp.GradScaler("cuda", init_scale=SCALE)
        loss = ShiftedCrossEntropyFn.apply(lg, labels)
        sc.scale(loss).backward()
    else:
        loss = ShiftedCrossEntropyFn.apply(lg, labels)
        (loss * SCALE).backward()
    # fp32 restatement of modeling_bloom.py:224-230 on the same half logits
    x = logits[:, :-1].float().reshape(-1, V)
    y = labels[:, 1:].reshape(-1)
    keep = y != -100
    lse = torch.logsumexp(x, dim=-1, keepdim=True)
    n = int(keep.sum())
    got = lg.grad[:, :-1].reshape(-1, V)
    assert torch.all(lg.grad[:, -1] == 0)
    rows = torch.arange(0, x.shape[0], 37, device=DEV)                      # a sample of rows, every column
    ref = torch.exp(x[rows] - lse[rows]) * (SCALE / n)
    ref[torch.arange(len(rows), device=DEV), y[rows].clamp(min=0)] -= SCALE / n
    ref = ref * keep[rows].unsqueeze(1)
    refh = ref.to(torch.float16).float()
    gh = got[rows].float()
    # the softmax terms survive: most of a kept row is non-zero, and the values are the fp32 gradient rounded once to half
    kept_rows = keep[rows]
    assert float((gh[kept_rows] != 0).float().mean()) > 0.5, "dlogits flushed to zero: the loss scale arrived after the half cast"
    err = (gh - refh).abs().max()
    tol = 2.0 ** -9 * float(ref.abs().max()) + 2.0 ** -24                  # one half rounding (+ a subnormal step)
    assert float(err) <= tol, (float(err), tol)
    assert abs(float(loss) - float((lse.squeeze(1) - x.gather(1, y.clamp(min=0).unsqueeze(1)).squeeze(1))[keep].mean())) < 1e-3
    del sc
    gc.collect()
    ops.set_expected_loss_grad(factor=1.0, scale=False)
