"""GPU: the SFT step replayed from a hipGraph (cleantransformer_amd/graph.py GraphedStep) is the eager step of ft_bloom.py:79-90 — same launches,
same order: identical losses and parameters on the golden tiny model (against the REFERENCE's trajectory too), fresh data reaching every
replay, a learning-rate change between replays taking effect, a shape change falling back to eager, and bit-identity of the device-record
AdamW (ctmi_adamw_step_dev) with the host-argument form."""
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


def _model(cd="fp32"):
    from test_gpu_bloom import build
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    return build(V, H, L, nh, compute_dtype=cd)


def _eager_step(m, opt, ids, am, labels):
    outputs, _ = m(input_ids=ids, attention_mask=am, labels=labels)
    loss = outputs[0]
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach()


def test_graphed_step_reproduces_the_reference_trajectory():
    """4 steps on the golden batch: 2 eager warm-up calls, capture, 2 replays — loss_t against the reference run (tests/golden/tiny_bloom.npz: torch.optim.AdamW
    (lr=1e-5) on the reference model), final parameters against its p4_*."""
    from cleantransformer_amd.graph import GraphedStep
    from cleantransformer_amd.optimizer import AdamW
    m = _model()
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    step = GraphedStep(m, opt, warmup=2)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    for t in range(4):
        loss = step(ids, am, ids.clone())
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 1e-5 * TINY["traj"][t, 0], (t, float(loss))
    assert step.replays == 2 and step.graph is not None and step.fallback_reason is None
    for n, p in m.named_parameters():
        ref = T(TINY["p4_" + n]).double()
        err = (p.detach().double().cpu() - ref).abs()
        assert bool((err <= 1e-7 + 1e-5 * ref.abs()).all()), (n, float(err.max()))


@pytest.mark.parametrize("cd", ["fp32", "bf16"])
def test_graphed_step_equals_eager_on_fresh_batches_with_a_scheduler(cd):
    """8 steps, a DIFFERENT batch every step (ids and padding), the learning rate changed after step 5: the graphed loop and the eager loop see the
    same numbers.  fp32 atomics in the embedding backward make neither loop bit-reproducible, so the bar is 1e-6 relative on the loss and
    a few percent of the distance travelled on the parameters — a frozen step count, a stale batch or a stale lr would be orders of magnitude off."""
    from cleantransformer_amd.graph import GraphedStep
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    g = torch.Generator().manual_seed(3)
    batches = []
    for t in range(8):
        ids = torch.randint(0, V, (B, S), generator=g)
        am = torch.ones(B, S, dtype=torch.long)
        am[t % B, S - 1 - (t % 3):] = 0
        batches.append((ids.to(DEV), am.to(DEV)))
    runs = []
    for graphed in (False, True):
        m = _model(cd)
        p0 = {n: p.detach().clone() for n, p in m.named_parameters()}
        opt = AdamW(m.parameters(), lr=1e-3, weight_decay=0.01, decoupled=True)
        step = GraphedStep(m, opt, warmup=2, enabled=graphed)
        losses = []
        for t, (ids, am) in enumerate(batches):
            if t == 5:
                opt.lr = 3e-4
            losses.append(float(step(ids, am, ids.clone())))
        if graphed:
            assert step.replays == 6, (step.replays, step.fallback_reason)
        assert opt.steps[0] == 9
        runs.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    (la, pa), (lb, pb) = runs
    tol = 1e-6 if cd == "fp32" else 2e-3
    for t in range(8):
        assert abs(la[t] - lb[t]) <= tol * abs(la[t]), (t, la[t], lb[t])
    assert la[0] != la[7]
    # Adam normalises every element's update to ~lr whatever its gradient, so an element whose gradient is at the noise level of the embedding
    # atomics may move differently in two runs of the SAME loop; the parameters are compared as a whole, relative to how far they travelled
    num = sum(float((pa[n].double() - pb[n].double()).pow(2).sum()) for n in pa) ** 0.5
    den = sum(float((pa[n].double() - p0[n].double()).pow(2).sum()) for n in pa) ** 0.5
    assert den > 0 and num / den < (2e-2 if cd == "fp32" else 0.25), (num, den)


def test_graphed_step_falls_back_on_a_new_shape_and_in_eval_mode():
    from cleantransformer_amd.graph import GraphedStep
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    m = _model()
    opt = AdamW(m.parameters(), lr=1e-4, decoupled=True)
    step = GraphedStep(m, opt, warmup=1)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    for _ in range(3):
        step(ids, am, ids.clone())
    assert step.replays == 2
    short = ids[:, :S // 2].contiguous()
    l1 = step(short, am[:, :S // 2].contiguous(), short.clone())              # another shape: eager, graph dropped
    assert step.graph is None and torch.isfinite(l1)
    for _ in range(3):
        step(short, am[:, :S // 2].contiguous(), short.clone())              # ... and captured again once it persists
    assert step.graph is not None and step.replays >= 3
    assert opt.steps[0] == 1 + 7                                              # every call was exactly one optimizer step


def test_adamw_device_record_form_is_bit_identical_to_host_arguments():
    from cleantransformer_amd import ops
    g = torch.Generator().manual_seed(5)
    sizes = [5, 1024, 16384, 40001, 1 << 17]

    def make():
        ps = [torch.randn(n, generator=torch.Generator().manual_seed(10 + i)).to(DEV) for i, n in enumerate(sizes)]
        return ps, [torch.zeros_like(x) for x in ps], [torch.zeros_like(x) for x in ps], [torch.zeros(x.numel(), dtype=torch.bfloat16, device=DEV) for x in ps]
    out = []
    for dev_form in (False, True):
        ps, ms, vs, sh = make()
        hyper = torch.zeros(12, device=DEV)
        for t in range(1, 5):
            gs = [torch.randn(x.numel(), generator=torch.Generator().manual_seed(100 * t + i)).to(DEV) for i, x in enumerate(ps)]
            kw = dict(lr=1e-2 / t, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=t, decoupled=(t % 2 == 0), mutate_grad=True, grad_scale=0.25 * t)
            if dev_form:
                ops.adamw_set_hyper(hyper, **kw)
                ops.adamw_step_dev(ps, gs, ms, vs, sh, hyper)
            else:
                ops.adamw_step(ps, gs, ms, vs, sh, **kw)
        torch.cuda.synchronize()
        out.append((ps, ms, vs, sh))
    for k in range(4):
        for a, b in zip(out[0][k], out[1][k]):
            assert torch.equal(a, b), k
