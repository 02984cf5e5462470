"""GPU (-m gpu): dropout (round 2) — the counter-based keep mask of the kernels (include/ctmi355.h ctmi_dropout, ctmi_attn_desc.dropout_*)
against its host-side restatement (cleantransformer_amd/rng.py keep_hash; tests/cpu_kernel_emulation.py applies it with plain torch ops):
because the mask is a pure function of (element counter, seed), dropout is checked for PARITY — the kernel output must equal the
explicit-mask computation to rounding — not only through statistical properties.  The reference draws its masks from torch's RNG
stream (torch.nn.Dropout), which no other implementation can reproduce; what is pinned to the reference is the semantics: Bernoulli(1-p)
keep, 1/(1-p) scaling, mask on the NORMALISED attention probabilities, same mask in forward and backward."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_kernel_emulation as EMU  # noqa: E402

DEV = "cuda:0"


def ops():
    from cleantransformer_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,p", [(8192 * 64, 0.1), (1000003, 0.5), (37, 0.25), (4096, 0.0)])
def test_dropout_kernel_equals_counter_mask(dtype, n, p):
    o = ops()
    x, r = rnd(n, seed=1).to(dtype), rnd(n, seed=2).to(dtype)
    seed = 0x9E3779B9 ^ n
    y = o.dropout(x.to(DEV), p, seed)
    want = EMU.dropout(x, p, seed)
    assert torch.equal(y.cpu(), want)                                            # x * scale in fp32, rounded once: bit-exact
    yr = o.dropout(x.to(DEV), p, seed, residual=r.to(DEV))
    assert torch.equal(yr.cpu(), EMU.dropout(x, p, seed, residual=r))
    if n > 10000 and p > 0:
        rate = float((y == 0).float().mean())
        assert abs(rate - p) < 5 * math.sqrt(p * (1 - p) / n) + 1e-4, rate      # Bernoulli(p) drop rate, 5 sigma
        assert abs(float(y.float().sum() / x.to(DEV).float().abs().sum())) < 0.05   # E[y] = x (zero-mean input: sum stays small)
    # other seed: other mask; same seed: same mask (the backward relies on it)
    if p > 0 and n > 100:
        assert not torch.equal(o.dropout(x.to(DEV), p, seed + 1), y)
        assert torch.equal(o.dropout(x.to(DEV), p, seed), y)
    # an unaligned view takes the scalar path
    if n > 64:
        xv = x.to(DEV)[3:]
        assert torch.equal(o.dropout(xv, p, seed).cpu(), EMU.dropout(x[3:], p, seed))


def test_dropout_autograd_node_uses_the_same_mask_backward():
    o = ops()
    x = rnd(64, 96, seed=3).to(DEV).requires_grad_(True)
    res = rnd(64, 96, seed=4).to(DEV).requires_grad_(True)
    y = o.DropoutFn.apply(x, 0.3, 777, res)
    g = rnd(64, 96, seed=5).to(DEV)
    y.backward(g)
    keep = (o.dropout(torch.ones_like(x), 0.3, 777) != 0)
    assert torch.equal(x.grad, torch.where(keep, g * (1.0 / 0.7), torch.zeros((), device=DEV)).to(x.dtype)) or relerr(x.grad, torch.where(keep, g / 0.7, torch.zeros((), device=DEV))) < 1e-6
    assert torch.equal(res.grad, g)


ATT = [(2, 64, 2, 64, "none"), (1, 200, 3, 64, "right"), (2, 130, 2, 32, "left"), (1, 96, 2, 128, "none"), (2, 257, 4, 64, "mixed")]


def _mask(kind, B, S):
    am = torch.ones(B, S, dtype=torch.long)
    if kind in ("right", "mixed"):
        am[0, S - S // 4:] = 0
    if kind in ("left", "mixed"):
        am[B - 1, :max(1, S // 3)] = 0
    return am


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,S,nh,hd,kind", ATT)
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_attention_dropout_equals_explicit_mask_computation(dtype, rtol, B, S, nh, hd, kind, p):
    """Fused attention with probability dropout, forward and backward, against softmax -> explicit mask -> matmul on the CPU with the
    mask restated from (counter, seed): ALiBi + causal + padding (incl. uniform all-masked rows), every head-dim tile, both dtypes."""
    o = ops()
    from cleantransformer_amd.models.modeling_bloom import alibi_slopes
    H = nh * hd
    seed = 1234567 + S
    qkv = rnd(B * S, 3 * H, seed=11).to(dtype)
    go = rnd(B * S, H, seed=12).to(dtype)
    am = _mask(kind, B, S)
    slopes = alibi_slopes(nh)
    # device
    desc = o.fused_qkv_desc(B, S, nh, hd, causal=True, dropout_p=p, dropout_seed=seed)
    qd, gd = qkv.to(DEV), go.to(DEV)
    mask = o.MaskInfo(am.to(DEV))
    out = torch.empty((B * S, H), dtype=dtype, device=DEV)
    sm, sl = o.attn_fwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, desc, slopes.to(DEV), mask)
    dq = torch.zeros_like(qd)
    o.attn_bwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, gd, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes.to(DEV), mask)
    torch.cuda.synchronize()
    # explicit-mask computation on the values the kernel saw (fp32 math)
    qc, gc = qkv.float(), go.float()
    oc = torch.empty(B * S, H)
    mc = EMU.MaskInfo(am)
    m_, l_ = EMU.attn_fwd(qc, qc[:, hd:], qc[:, 2 * hd:], oc, desc, slopes, mc)
    dqc = torch.zeros_like(qc)
    EMU.attn_bwd(qc, qc[:, hd:], qc[:, 2 * hd:], oc, gc, m_, l_, dqc, dqc[:, hd:], dqc[:, 2 * hd:], desc, slopes, mc)
    assert relerr(out.float(), oc) < rtol, relerr(out.float(), oc)
    assert relerr(dq.float(), dqc) < 3 * rtol, relerr(dq.float(), dqc)
    # p really dropped something, and p = 0 is a different (the plain) result
    desc0 = o.fused_qkv_desc(B, S, nh, hd, causal=True)
    out0 = torch.empty_like(out)
    o.attn_fwd(qd, qd[:, hd:], qd[:, 2 * hd:], out0, desc0, slopes.to(DEV), mask)
    assert relerr(out.float(), out0.float()) > 0.05


def test_generic_attention_with_additive_mask_and_dropout():
    """The additive-mask (transformer.py AttentionLayer) kernel variant with dropout against the explicit-mask computation."""
    o = ops()
    B, S, nh, hd = 2, 80, 2, 64
    H = nh * hd
    p, seed = 0.1, 4242
    q, k, v, go = (rnd(B * S, H, seed=s) for s in (1, 2, 3, 4))
    addm = torch.where(rnd(B, 1, S, S, seed=5) > 1.0, torch.full((), -1e9), torch.zeros(())).contiguous()
    st = (S * H, hd, H)
    desc = o._strided_desc(B, nh, S, S, hd, st, st, st, st, 1.0 / math.sqrt(hd), False, am_str=(S * S, 0, S, 1), dropout_p=p, dropout_seed=seed)
    out = torch.empty((B * S, H), device=DEV)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    sm, sl = o.attn_fwd(qd, kd, vd, out, desc, None, None, add_mask=addm.to(DEV))
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    o.attn_bwd(qd, kd, vd, out, go.to(DEV), sm, sl, dq, dk, dv, desc, None, None, add_mask=addm.to(DEV))
    oc = torch.empty(B * S, H)
    m_, l_ = EMU.attn_fwd(q, k, v, oc, desc, None, None, add_mask=addm)
    dqc, dkc, dvc = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    EMU.attn_bwd(q, k, v, oc, go, m_, l_, dqc, dkc, dvc, desc, None, None, add_mask=addm)
    for name, a, b in (("out", out, oc), ("dq", dq, dqc), ("dk", dk, dkc), ("dv", dv, dvc)):
        assert relerr(a, b) < 2e-4, (name, relerr(a, b))


def _bloom_run(device, ph, pa, seed, cd="fp32"):
    from oracle import bloom_ref as R
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    V, H, L, nh, B, S = 211, 64, 2, 8, 3, 16
    m = BloomForCausalLM(BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh, compute_dtype=cd,
                                     hidden_dropout=ph, attention_dropout=pa))
    m._tie_weight()
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd)
    m._tie_weight()
    m = m.to(device).train()
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(3)).to(device)
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 12:] = 0
    torch.manual_seed(seed)
    (loss, logits, _), _ = m(input_ids=ids, attention_mask=am.to(device), labels=ids.clone())
    loss.backward()
    return float(loss), logits.detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}


@pytest.mark.parametrize("ph,pa", [(0.1, 0.2), (0.0, 0.3)])
def test_bloom_with_dropout_gpu_equals_cpu_explicit_masks(monkeypatch, ph, pa):
    """The whole Bloom model with hidden / attention dropout on the GPU against the SAME model run on the CPU emulation of the
    kernel contracts (which tests/test_dropout_cpu.py pins to a plain torch restatement with explicit masks): same
    torch.manual_seed -> same seeds -> same masks, so loss, logits and every gradient must agree to fp32 rounding."""
    with monkeypatch.context() as mp:
        EMU.install(mp)
        l_cpu, lg_cpu, g_cpu = _bloom_run("cpu", ph, pa, seed=5)
    l_gpu, lg_gpu, g_gpu = _bloom_run(DEV, ph, pa, seed=5)
    assert abs(l_gpu - l_cpu) <= 1e-5 * abs(l_cpu), (l_gpu, l_cpu)
    assert relerr(lg_gpu, lg_cpu) < 1e-5
    for n in g_cpu:
        assert relerr(g_gpu[n], g_cpu[n]) < 2e-4, (n, relerr(g_gpu[n], g_cpu[n]))
    l_other, _, _ = _bloom_run(DEV, ph, pa, seed=6)
    assert l_other != l_gpu
    # bf16 compute: same masks (they depend on counters and seeds only), values to bf16 accuracy
    l_bf, _, g_bf = _bloom_run(DEV, ph, pa, seed=5, cd="bf16")
    assert abs(l_bf - l_cpu) <= 1e-2 * abs(l_cpu)


def test_gpt2_default_dropouts_train_on_the_gpu():
    """GPT-2 with the reference's default dropouts (embd / attn / resid 0.1, MLP Dropout() 0.5) fine-tunes: reproducible under
    torch.manual_seed, masks change with the seed, eval() is deterministic, and 30 AdamW steps on one batch reduce the loss."""
    from cleantransformer_amd.models.modeling_gpt import GPTConfig, GPTLMHeadModel
    from cleantransformer_amd.optimizer import AdamW
    for cd in ("fp32", "bf16"):
        cfg = GPTConfig(vocab_size=173, n_embd=64, n_positions=64, n_layer=2, n_head=4, n_ctx=64, compute_dtype=cd)
        torch.manual_seed(0)
        m = GPTLMHeadModel(cfg, version="gpt2").to(DEV).train()
        m._tie_weights()
        ids = torch.randint(0, 173, (3, 24), generator=torch.Generator().manual_seed(7)).to(DEV)
        am = torch.ones(3, 24, dtype=torch.long, device=DEV)

        def run(seed):
            torch.manual_seed(seed)
            for prm in m.parameters():
                prm.grad = None
            (loss, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
            loss.backward()
            return float(loss.detach()), torch.cat([prm.grad.reshape(-1) for prm in m.parameters()]).clone()
        l1, g1 = run(1)
        l2, g2 = run(1)
        l3, g3 = run(2)
        # token-embedding gradient rows are scatter-added with fp32 atomics (order may differ between runs): compare to rounding
        assert abs(l1 - l2) <= 1e-6 * abs(l1) and relerr(g1, g2) < 1e-5
        assert l1 != l3 and relerr(g1, g3) > 1e-2
        opt = AdamW(m.parameters(), lr=3e-3, weight_decay=0.0, decoupled=True)
        torch.manual_seed(3)
        losses = []
        for _ in range(30):
            (loss, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(math.isfinite(x) for x in losses) and sum(losses[-5:]) / 5 < 0.8 * sum(losses[:5]) / 5, (cd, losses[:3], losses[-3:])
        m.eval()
        with torch.no_grad():
            (e1, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
            (e2, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
        assert float(e1) == float(e2)
