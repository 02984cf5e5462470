"""CPU (no GPU): dropout host logic (round 2).
 1. the Python restatement of the kernels' mask function (cleantransformer_amd/rng.py) equals the library's own host export
    (ctmi_dropout_hash / ctmi_dropout_threshold) — the GPU tests then compare kernels with masks built from it;
 2. the product's dropout paths (Bloom block with hidden / attention dropout, GPT-2 with its default dropouts, the generic
    post-LN TransformerBlock) through the torch-CPU emulation of the kernel contracts, against a plain torch-autograd
    restatement of the reference's module math with the SAME explicit masks: torch.nn.Dropout semantics (Bernoulli keep,
    1/(1-p) scale, mask on the normalised attention probabilities, mask reused by the backward), seeds drawn from torch's CPU
    generator (reproducible under torch.manual_seed, like the reference's modules)."""
import math

import numpy as np
import pytest
import torch

import cpu_kernel_emulation as emu
from cleantransformer_amd import rng
from oracle import bloom_ref as R
from test_host_logic_cpu import build

F = torch.nn.functional


def test_python_hash_equals_library_host_export():
    from cleantransformer_amd import _lib
    lib = _lib.load()
    xs = [0, 1, 2, 0xFFFFFFFF, 0x9E3779B9, 123456789, 2 ** 31, 2 ** 31 - 1] + [int(v) for v in torch.randint(0, 2 ** 32, (2000,), dtype=torch.int64)]
    got = rng.hash32(torch.tensor(xs, dtype=torch.int64))
    for x, g in zip(xs, got.tolist()):
        assert lib.ctmi_dropout_hash(x) == g, x
    for seed in (0, 1, 0x80000000, 0xFFFFFFFF, 0x12345678):
        got2 = rng.keep_hash(torch.tensor(xs[:200], dtype=torch.int64), seed)
        for x, g in zip(xs[:200], got2.tolist()):
            assert lib.ctmi_dropout_keep_hash(x, seed) == g, (x, seed)
    # two seeds are two functions, not two windows of one sequence: relabelling the counters by the xor of the seeds does NOT map one
    # mask onto the other (it would with keep(i) = hash32(i ^ seed))
    c = torch.arange(1 << 14, dtype=torch.int64)
    s1, s2 = 0x0001ABCD, 0x7001ABCD
    same = (rng.keep_hash(c ^ (s1 ^ s2), s1) >= 2 ** 31) == (rng.keep_hash(c, s2) >= 2 ** 31)
    assert 0.45 < float(same.float().mean()) < 0.55
    for p in (0.0, 0.1, 0.25, 0.5, 0.999, 1e-9):
        assert lib.ctmi_dropout_threshold(p) == rng.drop_threshold(p), p
    # avalanche sanity: one flipped input bit flips ~half of the output bits
    a, b = rng.hash32(torch.arange(4096)), rng.hash32(torch.arange(4096) ^ 1)
    flips = sum(bin(int(v)).count("1") for v in (a ^ b).tolist()) / 4096
    assert 13 < flips < 19, flips


def test_seeds_follow_torch_manual_seed():
    torch.manual_seed(11)
    a = [rng.next_seed() for _ in range(4)]
    torch.manual_seed(11)
    b = [rng.next_seed() for _ in range(4)]
    assert a == b and len(set(a)) == 4 and all(0 <= s < 2 ** 32 for s in a)


def _keep(n_or_shape, seed, p):
    n = int(np.prod(n_or_shape))
    return (rng.keep_hash(torch.arange(n, dtype=torch.int64), seed) >= rng.drop_threshold(p)).view(n_or_shape)


def _drop(x, seed, p):
    return torch.where(_keep(tuple(x.shape), seed, p), x / (1.0 - p), torch.zeros(())) if p > 0 else x


def _bloom_block_with_masks(p, i, x, am, sh, ph, pa, seeds):
    """modeling_bloom.py:142-159 with its dropouts (:111, :122, :270), masks made explicit."""
    pre = f"bloom.blocks.{i}."
    B, S, H = x.shape
    nh, hd = sh.n_head, H // sh.n_head
    s_attn, s_h1, s_h2 = seeds
    ln1 = R.layernorm(x, p[pre + "input_layernorm.weight"], p[pre + "input_layernorm.bias"], sh.eps)
    qkv = F.linear(ln1, p[pre + "self_attention.query_key_value.weight"], p[pre + "self_attention.query_key_value.bias"])
    xq = qkv.view(B, S, nh, 3, hd)
    q, k, v = (xq[..., j, :].transpose(1, 2) for j in range(3))
    scores = R.build_alibi(am, nh).view(B, nh, 1, S) + torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
    scores = torch.masked_fill(scores, R.causal_key_mask(am, S), torch.finfo(torch.float32).min)
    probs = _drop(torch.softmax(scores, dim=-1), s_attn, pa)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, S, H)
    d = F.linear(ctx, p[pre + "self_attention.dense.weight"], p[pre + "self_attention.dense.bias"])
    h1 = x + _drop(d.reshape(B * S, H), s_h1, ph).view(B, S, H)
    ln2 = R.layernorm(h1, p[pre + "post_attention_layernorm.weight"], p[pre + "post_attention_layernorm.bias"], sh.eps)
    u = F.linear(ln2, p[pre + "mlp.dense_h_to_4h.weight"], p[pre + "mlp.dense_h_to_4h.bias"])
    m = F.linear(R.gelu_tanh(u), p[pre + "mlp.dense_4h_to_h.weight"], p[pre + "mlp.dense_4h_to_h.bias"])
    return h1 + _drop(m.reshape(B * S, H), s_h2, ph).view(B, S, H)


@pytest.mark.parametrize("ph,pa", [(0.1, 0.2), (0.0, 0.3), (0.25, 0.0)])
def test_bloom_with_dropout_equals_explicit_mask_restatement(monkeypatch, ph, pa):
    emu.install(monkeypatch)
    V, H, L, nh, B, S = 211, 64, 2, 8, 3, 16
    sh = R.BloomShape(V, H, L, nh)
    m = build(V, H, L, nh)
    for blk in m.bloom.blocks:
        blk.hidden_dropout = ph
        blk.self_attention.attention_dropout.p = pa
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(3))
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 12:] = 0
    torch.manual_seed(5)
    (loss, logits, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    loss.backward()
    # restatement with the same seeds (three per block, in forward order)
    torch.manual_seed(5)
    seeds = [(rng.next_seed(), rng.next_seed(), rng.next_seed()) for _ in range(L)]
    p = {n: v.clone().requires_grad_(True) for n, v in R.det_init(sh).items()}
    emb = p["bloom.word_embeddings.weight"]
    x = R.layernorm(F.embedding(ids, emb), p["bloom.word_embeddings_layernorm.weight"], p["bloom.word_embeddings_layernorm.bias"], sh.eps)
    for i in range(L):
        x = _bloom_block_with_masks(p, i, x, am, sh, ph, pa, seeds[i])
    hid = R.layernorm(x, p["bloom.ln_f.weight"], p["bloom.ln_f.bias"], sh.eps)
    lg = F.linear(hid, emb)
    ref = R.cross_entropy(lg[:, :-1].reshape(-1, V), ids[:, 1:].reshape(-1))
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref), (float(loss), float(ref))
    assert torch.allclose(logits, lg, rtol=1e-4, atol=1e-5)
    for n, prm in m.named_parameters():
        if n == "lm_head.weight":
            continue
        assert torch.allclose(prm.grad, p[n].grad, rtol=2e-4, atol=2e-7), (n, float((prm.grad - p[n].grad).abs().max()))
    # eval(): dropout off — identical to the p = 0 model
    m.eval()
    with torch.no_grad():
        (l_eval, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    l0, _, _, _ = R.bloom_forward(R.det_init(sh), sh, ids, am, labels=ids)
    assert abs(float(l_eval) - float(l0)) <= 1e-5 * float(l0)


def test_gpt2_default_dropouts_train_and_are_reproducible(monkeypatch):
    """GPT-2 with the reference's DEFAULT dropouts (embd / attn / resid 0.1 and the MLP's torch.nn.Dropout() = 0.5):
    round 1 raised here.  Same torch.manual_seed -> same loss and gradients; another seed -> different; eval() is deterministic."""
    emu.install(monkeypatch)
    from cleantransformer_amd.models.modeling_gpt import GPTConfig, GPTLMHeadModel
    cfg = GPTConfig(vocab_size=173, n_embd=64, n_positions=64, n_layer=2, n_head=4, n_ctx=64)
    assert cfg.embd_pdrop == 0.1 and cfg.attn_pdrop == 0.1 and cfg.resid_pdrop == 0.1
    torch.manual_seed(0)
    m = GPTLMHeadModel(cfg, version="gpt2").train()
    assert m.gpt.blocks[0].mlp[3].p == 0.5
    ids = torch.randint(0, 173, (3, 24), generator=torch.Generator().manual_seed(7))
    am = torch.ones(3, 24, dtype=torch.long)

    def run(seed):
        torch.manual_seed(seed)
        for prm in m.parameters():
            prm.grad = None
        (loss, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
        loss.backward()
        return float(loss), torch.cat([prm.grad.reshape(-1) for prm in m.parameters()])
    l1, g1 = run(1)
    l2, g2 = run(1)
    l3, g3 = run(2)
    assert l1 == l2 and torch.equal(g1, g2)
    assert l1 != l3 and not torch.equal(g1, g3)
    assert math.isfinite(l1) and torch.isfinite(g1).all()
    m.eval()
    with torch.no_grad():
        (e1, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
        (e2, _, _), _ = m(ids, attention_mask=am, labels=ids.clone())
    assert float(e1) == float(e2)


def test_generic_block_example_config_trains(monkeypatch):
    """transformer.py's own ExampleConfig (attention_probs_dropout_prob = hidden_dropout_prob = 0.1, :124-131) in training mode
    against the module math with explicit masks."""
    emu.install(monkeypatch)
    from cleantransformer_amd.transformer import ExampleConfig, TransformerBlock
    cfg = ExampleConfig()
    torch.manual_seed(3)
    blk = TransformerBlock(cfg).train()
    x = torch.randn(2, 5, cfg.hidden_size, generator=torch.Generator().manual_seed(9)).requires_grad_(True)
    torch.manual_seed(21)
    y = blk(x)
    y.sum().backward()
    torch.manual_seed(21)
    s_attn, s_h1, s_h2 = rng.next_seed(), rng.next_seed(), rng.next_seed()
    xr = x.detach().clone().requires_grad_(True)
    a = blk.attention
    B, S, H = xr.shape
    nh, hd = cfg.num_attention_heads, H // cfg.num_attention_heads
    q, k, v = (F.linear(xr, l.weight, l.bias).view(B, S, nh, hd).transpose(1, 2) for l in (a.q_linear, a.k_linear, a.v_linear))
    pr = _drop(torch.softmax(torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd), dim=-1), s_attn, 0.1)
    att = torch.matmul(pr, v).transpose(1, 2).reshape(B, S, H)
    y1 = R.layernorm(xr + _drop(att.reshape(-1, H), s_h1, 0.1).view(B, S, H), blk.norm1.weight, blk.norm1.bias, cfg.layer_norm_epsilong)
    f = F.linear(torch.relu(F.linear(y1, blk.ffw[0].weight, blk.ffw[0].bias)), blk.ffw[2].weight, blk.ffw[2].bias)
    y2 = R.layernorm(y1 + _drop(f.reshape(-1, H), s_h2, 0.1).view(B, S, H), blk.norm2.weight, blk.norm2.bias, cfg.layer_norm_epsilong)
    y2.sum().backward()
    assert torch.allclose(y, y2, rtol=1e-4, atol=1e-5)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-3, atol=1e-5)
