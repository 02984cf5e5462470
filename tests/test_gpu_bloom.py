"""GPU (-m gpu): the Bloom SFT step (forward / backward / AdamW) on the HIP path vs the golden vectors generated from
the reference and vs the CPU oracle.  Tolerances: token ids / argmax bit-exact; fp32 loss and grad-norm 1e-4 relative
(north star); bf16 mode ("loss-curve equivalent") a few 1e-3 on the loss, 3e-2 on the grad-norm.
"""
import hashlib
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bloom_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")
DEV = "cuda:0"
TINY = np.load(os.path.join(G, "tiny_bloom.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def build(V, H, L, nh, compute_dtype="fp32", params=None):
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    cfg = BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh, compute_dtype=compute_dtype)
    m = BloomForCausalLM(cfg)
    m._tie_weight()
    p = params if params is not None else R.det_init(R.BloomShape(V, H, L, nh))
    sd = dict(p)
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    m._tie_weight()
    return m.to(DEV).train()


def gnorm(m):
    return math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))


def close(name, got, ref, rtol, atol=0.0):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert torch.isfinite(got).all(), name
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} off, worst {float(err.max()):.3e}, ref scale {float(ref.abs().max()):.3e}"


def tiny_shape():
    return [int(v) for v in TINY["cfg"]]


def test_state_dict_keys_and_tied_weight():
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh)
    keys = list(m.state_dict().keys())
    assert keys[0] == "bloom.word_embeddings.weight" and keys[-1] == "lm_head.weight"
    assert [n for n, _ in m.named_parameters()] == R.param_names(R.BloomShape(V, H, L, nh))
    assert m.lm_head.weight is m.bloom.word_embeddings.weight
    h = hashlib.sha256()
    for _, p in m.named_parameters():
        h.update(p.detach().float().cpu().contiguous().numpy().tobytes())
    assert h.hexdigest().startswith("420b482203d139fb")                       # SURVEY Appendix A anchor


def test_tiny_forward_backward_matches_reference_golden():
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    (loss, logits, hidden), presents = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    close("loss0", loss, TINY["loss0"], 1e-5)
    close("logits0", logits, TINY["logits0"], 1e-4, 2e-6)
    close("hidden0", hidden, TINY["hidden0"], 1e-4, 2e-6)
    assert torch.equal(logits.argmax(-1).cpu(), T(TINY["logits0"]).argmax(-1))           # bit-exact token indices
    assert len(presents) == L and presents[0][0].shape == (B, nh, S, H // nh)
    loss.backward()
    for n, p in m.named_parameters():
        close("g0_" + n, p.grad, TINY["g0_" + n], 2e-4, 2e-7)
    gn = gnorm(m)
    assert abs(gn - TINY["traj"][0, 1]) <= 1e-4 * gn


def test_presents_after_backward_and_inplace_on_a_block_output_fail_loudly():
    """INTEGRATION.md §5 (round-3 advisor): after a differentiated forward the K/V presents are views of the block's activation slab —
    reading them before backward() works and matches the reference's tensors, reading them for the FIRST time after backward() raises a
    RuntimeError that says what to do; an in-place op on the hidden states (the output of a custom autograd node) is refused by torch at the op
    itself ("... is a view and is being modified inplace ... You can fix this by cloning") instead of producing a wrong gradient."""
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    (loss, _, _), presents = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    k0, v0 = presents[0]                                                        # read before backward: fine, and stays valid afterwards
    loss.backward()
    assert k0.shape == (B, nh, S, H // nh) and torch.isfinite(k0).all() and torch.isfinite(v0).all()
    with pytest.raises(RuntimeError, match="before backward"):
        presents[1][0]
    m.zero_grad()
    (loss, _, hidden), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    with pytest.raises(RuntimeError, match="inplace"):                          # torch refuses at the op: "... is a view and is being modified inplace"
        hidden.add_(1.0)
    loss.backward()                                                             # the graph is intact
    with torch.no_grad():                                                       # evaluation: eager copies, valid forever
        (_, _, _), pres = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    assert pres[L - 1][1].is_contiguous()


@pytest.mark.parametrize("S,force_side", [(256, True), (250, None)], ids=["grouped-launch-on-the-side-stream", "per-product-launches"])
def test_deferred_side_stream_join_equals_joined_backward_bit_for_bit(S, force_side):
    """(Round 5: geometries that take the grouped weight-gradient launch run their whole backward on ONE stream by default — T = 1024 here; the
    first case forces the side stream for them, the second uses T = 1000, which the grouped launch refuses: four products, side stream by default.)
    Round 4: with no accumulation pending and no gradient hooks the weight-gradient side stream is joined ONCE, at the end of the backward
    pass (ops.bloom_block_bwd(defer_join=True); two alternating scratch buffers, record_stream on what the side stream still touches).  Same
    kernels, same order per stream: every gradient must be bit-identical to the per-block join (CTMI_WGRAD_DEFER_JOIN=0 semantics), autograd
    must ADOPT the gradient tensors (a copy on the compute stream would read them before the side stream has written them), and with an
    existing .grad (accumulation) the node must fall back to the joined form."""
    from cleantransformer_amd import ops as o
    from cleantransformer_amd.models import modeling_bloom as MB
    V, H, L, nh, B = 512, 256, 6, 4, 4                                            # large enough for the side stream to lag the main stream
    assert o.block_wgrad_grouped(B, S, H, torch.bfloat16) == (force_side is True)
    torch.manual_seed(3)
    params = {n: (torch.randn(s) * 0.05 if len(s) > 1 else (torch.ones(s) if "layernorm.weight" in n or "ln_f.weight" in n else torch.zeros(s)))
              for n, s in ((n, R.param_shape(R.BloomShape(V, H, L, nh), n)) for n in R.param_names(R.BloomShape(V, H, L, nh)))}
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(4)).to(DEV)
    am = torch.ones(B, S, dtype=torch.long, device=DEV)

    def run(defer, steps=1, zero=True):
        old, old_side = o._DEFER_JOIN, MB._WGRAD_SIDE_STREAM
        o._DEFER_JOIN = defer
        MB._WGRAD_SIDE_STREAM = force_side
        try:
            m = build(V, H, L, nh, compute_dtype="bf16", params=params)
            for _ in range(steps):
                if zero:
                    m.zero_grad(set_to_none=True)
                (loss, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
                o._LAST_DEFERRED_GRAD_PTRS[:] = []
                loss.backward()
            torch.cuda.synchronize()
            return m, {n: p.grad.clone() for n, p in m.named_parameters()}
        finally:
            o._DEFER_JOIN, MB._WGRAD_SIDE_STREAM = old, old_side
    m1, g_join = run(False)
    assert o._LAST_DEFERRED_GRAD_PTRS == []
    m2, g_def = run(True)
    assert len(o._LAST_DEFERRED_GRAD_PTRS) == 12                                  # the last block that ran its backward (block 0) deferred
    blk0 = [p for n, p in m2.named_parameters() if n.startswith("bloom.blocks.0.")]
    assert sorted(p.grad.data_ptr() for p in blk0) == sorted(o._LAST_DEFERRED_GRAD_PTRS), "autograd copied a deferred gradient instead of adopting it"
    def same(a, b, n):
        if n == "bloom.word_embeddings.weight":                                   # tied table: the embedding backward adds its rows with fp32 atomics
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), n                  # (summation order varies from run to run, joined or not)
        else:
            assert torch.equal(a, b), n
    for n in g_join:
        same(g_join[n], g_def[n], n)
    # accumulation: the second backward finds .grad set -> joined form, and the sum is exact
    _, g_acc = run(True, steps=2, zero=False)
    assert o._LAST_DEFERRED_GRAD_PTRS == []
    for n in g_join:
        same(g_acc[n], g_join[n] * 2, n)


@pytest.mark.parametrize("which", ["fused", "torch"])
def test_tiny_four_step_trajectory(which):
    """ft_bloom.py:84-90 loop, 4 steps: loss_t and ||g||_t vs the reference run with torch.optim.AdamW(lr=1e-5)."""
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    if which == "fused":
        from cleantransformer_amd.optimizer import AdamW
        opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    else:
        opt = torch.optim.AdamW(m.parameters(), lr=1e-5)
    for t in range(4):
        (loss, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        opt.step()
        assert abs(float(loss) - TINY["traj"][t, 0]) <= 1e-5 * TINY["traj"][t, 0], (t, float(loss))
        assert abs(gn - TINY["traj"][t, 1]) <= 1e-4 * gn, (t, gn)
    for n, p in m.named_parameters():
        close("p4_" + n, p, TINY["p4_" + n], 1e-5, 1e-7)


def test_tiny_left_padding_uniform_rows():
    """Rows whose causal window is all padding become uniform over all keys (finfo.min fill)."""
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh)
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["lp_mask"]).to(DEV)
    (loss, logits, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    close("lp_loss", loss, TINY["lp_loss"], 1e-5)
    close("lp_logits", logits, TINY["lp_logits"], 1e-4, 2e-6)
    loss.backward()
    gn = gnorm(m)
    assert abs(gn - float(TINY["lp_gnorm"])) <= 1e-4 * gn
    named = dict(m.named_parameters())
    for k in TINY.files:
        if k.startswith("lp_g_"):
            close(k, named[k[5:]].grad, TINY[k], 2e-4, 2e-7)


def test_tiny_greedy_decode_bit_exact():
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh).eval()
    out = m.generate(T(TINY["greedy_prompt"]).to(DEV), attention_mask=T(TINY["greedy_mask"]).to(DEV),
                     generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    assert np.array_equal(out.cpu().numpy(), TINY["greedy_out"])


def test_tiny_bf16_mode_tracks_fp32():
    V, H, L, nh, B, S = tiny_shape()
    m = build(V, H, L, nh, compute_dtype="bf16")
    ids, am = T(TINY["ids"]).to(DEV), T(TINY["mask"]).to(DEV)
    (loss, logits, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
    assert logits.dtype == torch.bfloat16
    assert abs(float(loss) - float(TINY["loss0"])) <= 5e-3 * float(TINY["loss0"])
    loss.backward()
    gn = gnorm(m)
    assert abs(gn - TINY["traj"][0, 1]) <= 3e-2 * gn
    for n, p in m.named_parameters():
        assert p.grad.dtype == torch.float32 and p.dtype == torch.float32
        ref = T(TINY["g0_" + n]).double()
        err = float((p.grad.double().cpu() - ref).norm() / (ref.norm() + 1e-30))
        assert err < 6e-2, (n, err)


def test_c1_config_fp32_matches_reference():
    """BASELINE config 1: Bloom-560M 2-layer slice, B=2 S=128, full vocabulary, fp32."""
    doc = json.load(open(os.path.join(G, "c1_bloom.json")))
    c = doc["cfg"]
    m = build(c["V"], c["H"], c["L"], c["nh"])
    ids = torch.randint(0, c["V"], (c["B"], c["S"]), generator=torch.Generator().manual_seed(7))
    assert hashlib.sha256(ids.numpy().tobytes()).hexdigest() == doc["ids_sha256"]
    am = torch.ones(c["B"], c["S"], dtype=torch.long)
    am[c["pad_row"], c["pad_from"]:] = 0
    ids, am = ids.to(DEV), am.to(DEV)
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    for t in range(4):
        (loss, logits, hidden), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        if t == 0:
            assert logits.argmax(-1).cpu().tolist() == doc["argmax"]                       # bit-exact token ids
            close("c1.logits[:8]", logits[:, :, :8], torch.tensor(doc["logits_first8"]), 1e-4, 1e-5)
            close("c1.hidden[:4]", hidden[:, :, :4], torch.tensor(doc["hidden_first4"]), 1e-4, 1e-5)
            close("c1.probe", m.lm_head.weight.grad[100:110, 100:110], torch.tensor(doc["lm_head_grad_probe"]), 1e-3, 1e-12)
            for n, p in m.named_parameters():
                ref = doc["per_param_grad_norm"][n]
                assert abs(float(p.grad.double().pow(2).sum().sqrt()) - ref) <= 1e-4 * ref, n
        opt.step()
        assert abs(float(loss) - doc["traj"][t][0]) <= 1e-4 * doc["traj"][t][0], (t, float(loss), doc["traj"][t])
        assert abs(gn - doc["traj"][t][1]) <= 1e-4 * gn, (t, gn, doc["traj"][t])


def test_c1_config_bf16_loss_curve_equivalent():
    doc = json.load(open(os.path.join(G, "c1_bloom.json")))
    c = doc["cfg"]
    m = build(c["V"], c["H"], c["L"], c["nh"], compute_dtype="bf16")
    ids = torch.randint(0, c["V"], (c["B"], c["S"]), generator=torch.Generator().manual_seed(7)).to(DEV)
    am = torch.ones(c["B"], c["S"], dtype=torch.long)
    am[c["pad_row"], c["pad_from"]:] = 0
    am = am.to(DEV)
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    for t in range(4):
        (loss, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        opt.step()
        assert abs(float(loss) - doc["traj"][t][0]) <= 3e-3 * doc["traj"][t][0], (t, float(loss), doc["traj"][t])
        assert abs(gn - doc["traj"][t][1]) <= 3e-2 * gn, (t, gn, doc["traj"][t])


def _c5_inputs(doc):
    c = doc["cfg"]
    ids = torch.randint(0, c["V"], (c["B"], c["S"]), generator=torch.Generator().manual_seed(7))
    assert hashlib.sha256(ids.numpy().tobytes()).hexdigest() == doc["ids_sha256"]
    am = torch.ones(c["B"], c["S"], dtype=torch.long)
    am[c["pad_row"], c["pad_from"]:] = 0
    return c, ids.to(DEV), am.to(DEV)


def test_c5_bloom7b1_geometry_fp32_matches_reference():
    """BASELINE config 5 geometry (Bloom-7B1: H=4096, nh=32, head_dim=128, V=250880), one layer, B=1 S=512, fp32: the
    head-dim-128 attention tiles and the 4096-wide GEMM / LayerNorm / AdamW paths against values produced by the
    reference itself (tests/golden/make_golden.py c5)."""
    doc = json.load(open(os.path.join(G, "c5_bloom.json")))
    c, ids, am = _c5_inputs(doc)
    m = build(c["V"], c["H"], c["L"], c["nh"])
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    for t in range(2):
        (loss, logits, hidden), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        if t == 0:
            assert logits.argmax(-1).cpu().tolist() == doc["argmax"]                       # bit-exact token ids
            close("c5.logits", logits[:, ::16, :8], torch.tensor(doc["logits_first8"]), 1e-4, 1e-5)
            close("c5.hidden", hidden[:, ::16, :4], torch.tensor(doc["hidden_first4"]), 1e-4, 1e-5)
            close("c5.probe", m.lm_head.weight.grad[100:110, 100:110], torch.tensor(doc["lm_head_grad_probe"]), 1e-3, 1e-12)
            close("c5.qkv", m.bloom.blocks[0].self_attention.query_key_value.weight.grad[:6, :6],
                  torch.tensor(doc["qkv_grad_probe"]), 1e-3, 1e-10)
            for n, p in m.named_parameters():
                ref = doc["per_param_grad_norm"][n]
                assert abs(float(p.grad.double().pow(2).sum().sqrt()) - ref) <= 1e-4 * ref, n
        opt.step()
        assert abs(float(loss) - doc["traj"][t][0]) <= 1e-4 * doc["traj"][t][0], (t, float(loss), doc["traj"][t])
        assert abs(gn - doc["traj"][t][1]) <= 1e-4 * gn, (t, gn, doc["traj"][t])


def test_c5_bloom7b1_geometry_bf16_loss_curve_equivalent():
    doc = json.load(open(os.path.join(G, "c5_bloom.json")))
    c, ids, am = _c5_inputs(doc)
    m = build(c["V"], c["H"], c["L"], c["nh"], compute_dtype="bf16")
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    for t in range(2):
        (loss, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        opt.step()
        assert abs(float(loss) - doc["traj"][t][0]) <= 3e-3 * doc["traj"][t][0], (t, float(loss), doc["traj"][t])
        assert abs(gn - doc["traj"][t][1]) <= 3e-2 * gn, (t, gn, doc["traj"][t])


def test_oracle_vs_hip_random_config_fp32():
    """A config the goldens do not cover (odd sizes, nh not a power of two, post-LN residual switch)."""
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    V, H, L, nh, B, S = 517, 96, 3, 6, 3, 37
    for post in (False, True):
        sh = R.BloomShape(V, H, L, nh, apply_residual_connection_post_layernorm=post)
        p = R.det_init(sh)
        cfg = BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh,
                          apply_residual_connection_post_layernorm=post)
        m = BloomForCausalLM(cfg)
        m._tie_weight()
        sd = dict(p)
        sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
        m.load_state_dict(sd)
        m._tie_weight()
        m = m.to(DEV).train()
        ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(3))
        am = torch.ones(B, S, dtype=torch.long)
        am[0, 30:] = 0
        am[2, :9] = 0
        loss_ref, logits_ref, _, grads_ref = R.loss_and_grads(p, sh, ids, am)
        (loss, logits, _), _ = m(input_ids=ids.to(DEV), attention_mask=am.to(DEV), labels=ids.to(DEV).clone())
        close("rand.loss", loss, loss_ref, 1e-5)
        close("rand.logits", logits, logits_ref, 1e-4, 5e-6)
        loss.backward()
        for n, prm in m.named_parameters():
            close("rand.g." + n, prm.grad, grads_ref[n], 2e-4, 5e-7)


def test_full_size_properties_bf16():
    """BASELINE config 2 geometry per layer (H=1024, nh=16, S=1024, full vocab) with 2 layers and B=2, bf16:
    size-independent properties — causality (future tokens do not change earlier logits), gradient of the
    loss w.r.t. logits sums to zero per row (softmax - onehot), finite loss close to the fp32 oracle's C1-style value,
    and a descending loss over three optimizer steps."""
    V, H, L, nh, B, S = 250880, 1024, 2, 16, 2, 1024
    m = build(V, H, L, nh, compute_dtype="bf16")
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(11)).to(DEV)
    am = torch.ones(B, S, dtype=torch.long, device=DEV)
    with torch.no_grad():
        (lg1, _), _ = m(input_ids=ids, attention_mask=am)
        ids2 = ids.clone()
        ids2[:, 700:] = (ids2[:, 700:] + 1) % V
        (lg2, _), _ = m(input_ids=ids2, attention_mask=am)
    assert torch.equal(lg1[:, :700], lg2[:, :700])                                          # causal, bit-exact
    assert not torch.equal(lg1[:, 700:], lg2[:, 700:])
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=0.0, decoupled=True)
    losses = []
    for t in range(3):
        (loss, logits, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(math.isfinite(x) for x in losses) and losses[0] > losses[1] > losses[2], losses
    assert 12.0 < losses[0] < 20.0, losses                                                    # > ln V = 12.43


@pytest.mark.parametrize("backend", ["torch", "rccl"])
def test_ddp_wrapper_on_rccl_single_rank(backend):
    """(backend = "rccl": the same step with CTMI_DDP_BACKEND=rccl — the gradient collectives go to the library's own RCCL communicator,
    ctmi_ddp_* / ops.DirectComm, created with a channel cap; plus the raw calls: in-place all-reduce ordered behind a kernel of the
    compute stream, byte all_gather, broadcast, wait.)
    One-rank RCCL process group on the GPU: the data-parallel wrapper's collectives (bucket all-reduce from the autograd
    hooks, the tied-gradient early all-reduce, the id / row all_gather_into_tensor) run on the real backend and leave the
    gradients of the plain model unchanged (world = 1: averaging is the identity).  Multi-rank semantics are covered on CPU
    over gloo (tests/test_ddp_gloo.py); this is the check that RCCL accepts the calls, dtypes and stream usage."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", CTMI_DDP_TIED_CHUNK_ROWS="256")
os.environ["CTMI_DDP_BACKEND"] = "@BACKEND@"
os.environ["CTMI_DDP_MAX_CHANNELS"] = "8"
dist.init_process_group("nccl", rank=0, world_size=1)
if "@BACKEND@" == "rccl":
    from cleantransformer_amd import ops
    c = ops.DirectComm(ops.DirectComm.unique_id(), 0, 1, max_channels=4)
    x = torch.zeros(1 << 20, device="cuda:0")
    x.add_(3.0)                                      # a kernel on the compute stream the collective has to wait for
    c.all_reduce(x)
    y = torch.arange(1000, device="cuda:0", dtype=torch.int64); out = torch.empty(1000, device="cuda:0", dtype=torch.int64)
    c.all_gather(out, y)
    z = torch.full((4096,), 7.0, device="cuda:0", dtype=torch.bfloat16)
    c.all_reduce(z); c.broadcast(z, 0)
    c.wait()
    x.mul_(2.0)                                      # ordered behind the collective by wait()
    torch.cuda.synchronize()
    assert float(x.min()) == 6.0 and float(x.max()) == 6.0 and torch.equal(out, y) and float(z.float().min()) == 7.0
    c.close()
from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
torch.manual_seed(0)
dev = torch.device("cuda:0")
def make():
    torch.manual_seed(1)
    m = BloomForCausalLM(BloomConfig(vocab_size=1000, hidden_size=128, n_layer=2, num_attention_heads=4, compute_dtype="bf16")).to(dev)
    m._tie_weight()
    return m.train()
ids = torch.randint(0, 1000, (2, 64), device=dev)
am = torch.ones(2, 64, dtype=torch.long, device=dev)
ref = make()
(l0, _, _), _ = ref(input_ids=ids, attention_mask=am, labels=ids.clone()); l0.backward()
m = make(); ddp = DDP(m, device_ids=[0], bucket_cap_mb=0.25)
assert (ddp._direct is not None) == ("@BACKEND@" == "rccl")
# the test, not the product, makes one rank behave like one of many: the early path is taken, and the agreed capacity exceeds the local rows by 37
# (capacity != local rows: the padded id / row exchange — all_gather_into_tensor — on the real backend)
ts = ddp._tied_sync
ts._single_rank = lambda: False
_cap = ts._row_capacity
ts._row_capacity = lambda n_local: _cap(n_local) + 37
evs = ddp.record_launch_events()
for it in range(2):
    for p in m.parameters(): p.grad = None
    (l1, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone()); l1.backward()
torch.cuda.synchronize()
assert ddp._tied_sync.steps == 2, ddp._tied_sync.steps
assert sum(1 for k, _ in evs if k == "tied") == 2 * 4, evs   # V = 1000 in row pieces of 256: four dense all-reduces per step
assert len(ddp.bucket_summary()) >= 3
assert abs(float(l0) - float(l1)) < 1e-6
for (n, a), (_, b) in zip(ref.named_parameters(), m.named_parameters()):
    assert b.grad is not None and a.grad.shape == b.grad.shape, n
    tol = 1e-5 * float(a.grad.abs().max()) + 1e-12          # embedding rows: fp32 atomics, order may differ
    assert float((a.grad - b.grad).abs().max()) <= tol, (n, float((a.grad - b.grad).abs().max()), tol)
dist.destroy_process_group()
print("RCCL_DDP_OK")
'''
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code.replace("@BACKEND@", backend)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_DDP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _two_rank_worker(rank, world, port, ret):
    import torch.distributed as dist
    import sys
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    m = build(V, H, L, nh)
    if rank != 0:                                              # the wrapper must broadcast rank 0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    ddp = DDP(m, device_ids=[0], bucket_cap_mb=0.05)
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B].to(DEV)
    am = torch.ones(B, S, dtype=torch.long)
    if rank == 1:
        am[0, 11:] = 0
    am = am.to(DEV)
    ddp.train()
    os.environ["CTMI_DDP_TIED_CHUNK_ROWS"] = "64"             # V = 211: the tied dense part in four row pieces
    for it in range(2):                                        # second pass: gradients are bucket views by then
        for p in m.parameters():
            p.grad = None
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
        if it == 1:
            evs = ddp.record_launch_events()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss.backward()
        if it == 1:
            e1.record()
    torch.cuda.synchronize()
    if rank == 0:
        # where in the backward's stream of kernels each collective was launched (fraction of the backward's device time)
        span = e0.elapsed_time(e1)
        ret["launch_pos"] = [(k, e0.elapsed_time(ev) / span) for k, ev in evs]
        ret["loss0"] = float(loss)
        ret["early"] = ddp._tied_sync.steps
        ret["nbuckets"] = len(ddp.bucket_summary())
        for n, p in m.named_parameters():
            ret["g_" + n] = p.grad.float().cpu().numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_sharing_the_gpu_match_torch_ddp_golden():
    """Two processes on the one GPU of the test box, gloo as the transport (RCCL refuses two ranks on one device): the REAL
    kernels, streams and autograd hooks under the multi-rank code paths — rank-0 broadcast, bucketed averaging, the tied
    embedding / LM-head gradient's early dense all-reduce + row exchange — against gradients produced by torch-DDP around the
    reference model (tests/golden/ddp_tiny.npz, world 2)."""
    import socket
    import torch.multiprocessing as mp
    gold = np.load(os.path.join(G, "ddp_tiny.npz"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_two_rank_worker, args=(2, port, ret), nprocs=2, join=True)
    assert abs(ret["loss0"] - float(gold["w2___loss0"])) < 1e-5
    assert ret["early"] == 2 and ret["nbuckets"] >= 3
    for k in gold.files:
        if k.startswith("w2_bloom") or k.startswith("w2_lm_head"):
            name = k[len("w2_"):]
            a, b = ret["g_" + name], gold[k]
            assert a.shape == b.shape and np.allclose(a, b, rtol=1e-4, atol=2e-7), (name, float(np.abs(a - b).max()))   # fp32 GEMM summation order on the GPU
    # overlap by construction: the collectives are launched from the gradient hooks INSIDE the backward's kernel stream, not queued
    # behind it — the tied dense pieces first (the LM head is the first backward op), the buckets spread over the rest
    pos = ret["launch_pos"]
    tied = [p for k, p in pos if k == "tied"]
    buckets = [p for k, p in pos if k == "bucket"]
    assert len(tied) == 4 and len(buckets) >= 2, pos
    assert max(tied) < min(buckets) and tied == sorted(tied) and buckets == sorted(buckets), pos
    assert tied[0] < 0.5 and buckets[0] < 0.9 and buckets[-1] <= 1.0 + 1e-6, pos
    assert buckets[-1] - buckets[0] > 0.05, pos                       # not one clump at the end


class _cpu_threads:
    """The oracle's CPU kernels are fastest around 32 threads on the GPU boxes' 128-core hosts (1024 tokens of Bloom-560M: 5 s at
    32 threads, 29 s at 128, minutes at 256); the setting is process-wide, so it is restored for the tests that follow."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(max(1, min(self.n, os.cpu_count() or 1)))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def test_config1_full_size_bf16_and_fp32_vs_oracle():
    """BASELINE configs[1] at its STATED size — Bloom-560M: 24 layers, H=1024, nh=16, V=250880, B=8, S=1024 — against the fp32 CPU
    oracle evaluated at the same size on the same parameters and tokens (the oracle is pinned to the reference; about a minute on
    the GPU box's host cores).  Bars: the GPU in fp32 mode 1e-4 on loss and global gradient norm (north star); in bf16 mode
    (the measured configuration) 3e-3 / 3e-2; plus the size-independent properties at THIS size: causality of the logits
    (bit-exact), rows of dlogits sum to zero, a descending loss over three optimizer steps."""
    V, H, L, nh, B, S = 250880, 1024, 24, 16, 8, 1024
    sh = R.BloomShape(V, H, L, nh)
    p = R.det_init(sh)
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(21))
    am = torch.ones(B, S, dtype=torch.long)
    am[3, 900:] = 0                                                   # one right-padded row, like the reference's collate
    idd, amd = ids.to(DEV), am.to(DEV)
    with _cpu_threads(32):
        loss_o, logits_o, _, grads_o = R.loss_and_grads(p, sh, ids, am)
        gn_o, am_o = R.grad_norm(grads_o.values()), logits_o.argmax(-1)
    loss_o = float(loss_o)
    del logits_o
    grads_o["lm_head.weight"] = grads_o["bloom.word_embeddings.weight"]           # tied: one tensor, one gradient

    def per_parameter(m, rel_bar, what):
        """Round-4 verdict: a wrong gradient in ONE matrix of one layer is ~1e-3 of the global norm and passes a global bar.  Every parameter's
        gradient, norm-wise: ||g - g_oracle|| <= rel_bar * ||g_oracle|| + 1e-7 * (global norm) — the absolute term for parameters whose
        gradient is ~0.  Returns the worst (relative error, name)."""
        worst = (0.0, "")
        for n, q in m.named_parameters():
            ref = grads_o[n].double()
            d = float((q.grad.double().cpu() - ref).norm())
            rn = float(ref.norm())
            worst = max(worst, (d / (rn + 1e-30), n))
            assert d <= rel_bar * rn + 1e-7 * gn_o, (what, n, d / (rn + 1e-30), rn)
        return worst

    def gpu(cd):
        m = build(V, H, L, nh, compute_dtype=cd, params=p)
        (loss, logits, _), _ = m(input_ids=idd, attention_mask=amd, labels=idd.clone())
        loss.backward()
        out = (float(loss.detach()), gnorm(m), logits.argmax(-1).cpu())
        del loss, logits
        return m, out

    m32, (l32, g32, a32) = gpu("fp32")
    assert abs(l32 - loss_o) <= 1e-4 * loss_o, (l32, loss_o)
    assert abs(g32 - gn_o) <= 1e-4 * gn_o, (g32, gn_o)
    w32 = per_parameter(m32, 1e-4, "fp32")                             # the north star's 1e-4, per parameter (294 tensors)
    del m32
    assert float((a32 == am_o).float().mean()) > 0.999                # fp32 near-ties at V = 250880 may flip an argmax
    m, (loss0, gn, a16) = gpu("bf16")
    assert abs(loss0 - loss_o) <= 3e-3 * loss_o, (loss0, loss_o)
    assert abs(gn - gn_o) <= 3e-2 * gn_o, (gn, gn_o)
    w16 = per_parameter(m, 4e-2, "bf16")                               # bf16 operands through 24 layers: norm-wise per parameter (measured worst 1.7e-2, printed below)
    print(f"\nconfigs[1] per-parameter gradient error vs the oracle: fp32 worst {w32[0]:.2e} ({w32[1]}), bf16 worst {w16[0]:.2e} ({w16[1]})")
    del grads_o
    assert float((a16 == am_o).float().mean()) > 0.97                 # bf16 rounding may flip near-ties only
    # properties at this size
    with torch.no_grad():
        (lg1, _), _ = m(input_ids=idd, attention_mask=amd)
        ids2 = idd.clone()
        ids2[:, 700:] = (ids2[:, 700:] + 1) % V
        (lg2, _), _ = m(input_ids=ids2, attention_mask=amd)
    assert torch.equal(lg1[:, :700], lg2[:, :700]) and not torch.equal(lg1[:, 700:], lg2[:, 700:])
    del lg1, lg2
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=0.0, decoupled=True)
    losses = []
    for t in range(3):
        lg = None
        (loss, lg, _), _ = m(input_ids=idd, attention_mask=amd, labels=idd.clone())
        if t == 0:
            lg.retain_grad()
        opt.zero_grad()
        loss.backward()
        if t == 0:
            rows = lg.grad[:, :-1].float().sum(-1)                    # softmax - onehot sums to 0 per loss-carrying row
            assert float(rows.abs().max()) < 2e-3 / (B * (S - 1)) * 50, float(rows.abs().max())
            assert float(lg.grad[:, -1].abs().max()) == 0.0           # the shifted-out last position carries no loss
        opt.step()
        losses.append(float(loss.detach()))
    assert abs(losses[0] - loss0) <= 1e-6 * loss0                      # same parameters: same loss
    assert all(math.isfinite(x) for x in losses) and losses[0] > losses[1] > losses[2], losses


def test_config1_full_size_greedy_decode_bit_exact_vs_oracle():
    """north_star: "bit-exact for token indexing / argmax decode" — at the FULL vocabulary and depth (V = 250 880, 24 layers, fp32 parity mode):
    GenerationMixin._greedy_search (generation_util.py:57-119: KV-cached forward, argmax, append) against oracle.bloom_ref.greedy_decode, 8 new
    tokens (+ the two of the loop-exit quirk) for a ragged LEFT-padded batch.  The oracle is re-run step by step beside it to know every
    step's top-1 / top-2 margin: wherever that margin exceeds fp32 round-off (1e-4 of the logit scale) the ids must be identical; a narrower
    margin — an fp32 near-tie among 250 880 candidates, not seen with this seed — would be reported, not silently accepted."""
    V, H, L, nh = 250880, 1024, 24, 16
    sh = R.BloomShape(V, H, L, nh)
    p = R.det_init(sh)
    g = torch.Generator().manual_seed(77)
    B, S0, NEW = 2, 12, 8
    ids = torch.randint(0, V, (B, S0), generator=g)
    am = torch.ones(B, S0, dtype=torch.long)
    am[1, :5] = 0                                                       # left padding, as a batched generation call pads its prompts
    with _cpu_threads(32):
        want = R.greedy_decode(p, sh, ids, am, max_gen_len=NEW, end_ids=None, pad_id=3)
        # margins of the oracle's own decisions
        margins, cur, mask, pasts, step = [], ids.clone(), am.clone(), None, 0
        while cur.shape[1] < want.shape[-1]:
            with torch.no_grad():
                _, logits, _, pasts = R.bloom_forward(p, sh, cur[:, step:], mask, None, pasts)
            top2 = logits[:, -1, :].double().topk(2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]) / (logits[:, -1, :].double().abs().max(dim=-1).values + 1e-30))
            nxt = logits[:, -1, :].argmax(-1)
            cur = torch.cat([cur, nxt[:, None]], dim=-1)
            mask = torch.cat([mask, mask[:, -1:]], dim=-1)
            step = cur.shape[1] - 1
        assert torch.equal(cur, want.view(B, -1))
    m = build(V, H, L, nh, compute_dtype="fp32", params=p).eval()
    out = m.generate(ids.to(DEV), attention_mask=am.to(DEV),
                     generation_configs=dict(beam_size=1, max_gen_len=NEW, do_sample=False, end_ids=None, pad_id=3))
    got = out.cpu().view(B, -1)
    assert got.shape == cur.shape, (got.shape, cur.shape)
    tight = min(float(x.min()) for x in margins)
    for b in range(B):
        for t in range(S0, cur.shape[1]):
            if int(got[b, t]) != int(cur[b, t]):
                mg = float(margins[t - S0][b])
                raise AssertionError(f"greedy decode differs at batch {b}, position {t}: got {int(got[b, t])}, oracle {int(cur[b, t])}, the oracle's "
                                     f"top-2 margin there is {mg:.2e} of the logit scale ({'a near-tie' if mg < 1e-4 else 'NOT a tie'})")
    assert tight > 1e-4, f"bit-exact, but the narrowest top-2 margin of this seed ({tight:.2e}) is inside fp32 round-off: pick another seed"


def test_config4_bloom7b1_geometry_at_stated_sequence_length():
    """BASELINE configs[4] geometry at its STATED sequence length: Bloom-7B1 widths (H = 4096, 32 heads, head_dim 128, V = 250880),
    S = 2048, bf16 — 2 layers and B = 1 so that the test fits a few seconds and the fp32 master state fits next to the other tests (the
    full-depth 8-GPU run is the driver's).  Properties that do not depend on depth: causality of the logits (bit-exact: the hd = 128
    attention tiles, S = 2048 = 8 query blocks of the 256-row forward), rows of dlogits sum to zero, a descending loss over three
    optimizer steps; and the loss of a ONE-layer slice against the fp32 CPU oracle (pinned to the reference) on the same parameters
    and tokens within the bf16 bar of 3e-3."""
    V, H, nh, B, S = 250880, 4096, 32, 1, 2048
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(41))
    am = torch.ones(B, S, dtype=torch.long)
    am[0, 1900:] = 0                                                    # right padding, like the reference's collate
    idd, amd = ids.to(DEV), am.to(DEV)
    # ---- one layer vs the oracle
    sh1 = R.BloomShape(V, H, 1, nh)
    p1 = R.det_init(sh1)
    with _cpu_threads(32), torch.no_grad():
        loss_o = float(R.bloom_forward(p1, sh1, ids, am, labels=ids.clone())[0])
    m1 = build(V, H, 1, nh, compute_dtype="bf16", params=p1)
    with torch.no_grad():
        (l1, _, _), _ = m1(input_ids=idd, attention_mask=amd, labels=idd.clone())
    assert abs(float(l1) - loss_o) <= 3e-3 * loss_o, (float(l1), loss_o)
    del m1, p1
    torch.cuda.empty_cache()
    # ---- two layers: size-independent properties
    m = build(V, H, 2, nh, compute_dtype="bf16")
    with torch.no_grad():
        (lg1, _), _ = m(input_ids=idd, attention_mask=amd)
        ids2 = idd.clone()
        ids2[:, 1300:] = (ids2[:, 1300:] + 1) % V
        (lg2, _), _ = m(input_ids=ids2, attention_mask=amd)
    assert torch.equal(lg1[:, :1300], lg2[:, :1300]) and not torch.equal(lg1[:, 1300:], lg2[:, 1300:])
    del lg1, lg2
    from cleantransformer_amd.optimizer import AdamW
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=0.0, decoupled=True)
    losses = []
    for t in range(3):
        lg = None
        (loss, lg, _), _ = m(input_ids=idd, attention_mask=amd, labels=idd.clone())
        if t == 0:
            lg.retain_grad()
        opt.zero_grad()
        loss.backward()
        if t == 0:
            rows = lg.grad[:, :-1].float().sum(-1)
            assert float(rows.abs().max()) < 2e-3 / (B * (S - 1)) * 50, float(rows.abs().max())
            assert float(lg.grad[:, -1].abs().max()) == 0.0
        opt.step()
        losses.append(float(loss.detach()))
    assert all(math.isfinite(x) for x in losses) and losses[0] > losses[1] > losses[2], losses
    assert losses[0] > 12.0, losses                                    # > ln V = 12.43 (det_init at H = 4096 starts near 23)


def test_bf16_loss_curve_tracks_fp32_200_steps():
    """"Loss-curve equivalent" (north star; the loop of examples/ft_bloom.py:84-95): the C1 geometry (Bloom-560M widths, 2 layers,
    B=2, S=128, full vocabulary) trained for 200 optimizer steps at lr 1e-4 on a rotating set of 8 batches, once in fp32 and once
    in bf16 compute, both on the GPU.  The fp32 path is pinned to the reference at this geometry (test_c1_config_fp32_*).
    The loss falls from 16.4 to below 0.1 (the 8 batches are memorised), steepest around step 120.  Band at EVERY step:
    |loss_bf16 - loss_fp32| <= max(3 % of the fp32 loss, 0.3 % of the initial loss) — measured: <= 0.03 absolute everywhere, i.e.
    the curves are indistinguishable on the scale of the run — and <= 1.5 % on the 20-step moving average."""
    from cleantransformer_amd.optimizer import AdamW
    V, H, L, nh, B, S = 250880, 1024, 2, 16, 2, 128
    g = torch.Generator().manual_seed(123)
    batches = [torch.randint(0, V, (B, S), generator=g).to(DEV) for _ in range(8)]
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 100:] = 0
    am = am.to(DEV)
    curves = {}
    for cd in ("fp32", "bf16"):
        m = build(V, H, L, nh, compute_dtype=cd)
        opt = AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, decoupled=True)
        losses = []
        for t in range(200):
            ids = batches[t % len(batches)]
            (loss, _, _), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        curves[cd] = torch.stack(losses).float().cpu()
        del m, opt
    a, b = curves["fp32"].double(), curves["bf16"].double()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert float(a[-8:].mean()) < 0.1 * float(a[:8].mean()), (float(a[:8].mean()), float(a[-8:].mean()))    # it really trains
    band = torch.maximum(0.03 * a, torch.full_like(a, 0.003 * float(a[0])))
    over = (b - a).abs() - band
    assert float(over.max()) <= 0.0, (int(over.argmax()), float(a[int(over.argmax())]), float(b[int(over.argmax())]))
    k = torch.ones(20, dtype=torch.float64) / 20
    ma = torch.nn.functional.conv1d(a.view(1, 1, -1), k.view(1, 1, -1)).view(-1)
    mb = torch.nn.functional.conv1d(b.view(1, 1, -1), k.view(1, 1, -1)).view(-1)
    assert float(((mb - ma).abs() / ma).max()) <= 1.5e-2


def _bf16_uneven_worker(rank, world, port, ret):
    import copy
    import torch.distributed as dist
    import sys
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    V, H, L, nh, B = 211, 64, 2, 8, 2
    seqs = [16, 11]
    m = build(V, H, L, nh)
    plain = copy.deepcopy(m)
    plain._tie_weight()
    ddp = DDP(m, device_ids=[0], bucket_cap_mb=0.05, comm_dtype=torch.bfloat16).train()
    batches = [torch.randint(0, V, (B, s), generator=torch.Generator().manual_seed(40 + r)).to(DEV) for r, s in enumerate(seqs)]
    want = None
    for ids in batches:
        for p in plain.parameters():
            p.grad = None
        (l, _, _), _ = plain(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())
        l.backward()
        g = [p.grad.clone() / world for p in plain.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    ids = batches[rank]
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())
        loss.backward()
    torch.cuda.synchronize()
    worst_tied, worst_rest = 0.0, 0.0
    for (n, p), w in zip(m.named_parameters(), want):
        err = float((p.grad - w).abs().max()) / (float(w.abs().max()) + 1e-30)
        if "word_embeddings.weight" in n:
            worst_tied = max(worst_tied, err)
        else:
            worst_rest = max(worst_rest, err)
    if rank == 0:
        ret["tied"], ret["rest"], ret["early"] = worst_tied, worst_rest, ddp._tied_sync.steps
        ret["wire"] = str(ddp._buckets[0].comm.dtype)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_bf16_buckets_and_uneven_sequence_lengths_on_the_gpu():
    """SURVEY §8(f)2 "bf16 grads/buckets" ON THE DEVICE (round-1 verdict: it had only run on CPU over gloo): two ranks sharing the
    GPU, comm_dtype=torch.bfloat16 — the bucket cast kernels (fp32 -> bf16 wire copy -> fp32), the averaged gradients within
    bf16 rounding of the mean of the local gradients — with DIFFERENT sequence lengths per rank (the reference's collate pads per
    rank), so the tied [V,H] gradient's row exchange pads to the agreed capacity; that gradient keeps its fp32 early path."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_bf16_uneven_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["wire"] == "torch.bfloat16" and ret["early"] == 2
    assert ret["tied"] <= 1e-5, ret["tied"]                              # fp32 path (dense all-reduce + row exchange)
    assert 0.0 < ret["rest"] <= 2.0 ** -7, ret["rest"]                   # bf16 wire: rounded, and not by accident exact


def test_bench_two_rank_code_path_executes():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on this one-GPU box: both ranks
    share cuda:0 and use gloo, the model is shrunk — all of it patched in from outside by tests/bench_plumbing.py, bench.py has no such
    switches — what is checked is that the N>1 code path of bench.py
    runs end to end (DDP wrap, launch policy switch, barrier / max-over-ranks timing, one JSON line from rank 0), not a number."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join("tests", "bench_plumbing.py"), "2,4096", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--seq", "256", "--comm-dtype", "bf16"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 2 and doc["scaling"] == "weak" and doc["value"] > 0
    assert doc["metric"].startswith("PLUMBING RUN") and doc["config"]["parallelism"] == "dp2" and doc["config"]["comm_dtype"] == "bf16"
    assert doc["roofline"]["breakdown_ms_per_step"]["ms"]["gemm_fwd"] > 0 and doc["config"]["padded_sample"]["ms_per_step"] > 0
    assert math.isfinite(doc["final_loss"])
    # round 6: the collectives' exposed time is part of the line at world > 1 (steps with every gradient collective skipped, same policy)
    comm = doc["config"]["comm"]
    assert comm["ranks"] == 2 and comm["exposed"]["compute_only_ms_per_step"] > 0 and comm["exposed"]["backend"] == "gloo"
    assert abs(comm["exposed_ms"] - (doc["ms_per_step"] - comm["exposed"]["compute_only_ms_per_step"])) < 2e-3
    assert isinstance(comm["candidates"], list) and len(comm["candidates"]) >= 3
