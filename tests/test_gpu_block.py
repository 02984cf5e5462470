"""GPU (-m gpu): the block-level entry points of the C ABI (round 2) — ctmi_bloom_block_fwd / ctmi_bloom_block_bwd,
ctmi_reduce_jobs, ctmi_ce_fwd_bwd / ctmi_scale_if — against (a) the CPU oracle and (b) the same computation composed from
the individually verified per-op kernels (tests/cpu_kernel_emulation.py's block sequences run with K = ops).

Tolerances: fp32 <= 1e-4 relative (north star); results that come out of the SAME kernels in the same order (activations,
dx, weight gradients, LayerNorm affine gradients) must be bit-identical between the one-call and the per-op form; the bias
gradients that moved into the LayerNorm backward are summed in a different order and are compared with a summation-sized
tolerance."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bloom_ref as R  # noqa: E402
import cpu_kernel_emulation as EMU  # noqa: E402  (tests/ is on sys.path under pytest's rootdir/conftest handling)

DEV = "cuda:0"


def ops():
    from cleantransformer_amd import ops as o
    return o


def lib():
    from cleantransformer_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ reduce jobs
def test_reduce_jobs_matches_ordered_sum():
    o, L = ops(), lib()
    specs = [(1024, 37, 4 * 1024, 0), (3072, 128, 3072, 1), (70, 5, 70, 0), (4096, 512, 2 * 4096, 0), (1, 1, 1, 1)]
    jobs = (L.ReduceJob * len(specs))()
    keep, want = [], []
    for i, (n, parts, stride, acc) in enumerate(specs):
        src = rnd(parts * stride + 8, seed=10 + i).to(DEV)
        dst = rnd(n, seed=50 + i).to(DEV)
        alpha = 1.0 if i % 2 == 0 else 0.5
        ref = alpha * src[: parts * stride].view(parts, stride)[:, :n].double().sum(0)
        if acc:
            ref = ref + dst.double()
        want.append(ref)
        keep.append((src, dst))
        jobs[i].src, jobs[i].dst, jobs[i].n, jobs[i].part_stride = src.data_ptr(), dst.data_ptr(), n, stride
        jobs[i].nparts, jobs[i].accumulate, jobs[i].alpha = parts, acc, alpha
    L.check(L.load().ctmi_reduce_jobs(jobs, len(specs), o._stream()), "reduce_jobs")
    torch.cuda.synchronize()
    for (src, dst), ref in zip(keep, want):
        assert relerr(dst, ref) < 2e-6


def test_reduce_jobs_more_than_one_launch_worth():
    o, L = ops(), lib()
    n_jobs = 37                                                   # > CTMI_REDUCE_MAX_JOBS (16): split over launches
    jobs = (L.ReduceJob * n_jobs)()
    keep = []
    for i in range(n_jobs):
        src = rnd(7 * 130, seed=i).to(DEV)
        dst = torch.zeros(90 + i, device=DEV)
        keep.append((src, dst))
        jobs[i].src, jobs[i].dst, jobs[i].n, jobs[i].part_stride = src.data_ptr(), dst.data_ptr(), 90 + i, 130
        jobs[i].nparts, jobs[i].accumulate, jobs[i].alpha = 7, 0, 1.0
    L.check(L.load().ctmi_reduce_jobs(jobs, n_jobs, o._stream()), "reduce_jobs")
    torch.cuda.synchronize()
    for i, (src, dst) in enumerate(keep):
        assert relerr(dst, src.view(7, 130)[:, : 90 + i].double().sum(0)) < 2e-6


# ------------------------------------------------------------------------------------------------ fused loss + gradient
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("N,Cn,seq", [(32, 1000, 8), (12, 250880, 6), (64, 4104, 64), (5, 8, 5)])
def test_ce_fwd_bwd_equals_two_pass_and_oracle(dtype, rtol, N, Cn, seq):
    o = ops()
    x = rnd(N, Cn, seed=3, scale=2.0)
    lab = torch.randint(0, Cn, (N,), generator=torch.Generator().manual_seed(4))
    lab[1] = -100
    xd = x.to(DEV).to(dtype)
    lo, lse, dl = o.ce_fwd_bwd(xd, lab.to(DEV), seq=seq, shift=1)
    lo2, lse2 = o.ce_fwd(xd, lab.to(DEV), seq=seq, shift=1)
    dl2 = o.ce_bwd(xd, lab.to(DEV), lse2, lo2, None, seq=seq, shift=1)
    torch.cuda.synchronize()
    # oracle on the values the kernel saw
    xs = xd.float().cpu()
    tgt = EMU._targets(lab, N, seq, 1, -100)
    live = tgt >= 0
    lse_ref = torch.logsumexp(xs.double(), -1)
    denom = float(live.sum())
    loss_ref = float((lse_ref[live] - xs.double()[live].gather(1, tgt[live][:, None])[:, 0]).sum() / denom)
    assert abs(float(lo[0]) - loss_ref) <= 2e-5 * abs(loss_ref)
    assert abs(float(lo[0]) - float(lo2[0])) <= 1e-6 * abs(loss_ref)
    assert abs(float(lo[1]) - 1.0 / denom) <= 1e-7 / denom
    assert relerr(lse, lse_ref) < 1e-6 and relerr(lse, lse2) < 1e-6
    p = torch.exp(xs.double() - lse_ref[:, None])
    p[torch.arange(N)[live], tgt[live]] -= 1.0
    d_ref = torch.where(live[:, None], p / denom, torch.zeros((), dtype=torch.float64))
    assert relerr(dl, d_ref) < rtol
    assert relerr(dl, dl2) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert float(dl.float()[~live.to(DEV)].abs().sum()) == 0.0                   # rows without a target: exact zeros
    # upstream gradient: 1 -> untouched (bit-exact), k -> scaled
    keep = dl.clone()
    o.scale_if_(dl, torch.ones(1, device=DEV))
    assert torch.equal(dl, keep)
    o.scale_if_(dl, torch.full((1,), 0.25, device=DEV))
    assert torch.equal(dl.float(), keep.float() * 0.25)                         # a power of two: exact in both dtypes


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_fused_loss_at_a_padded_row_pitch_and_the_lm_head_backward_finds_its_buffer(dtype, rtol):
    """Round 6: an odd class count (GPT-2: 50257) keeps its logits in rows of ops.pad_rows(V); the single-pass loss takes that pitch (it did not
    until round 6: such steps ran the two-pass form), gives dlogits the same pitch with zero pad columns, equals the two-pass form — and the loss
    node's BACKWARD (autograd's worker thread, where the per-thread ZERO_PADDED table is read) registers the padded buffer, so that the LM-head
    backward runs over the padded extent instead of re-packing 0.8 GB: a forward-thread registration cost GPT-2-medium 10 ms per step."""
    from cleantransformer_amd.models.modeling_bloom import ShiftedCrossEntropyFn
    o = ops()
    B, S, V = 2, 16, 1001
    Vp = o.pad_rows(V)
    buf = rnd(B * S, Vp, seed=5, scale=2.0).to(DEV).to(dtype)
    l2 = buf[:, :V]
    lab = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(6)).to(DEV)
    assert o.ce_fused_ok(l2) and l2.stride(0) == Vp != V
    lo, lse, dl = o.ce_fwd_bwd(l2, lab, seq=S, shift=1)
    lo2, lse2 = o.ce_fwd(l2, lab, seq=S, shift=1)
    dl2 = o.ce_bwd(l2, lab, lse2, lo2, None, seq=S, shift=1)
    assert dl.stride(0) == Vp and dl._base is not None and float(dl._base[:, V:].float().abs().sum()) == 0.0
    assert abs(float(lo[0]) - float(lo2[0])) <= 1e-6 * abs(float(lo2[0])) and relerr(lse, lse2) < 1e-6
    assert relerr(dl, dl2) < (1e-6 if dtype == torch.float32 else 4e-3)
    ref = l2.detach().float().cpu().double().reshape(B, S, V)
    l_ref = torch.nn.functional.cross_entropy(ref[:, :-1].reshape(-1, V), lab.cpu()[:, 1:].reshape(-1))
    assert abs(float(lo[0]) - float(l_ref)) <= (2e-5 if dtype == torch.float32 else 2e-3) * abs(float(l_ref))
    # through the autograd node: the gradient arrives as a VIEW of the padded buffer and is registered on the backward thread
    x = l2.reshape(B, S, V).detach().requires_grad_(True)                    # (reshape of the padded view keeps the pitch: no copy)
    assert x.stride(1) == Vp
    seen = {}
    x.register_hook(lambda g: seen.setdefault("entry", o.ZERO_PADDED.get(g.reshape(B * S, V).data_ptr())) and None)
    loss = ShiftedCrossEntropyFn.apply(x, lab)
    loss.backward()
    assert seen["entry"] is not None and seen["entry"][0] == Vp, "the loss node's backward did not register its padded dlogits"
    assert relerr(x.grad, dl.reshape(B, S, V)) < rtol


def test_fused_loss_node_autograd_scaling_and_double_backward_guard():
    from cleantransformer_amd.models.modeling_bloom import ShiftedCrossEntropyFn
    B, S, V = 2, 8, 512
    x = rnd(B, S, V, seed=9).to(DEV).requires_grad_(True)
    lab = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(DEV)
    loss = ShiftedCrossEntropyFn.apply(x, lab)
    (loss * 4.0).backward()
    ref = x.detach().cpu().double().requires_grad_(True)
    l_ref = torch.nn.functional.cross_entropy(ref[:, :-1].reshape(-1, V), lab.cpu()[:, 1:].reshape(-1))
    (l_ref * 4.0).backward()
    assert abs(float(loss) - float(l_ref)) < 1e-5
    assert relerr(x.grad, ref.grad) < 1e-5
    x2 = x.detach().clone().requires_grad_(True)
    loss2 = ShiftedCrossEntropyFn.apply(x2, lab)
    loss2.backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        loss2.backward()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_loss_folds_the_expected_upstream_gradient(dtype):
    """trainer.py:468-504 backpropagates `loss / ga`, ft_bloom_DDP.py:123-127 `scaler.scale(loss)`: when the loop announces that factor
    before the forward (ops.expected_loss_grad / ops.set_expected_loss_grad) the fused loss writes dlogits for it in its one pass and the
    backward does NOT rescale — no second pass over [T,V] (ctmi_scale_if_passes does not move) — and bf16 dlogits equal the fp32 gradient
    times the factor rounded ONCE.  A gradient that differs from the announcement is still applied correctly (one rescale pass)."""
    from cleantransformer_amd.models.modeling_bloom import ShiftedCrossEntropyFn
    o = ops()
    B, S, V = 2, 16, 1024
    x0 = rnd(B, S, V, seed=19).to(DEV).to(dtype)
    lab = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(2)).to(DEV)
    ref = x0.detach().cpu().double().requires_grad_(True)
    torch.nn.functional.cross_entropy(ref[:, :-1].reshape(-1, V), lab.cpu()[:, 1:].reshape(-1)).backward()
    g64 = ref.grad                                                                    # d loss / d logits for an upstream gradient of 1
    scale_dev = torch.full((1,), 1024.0, device=DEV)
    for fac, sdev, upstream, rescales in ((0.25, None, 0.25, 0), (1.0, scale_dev, 1024.0, 0), (0.5, scale_dev, 512.0, 0), (0.25, None, 1.0, 1)):
        o.set_expected_loss_grad(factor=1.0, scale=False)
        if sdev is not None:
            o.set_expected_loss_grad(scale=sdev)
        before = o.scale_if_passes()
        x = x0.clone().requires_grad_(True)
        with o.expected_loss_grad(fac):
            loss = ShiftedCrossEntropyFn.apply(x, lab)
        (loss * upstream).backward()
        torch.cuda.synchronize()
        assert o.scale_if_passes() - before == rescales, (fac, sdev is not None, upstream)
        want = g64 * upstream
        if dtype == torch.float32:
            assert relerr(x.grad, want) < 2e-6
        elif rescales == 0:
            # one rounding of the fp32 product: the kernel's fp32 softmax differs from fp64 by ~1e-7 relative, so allow the neighbouring
            # bf16 value where the exact product sits within that of a rounding boundary
            got, w32 = x.grad.float().cpu(), want.float()
            exact = w32.to(torch.bfloat16).float()
            ulp = (exact.abs() * 2.0 ** -7).clamp_min(1e-30)
            assert float(((got - exact).abs() / ulp).max()) <= 1.0 + 1e-3
            assert float((got != exact).float().mean()) < 2e-3, float((got != exact).float().mean())
        else:
            assert relerr(x.grad, want) < 8e-3
    o.set_expected_loss_grad(factor=1.0, scale=False)


# ------------------------------------------------------------------------------------------------ one block per call
def _block_inputs(B, S, H, nh, dtype, seed, pad):
    o = ops()
    T = B * S
    g = torch.Generator().manual_seed(seed)

    def r(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale
    names = lib().BLK_PARAMS
    shapes = {"ln1_w": (H,), "ln1_b": (H,), "wqkv": (3 * H, H), "bqkv": (3 * H,), "wd": (H, H), "bd": (H,), "ln2_w": (H,), "ln2_b": (H,),
              "w1": (4 * H, H), "b1": (4 * H,), "w2": (H, 4 * H), "b2": (H,)}
    master = {}
    for n in names:
        shp = shapes[n]
        master[n] = (1.0 + 0.1 * r(*shp)) if n in ("ln1_w", "ln2_w") else r(*shp, scale=0.05 if len(shp) == 2 else 0.02)
    x = r(T, H)
    dout = r(T, H, scale=0.1)
    am = torch.ones(B, S, dtype=torch.long)
    if pad == "right":
        am[0, S - S // 3:] = 0
    elif pad == "left":
        am[B - 1, : S // 4] = 0
    params = tuple(master[n].to(DEV).to(dtype if master[n].dim() == 2 else torch.float32).contiguous() for n in names)
    mask = o.MaskInfo(am.to(DEV))
    from cleantransformer_amd.models.modeling_bloom import alibi_slopes
    slopes = alibi_slopes(nh).to(DEV)
    return x.to(DEV).to(dtype), dout.to(DEV).to(dtype), params, mask, slopes, master, am


GEOS = [(2, 16, 64, 8), (2, 96, 256, 4), (1, 200, 1024, 16), (2, 256, 1024, 16)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("post", [False, True])
@pytest.mark.parametrize("B,S,H,nh", GEOS)
def test_block_one_call_equals_per_op_sequence(dtype, post, B, S, H, nh):
    o = ops()
    pad = "right" if S % 32 == 0 else "left"
    x, dout, params, mask, slopes, _, _ = _block_inputs(B, S, H, nh, dtype, seed=B * 1000 + S + H, pad=pad)
    eps = 1e-5
    acts = o.bloom_block_fwd(x, params, mask, slopes, eps, post, B, S, nh)
    ref = EMU.bloom_block_fwd(x, params, mask, slopes, eps, post, B, S, nh, K=o)
    T = B * S
    for slot, cols in (("ln1", H), ("qkv", 3 * H), ("att", H), ("h1", H), ("ln2", H), ("u", 4 * H), ("g", 4 * H), ("out", H)):
        assert torch.equal(acts.view(slot, T, cols), getattr(ref, slot)), slot
    for slot in ("mean1", "rstd1", "mean2", "rstd2"):
        assert torch.equal(acts.view(slot, 1, T, torch.float32).view(-1), getattr(ref, slot)), slot
    for slot in ("stat_m", "stat_l"):
        assert torch.equal(acts.view(slot, 1, B * nh * S, torch.float32).view(B, nh, S), getattr(ref, slot)), slot

    results = {}
    for side in (False, True):
        dx, grads = o.bloom_block_bwd(acts, x, params, mask, slopes, eps, post, dout, use_side_stream=side)
        torch.cuda.synchronize()
        results[side] = (dx, grads)
    dx_ref, g_ref = EMU.bloom_block_bwd(ref, x, params, mask, slopes, eps, post, dout, K=o)
    torch.cuda.synchronize()
    names = lib().BLK_PARAMS
    # dw2 / dw1 / db1 come out of the same kernels on the same inputs: bit-identical.  Everything downstream of the second
    # LayerNorm's backward goes through a different instantiation of that kernel in the one-call form (the one that also emits
    # the bias column sums): same formula, but hipcc's fp contraction may differ by an ulp, so those are compared to rounding.
    rt = 2e-5 if dtype == torch.float32 else 2e-2
    # round 5: bf16 blocks whose geometry fits the grouped weight-gradient launch (csrc/gemm.hip ctmi_wgrad_grouped) take it in the one-call form
    grouped = dtype != torch.float32 and H % 256 == 0 and T % 32 == 0 and T >= 64 and os.environ.get("CTMI_WGRAD_GROUP", "1") != "0"
    for side in (False, True):
        dx, grads = results[side]
        assert relerr(dx, dx_ref) < rt, f"dx side={side}"
        for n, g, gr in zip(names, grads, g_ref):
            if n in ("w2", "w1", "b1") and not grouped:
                assert torch.equal(g, gr), (n, side)
            elif n in ("w2", "w1"):
                assert relerr(g, gr) < 5e-6, (n, side)                          # grouped launch: same products, other fp32 summation order
            elif n in ("bd", "b2", "bqkv", "b1"):
                # column sums: bd / b2 come out of the LayerNorm backward (other summation order)
                scale = float(gr.abs().max()) + 1e-30
                assert float((g - gr).abs().max()) <= (2e-5 if dtype == torch.float32 else 2e-2) * scale * math.sqrt(T), (n, side)
            else:
                assert relerr(g, gr) < rt, (n, side)
    assert torch.equal(results[False][0], results[True][0])
    for a, b in zip(results[False][1], results[True][1]):
        assert torch.equal(a, b)                                                # one stream or two: same bits


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,H,nh", [(2, 16, 64, 8), (2, 96, 256, 4), (2, 256, 1024, 16)])
def test_block_gpt2_spelling_with_in_out_weights(dtype, B, S, H, nh):
    """The GPT-2 flags of the block-level call (blocked q | k | v, -1e4 fill, [in,out] weight gradients) with the weights passed in
    Conv1D's own [in,out] layout (CTMI_BLK_W_IN_OUT: forward GEMMs read them K-major, data gradients as the row-major operand)
    against (a) the same call on [out,in] copies of the same values and (b) the per-op kernel sequence."""
    o, L = ops(), lib()
    x, dout, params, mask, _, _, _ = _block_inputs(B, S, H, nh, dtype, seed=4242 + S + H, pad="left")
    eps, scale = 1e-5, 1.0 / math.sqrt(H // nh)
    names = L.BLK_PARAMS
    p_io = tuple(p.t().contiguous() if n in ("wqkv", "wd", "w1", "w2") else p for n, p in zip(names, params))
    f_oi, f_io = L.BLK_QKV_BLOCKED | L.BLK_WGRAD_IN_OUT, L.BLK_QKV_BLOCKED | L.BLK_WGRAD_IN_OUT | L.BLK_W_IN_OUT
    a_oi = o.bloom_block_fwd(x, params, mask, None, eps, False, B, S, nh, flags=f_oi, attn_scale=scale, future_fill=-1e4)
    a_io = o.bloom_block_fwd(x, p_io, mask, None, eps, False, B, S, nh, flags=f_io, attn_scale=scale, future_fill=-1e4)
    ref = EMU.bloom_block_fwd(x, p_io, mask, None, eps, False, B, S, nh, flags=f_io, attn_scale=scale, future_fill=-1e4, K=o)
    T = B * S
    rt = 2e-5 if dtype == torch.float32 else 2e-2
    for slot, cols in (("qkv", 3 * H), ("att", H), ("h1", H), ("u", 4 * H), ("g", 4 * H), ("out", H)):
        assert torch.equal(a_io.view(slot, T, cols), getattr(ref, slot)), slot                 # same kernels, one call or many
        assert relerr(a_io.view(slot, T, cols), a_oi.view(slot, T, cols)) < rt, slot           # other operand layout, same math
    dx_oi, g_oi = o.bloom_block_bwd(a_oi, x, params, mask, None, eps, False, dout)
    dx_io, g_io = o.bloom_block_bwd(a_io, x, p_io, mask, None, eps, False, dout)
    dx_ref, g_ref = EMU.bloom_block_bwd(ref, x, p_io, mask, None, eps, False, dout, K=o)
    torch.cuda.synchronize()
    assert relerr(dx_io, dx_oi) < rt and relerr(dx_io, dx_ref) < rt
    for n, a, b, c in zip(names, g_io, g_oi, g_ref):
        assert a.shape == b.shape == c.shape, n
        if n in ("bd", "b2", "bqkv"):
            sc = float(c.abs().max()) + 1e-30
            assert float((a - c).abs().max()) <= rt * sc * math.sqrt(T) and float((a - b).abs().max()) <= rt * sc * math.sqrt(T), n
        else:
            assert relerr(a, b) < rt and relerr(a, c) < rt, n
        if n in ("wqkv", "wd", "w1", "w2"):
            assert a.shape == p_io[names.index(n)].shape, n                                      # [in, out], the parameter's layout
    # [in,out] weights with [out,in] gradients (the fourth flag combination): the same numbers, transposed
    a_x = o.bloom_block_fwd(x, p_io, mask, None, eps, False, B, S, nh, flags=L.BLK_QKV_BLOCKED | L.BLK_W_IN_OUT, attn_scale=scale, future_fill=-1e4)
    dx_x, g_x = o.bloom_block_bwd(a_x, x, p_io, mask, None, eps, False, dout)
    torch.cuda.synchronize()
    assert torch.equal(a_x.out, a_io.out) and relerr(dx_x, dx_io) < rt
    for n, a, b in zip(names, g_x, g_io):
        if n in ("wqkv", "wd", "w1", "w2"):
            assert a.shape == tuple(reversed(b.shape)) and relerr(a, b.t()) < rt, n


@pytest.mark.parametrize("post", [False, True])
def test_block_fp32_matches_oracle(post):
    """fp32: the one-call block against the CPU oracle's block (north-star bar 1e-4; achieved ~1e-6)."""
    o = ops()
    B, S, H, nh = 2, 48, 128, 4
    x, dout, params, mask, slopes, master, am = _block_inputs(B, S, H, nh, torch.float32, seed=77, pad="right")
    acts = o.bloom_block_fwd(x, params, mask, slopes, 1e-5, post, B, S, nh)
    dx, grads = o.bloom_block_bwd(acts, x, params, mask, slopes, 1e-5, post, dout)
    torch.cuda.synchronize()
    # oracle: one block as an autograd graph over the same fp32 values
    sh = R.BloomShape(11, H, 1, nh, apply_residual_connection_post_layernorm=post)
    names = lib().BLK_PARAMS
    ref_names = {"ln1_w": "input_layernorm.weight", "ln1_b": "input_layernorm.bias", "wqkv": "self_attention.query_key_value.weight",
                 "bqkv": "self_attention.query_key_value.bias", "wd": "self_attention.dense.weight", "bd": "self_attention.dense.bias",
                 "ln2_w": "post_attention_layernorm.weight", "ln2_b": "post_attention_layernorm.bias", "w1": "mlp.dense_h_to_4h.weight",
                 "b1": "mlp.dense_h_to_4h.bias", "w2": "mlp.dense_4h_to_h.weight", "b2": "mlp.dense_4h_to_h.bias"}
    p = {"bloom.blocks.0." + ref_names[n]: master[n].clone().requires_grad_(True) for n in names}
    xr = x.cpu().view(B, S, H).clone().requires_grad_(True)
    out_ref, _ = R.bloom_block(p, 0, xr, R.build_alibi(am, nh), R.causal_key_mask(am, S), sh)
    out_ref.backward(dout.cpu().view(B, S, H))
    assert relerr(acts.out, out_ref.reshape(B * S, H)) < 1e-5
    assert relerr(dx, xr.grad.reshape(B * S, H)) < 1e-4
    for n, g in zip(names, grads):
        assert relerr(g, p["bloom.blocks.0." + ref_names[n]].grad) < 1e-4, n
