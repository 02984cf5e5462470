"""GPU (-m gpu): every ctmi355 kernel against the CPU oracle on the same seeded inputs, called through the C ABI.

Tolerances (written here, per the north star): fp32 kernels <= 1e-4 relative (most are ~1e-6); bf16 kernels are
compared with the fp32 oracle evaluated on the bf16-rounded inputs, with a bf16-sized tolerance (2^-8 relative per
rounding, a few roundings per op).  Integer results (token ids, argmax, masks) are bit-exact.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bloom_ref as R  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def ops():
    from cleantransformer_amd import ops as o
    return o


def lib():
    from cleantransformer_amd import _lib
    return _lib


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def check(name, got, ref, rtol, atol=0.0):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} off; worst abs {float(err.max()):.3e} "
                             f"rel-norm {rel_err(got, ref):.3e}; first bad idx {idx} got {float(got[tuple(idx)])} ref {float(ref[tuple(idx)])}")


def check_norm(name, got, ref, rel, row_rel, row_abs_frac=1e-2):
    """Norm-wise bounds for gradient tensors (round-3 verdict: a 10 % element tolerance lets a wrong scale on a minority of elements
    through): the global relative error ||got - ref|| / ||ref|| <= rel AND, for every row of the last dimension,
    ||d_row|| <= row_rel * ||ref_row|| + row_abs_frac * rms_r ||ref_r||  (the absolute term covers rows whose gradient is ~0)."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    glob = float((g2 - r2).norm() / (r2.norm() + 1e-300))
    rn, dn = r2.norm(dim=1), (g2 - r2).norm(dim=1)
    scale = float(rn.pow(2).mean().sqrt())
    excess = dn - (row_rel * rn + row_abs_frac * scale)
    worst = int(excess.argmax())
    if os.environ.get("CTMI_TEST_VERBOSE"):
        print(f"[check_norm] {name}: global {glob:.3e} (bound {rel:.1e}); worst row {worst}: |d| {float(dn[worst]):.3e} |ref| {float(rn[worst]):.3e} "
              f"rms|ref| {scale:.3e}; max row rel (rows >= rms/10) {float((dn / rn.clamp_min(1e-300))[rn > scale / 10].max()):.3e}")
    assert glob <= rel, f"{name}: relative norm error {glob:.3e} > {rel:.1e}"
    assert float(excess[worst]) <= 0, (f"{name}: row {worst}: |got - ref| = {float(dn[worst]):.3e} > {row_rel:.1e} * {float(rn[worst]):.3e} + "
                                       f"{row_abs_frac:.0e} * {scale:.3e}")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def to_dev(t, dtype):
    return t.to(DEV).to(dtype).contiguous()


def bf(t):
    """value of t after rounding to bf16 (what the bf16 kernels see)."""
    return t.to(torch.bfloat16).float()


DT = [(torch.float32, 1e-5, 1e-6), (torch.bfloat16, 2e-2, 2e-2), (torch.float16, 4e-3, 4e-3)]     # fp16 (round 5): 11 significant bits against bf16's 8


def lo(dtype):
    """value after rounding to the 16-bit compute dtype `dtype` (what those kernels see); identity for fp32"""
    return (lambda t: t) if dtype == torch.float32 else (lambda t: t.to(dtype).float())


# ------------------------------------------------------------------------------------------------ probes (diagnostics)
def test_probe_mfma_layouts():
    """The MFMA operand/accumulator lane layouts assumed in csrc/mma.h (asymmetric operands catch transposes)."""
    o, L = ops(), lib()
    import ctypes as C
    for which, kdim in ((1, 32), (2, 4)):
        A = torch.arange(16 * kdim, dtype=torch.float32).reshape(16, kdim) % 7 - 3
        B = (torch.arange(kdim * 16, dtype=torch.float32).reshape(kdim, 16) * 3 % 5) - 2
        inp = torch.zeros(1024)
        if which == 1:
            inp[:512] = A.reshape(-1)
            inp[512:1024] = B.reshape(-1)
        else:
            inp[:64] = A.reshape(-1)
            inp[64:128] = B.reshape(-1)
        d_in, d_out = inp.to(DEV), torch.zeros(256, device=DEV)
        L.check(L.load().ctmi_probe(which, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), None), "probe")
        torch.cuda.synchronize()
        got = d_out.cpu().reshape(64, 4)
        D = A @ B
        exp = torch.empty(64, 4)
        for lane in range(64):
            for r in range(4):
                exp[lane, r] = D[(lane >> 4) * 4 + r, lane & 15]
        assert torch.equal(got, exp), f"MFMA layout assumption broken for probe {which}:\n{got[:8]}\n{exp[:8]}"


def test_probe_lds_transpose_read_dump():
    """ds_read_b64_tr_b16 with lane-linear addresses: dumped to gpurun_out for kernel design (no assertion on layout)."""
    L = lib()
    import ctypes as C
    outs = {}
    for name, addr in (("linear8", [l * 8 for l in range(64)]),
                       ("rows32", [(l & 15) * 32 + (l >> 4) * 8 for l in range(64)]),
                       ("rows128", [(l & 15) * 128 + (l >> 4) * 8 for l in range(64)])):
        d_in = torch.tensor(addr + [0] * (1024 - 64), dtype=torch.float32, device=DEV)
        d_out = torch.zeros(256, device=DEV)
        L.check(L.load().ctmi_probe(0, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), None), "probe")
        torch.cuda.synchronize()
        outs[name] = d_out.cpu().reshape(64, 4).to(torch.int64)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/tr_read_probe.txt", "w") as f:
        for k, v in outs.items():
            f.write(f"== {k} (u16 element index each lane received)\n")
            for lane in range(64):
                f.write(f"lane {lane:2d}: {v[lane].tolist()}\n")
    # lane-linear 8-byte addresses: lane l reads elements 4l..4l+3 of a [16 x 4] block per 16-lane group, transposed
    v = outs["linear8"]
    assert v.min() >= 0 and v.max() < 4096


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype,rtol,atol", DT)
@pytest.mark.parametrize("rows,cols", [(7, 48), (33, 64), (5, 211), (300, 1024), (64, 4096), (3, 24), (40, 2048), (9, 3072), (1030, 1536)])
def test_layernorm_fwd_bwd(dtype, rtol, atol, rows, cols):
    o = ops()
    x = rnd(rows, cols, seed=1) * 2 + 0.3
    w = 1 + 0.1 * rnd(cols, seed=2)
    b = 0.05 * rnd(cols, seed=3)
    gy = rnd(rows, cols, seed=4)
    dres = rnd(rows, cols, seed=5)
    bf = lo(dtype)
    xs, gys, drs = bf(x), bf(gy), bf(dres)
    xr = xs.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = R.layernorm(xr, wr, br, 1e-5)
    y_ref.backward(gys)
    y, mean, rstd = o.layernorm_fwd(to_dev(x, dtype), w.to(DEV), b.to(DEV), 1e-5)
    check("ln.y", y.float(), y_ref, rtol, atol)
    check("ln.mean", mean, xs.mean(-1), 1e-5, 1e-6)
    dx, dw, db = o.layernorm_bwd(to_dev(gy, dtype), to_dev(x, dtype), w.to(DEV), mean, rstd)
    check("ln.dx", dx.float(), xr.grad, rtol * 5, atol * 3)
    check("ln.dw", dw, wr.grad, 2e-4, 1e-4 * max(1, rows ** 0.5))
    check("ln.db", db, br.grad, 2e-4, 1e-4 * max(1, rows ** 0.5))
    dx2, _, _ = o.layernorm_bwd(to_dev(gy, dtype), to_dev(x, dtype), w.to(DEV), mean, rstd, dres=to_dev(dres, dtype))
    check("ln.dx+dres", dx2.float(), xr.grad + drs, rtol * 5, atol * 4)


def test_layernorm_module_golden_and_multidim():
    from cleantransformer_amd.transformer import LayerNorm
    OPS = np.load(os.path.join(G, "ops.npz"))
    ln = LayerNorm(48).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(OPS["ln_w"]))
        ln.bias.copy_(torch.from_numpy(OPS["ln_b"]))
    x = torch.from_numpy(OPS["ln_x"]).to(DEV).requires_grad_(True)
    y = ln(x)
    check("golden ln_y", y, torch.from_numpy(OPS["ln_y"]), 1e-5, 1e-6)
    y.backward(torch.from_numpy(OPS["ln_gy"]).to(DEV))
    check("golden ln_gx", x.grad, torch.from_numpy(OPS["ln_gx"]), 1e-4, 1e-6)
    check("golden ln_gw", ln.weight.grad, torch.from_numpy(OPS["ln_gw"]), 1e-4, 1e-5)
    check("golden ln_gb", ln.bias.grad, torch.from_numpy(OPS["ln_gb"]), 1e-4, 1e-5)
    ln2 = LayerNorm([4, 6]).to(DEV)                       # transformer.py:134-141 self-check shape
    check("golden ln2", ln2(torch.from_numpy(OPS["ln2_x"]).to(DEV)), torch.from_numpy(OPS["ln2_y"]), 1e-5, 1e-6)


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(128, 128, 64), (256, 384, 128), (100, 72, 40), (64, 211, 64), (37, 19, 211), (130, 260, 1000), (512, 1024, 1024), (512, 768, 256),
               (256, 512, 96), (160, 288, 352), (1184, 1568, 32)]   # odd numbers of 32-wide K-steps (3 / 5 / 9 / 11 / 37 / 49) and a single one: the
#                                                                     128-row ping-pong tile consumes K-steps in PAIRS and falls back to single steps


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float32, 1e-5, 1e-5), (torch.bfloat16, 1.2e-2, 1.2e-2), (torch.float16, 2e-3, 2e-3)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_forward_dgrad_wgrad(dtype, rtol, atol, M, N, K):
    o, L = ops(), lib()
    sc = 1.0 / math.sqrt(K)
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    bf = lo(dtype)
    xs, ws, rs = bf(x), bf(w), bf(res)
    xd, wd, rd = to_dev(x, dtype), to_dev(w, dtype), to_dev(res, dtype)
    # forward: y = x w^T + b (+res)
    y = o.linear_fwd(xd, wd, bias.to(DEV))
    check("gemm.fwd", y.float() * sc, (xs @ ws.t() + bias) * sc, rtol, atol)
    y = o.linear_fwd(xd, wd, bias.to(DEV), residual=rd)
    check("gemm.fwd+res", y.float() * sc, (xs @ ws.t() + bias + rs) * sc, rtol, atol)
    y = o.linear_fwd(xd, wd, None)
    check("gemm.fwd.nobias", y.float() * sc, (xs @ ws.t()) * sc, rtol, atol)
    # GELU epilogue keeps the pre-activation
    u = torch.empty((M, N), dtype=dtype, device=DEV)
    gl = o.linear_fwd(xd, wd, (bias * 0.1).to(DEV), epilogue=L.EPI_GELU, aux_out=u)
    pre = (xs @ ws.t()) * 1.0 + bias * 0.1
    check("gemm.gelu.pre", u.float() * sc, pre * sc, rtol, atol)
    pre_seen = bf(pre)
    check("gemm.gelu.out", gl.float() * sc, R.gelu_tanh(pre_seen) * sc, rtol * 3, atol * 3)
    # GELUG: same activation output, but the saved tensor is gelu'(pre) for the backward's MUL epilogue (round 2)
    gd = torch.empty((M, N), dtype=dtype, device=DEV)
    gl2 = o.linear_fwd(xd, wd, (bias * 0.1).to(DEV), epilogue=L.EPI_GELUG, aux_out=gd)
    assert torch.equal(gl2, gl)
    check("gemm.gelug.grad", gd.float(), R.gelu_tanh_bwd(torch.ones_like(pre_seen), pre_seen), max(rtol, 1e-4), max(atol, 1e-4))   # v_exp/v_rcp form
    rl = o.linear_fwd(xd, wd, (bias * 0.1).to(DEV), epilogue=L.EPI_RELU)
    check("gemm.relu", rl.float() * sc, torch.relu(pre) * sc, rtol, atol)
    # dgrad: dx = dy w   (w read K-major)
    dy = rnd(M, N, seed=5)
    dys = bf(dy)
    dyd = to_dev(dy, dtype)
    scn = 1.0 / math.sqrt(N)
    dx = o.linear_dgrad(dyd, wd)
    check("gemm.dgrad", dx.float() * scn, (dys @ ws) * scn, rtol, atol)
    aux = rnd(M, K, seed=6)
    auxs = bf(aux)
    dxg = o.linear_dgrad(dyd, wd, epilogue=L.EPI_DGELU, aux_in=to_dev(aux, dtype))
    check("gemm.dgrad.dgelu", dxg.float() * scn, R.gelu_tanh_bwd(dys @ ws, auxs) * scn, rtol * 2, atol * 2)
    dxm = o.linear_dgrad(dyd, wd, epilogue=L.EPI_MUL, aux_in=to_dev(aux, dtype))
    check("gemm.dgrad.mul", dxm.float() * scn, (dys @ ws) * auxs * scn, rtol * 2, atol * 2)
    dxr = o.linear_dgrad(dyd, wd, epilogue=L.EPI_DRELU, aux_in=to_dev(aux, dtype))
    check("gemm.dgrad.drelu", dxr.float() * scn, torch.where(auxs > 0, dys @ ws, torch.zeros(())) * scn, rtol, atol)
    # wgrad: dW = dy^T x  (fp32 out, both operands K-major), and accumulation
    scm = 1.0 / math.sqrt(M)
    dw = o.linear_wgrad(dyd, xd)
    assert dw.dtype == torch.float32
    check("gemm.wgrad", dw * scm, (dys.t() @ xs) * scm, rtol, atol)
    dw2 = o.linear_wgrad(dyd, xd, out=dw.clone(), accumulate=True)
    check("gemm.wgrad.acc", dw2 * scm, 2 * (dys.t() @ xs) * scm, rtol, atol)
    # bias grad
    db = o.colsum(dyd)
    check("colsum", db * scm, dys.sum(0) * scm, 1e-5, 1e-5)


def test_gemm_transpose_detecting_identity():
    """A = I with an asymmetric B (guide rule: symmetric inputs hide transposed C writes)."""
    o = ops()
    for dtype in (torch.float32, torch.bfloat16):
        n = 128
        eye = torch.eye(n)
        Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 13) - 6
        y = o.linear_fwd(to_dev(eye, dtype), to_dev(Bm, dtype), None)          # I @ B^T = B^T
        assert torch.equal(y.float().cpu(), Bm.t()), dtype


# ------------------------------------------------------------------------------------------------ attention
def _attn_oracle(qkv, am, nh, causal=True):
    B, S, _ = qkv.shape
    alibi = R.build_alibi(am, nh)
    masked = R.causal_key_mask(am, S) if causal else ~am[:, None, None, :].expand(B, 1, S, S).to(torch.bool)
    return R.attention_core(qkv, alibi, masked, nh)[0]


ATT_CASES = [  # B, S, nh, hd, mask kind
    (2, 16, 8, 8, "ones"), (3, 10, 8, 8, "mixed"), (2, 70, 4, 16, "mixed"), (2, 128, 2, 64, "right"),
    (1, 200, 2, 64, "left"), (2, 64, 2, 128, "ones"), (2, 130, 3, 32, "mixed")]


def _mask(kind, B, S):
    am = torch.ones(B, S, dtype=torch.long)
    if kind in ("right", "mixed") and B > 1:
        am[1, (S * 3) // 4:] = 0
    if kind in ("left", "mixed"):
        am[B - 1 if kind == "mixed" and B > 2 else 0, :max(1, S // 3)] = 0
    return am


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float32, 2e-5, 2e-6), (torch.bfloat16, 2e-2, 1e-2), (torch.float16, 4e-3, 2e-3)])
@pytest.mark.parametrize("B,S,nh,hd,kind", ATT_CASES)
def test_bloom_attention_fwd_bwd(dtype, rtol, atol, B, S, nh, hd, kind):
    o = ops()
    from cleantransformer_amd.models.modeling_bloom import alibi_slopes
    H = nh * hd
    qkv = rnd(B, S, 3 * H, seed=11)
    go = rnd(B, S, H, seed=12)
    am = _mask(kind, B, S)
    bf = lo(dtype)
    qs, gs = bf(qkv), bf(go)
    qr = qs.clone().requires_grad_(True)
    ctx_ref = _attn_oracle(qr, am, nh)
    ctx_ref.backward(gs)
    mask = o.MaskInfo(am.to(DEV))
    assert torch.equal(mask.kpos.cpu(), R.alibi_positions(am).float())                       # integer-valued: bit-exact
    assert torch.equal(mask.kvalid.cpu().long(), am)
    slopes = alibi_slopes(nh).to(DEV)
    qd = to_dev(qkv.reshape(B * S, 3 * H), dtype)
    desc = o.fused_qkv_desc(B, S, nh, hd, causal=True)
    out = torch.empty((B * S, H), dtype=dtype, device=DEV)
    sm, sl = o.attn_fwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, desc, slopes, mask)
    check("attn.out", out.float().view(B, S, H), ctx_ref, rtol, atol)
    dq = torch.zeros_like(qd)
    o.attn_bwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, to_dev(go.reshape(B * S, H), dtype), sm, sl,
               dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask)
    check("attn.dqkv", dq.float().view(B, S, 3 * H), qr.grad, rtol * 5, atol * 5)
    if dtype != torch.float32:
        check_norm("attn.dqkv (norms)", dq.float().view(B, S, 3 * H), qr.grad, 1e-2 if dtype == torch.bfloat16 else 2e-3, 2e-2 if dtype == torch.bfloat16 else 4e-3)
    else:
        check_norm("attn.dqkv (norms)", dq.float().view(B, S, 3 * H), qr.grad, 5e-5, 2e-4, 1e-4)


W32_CASES = [  # B, S, nh, hd, mask kind, future fill (0 = finfo.min: Bloom; -1e4: GPT-2)
    (2, 64, 2, 64, "ones", 0.0), (2, 128, 2, 64, "right", 0.0), (1, 192, 2, 64, "left", 0.0), (2, 256, 3, 64, "mixed", 0.0),
    (2, 320, 2, 64, "mixed", 0.0), (1, 512, 2, 64, "holes", 0.0), (2, 1024, 2, 64, "mixed", 0.0), (1, 576, 2, 64, "left", -1e4),
    (2, 256, 2, 128, "mixed", 0.0), (1, 448, 2, 128, "left", 0.0), (1, 384, 2, 128, "holes", -1e4), (1, 4096, 1, 64, "right", 0.0)]


def _w32_mask(kind, B, S):
    am = _mask(kind, B, S) if kind != "holes" else torch.ones(B, S, dtype=torch.long)
    if kind == "holes":
        am[:, 5::7] = 0
        am[0, :3] = 0
    return am


@pytest.mark.parametrize("td", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,S,nh,hd,kind,fill", W32_CASES)
def test_attention_128row_kernels_vs_oracle_and_general_kernels(B, S, nh, hd, kind, fill, td):
    """(round 5: the same kernels on IEEE-half operands as well — v_mfma_f32_32x32x16_f16 — with the same bounds: fp16 only has more mantissa.)
    csrc/attention_w32.hip (the bf16 training path: 32x32x16 MFMA, masks and row terms as MFMA C operands) on shapes with left /
    right / scattered padding, both fill kinds and both head sizes: against the CPU oracle (finfo.min fill; modeling_bloom.py:84-116),
    against the general kernels of csrc/attention.hip in the same process (GPT-2's -1e4 replacement, whose left-padding quirk rows the
    general kernels reproduce from the reference's golden: tests/test_gpu_gpt.py), on the PUBLISHED statistics (row maximum in the
    natural domain and row sum: the two families must be interchangeable — the first build of the C-operand form read the maximum before
    the last MFMA had landed and only this comparison saw it), and crosswise: the new forward feeding the general backward."""
    o = ops()
    from cleantransformer_amd.models.modeling_bloom import alibi_slopes
    FMIN = torch.finfo(torch.float32).min
    H = nh * hd
    bf = lo(td)
    qkv, go = bf(rnd(B, S, 3 * H, seed=S + hd, scale=0.7)), bf(rnd(B, S, H, seed=S + hd + 1, scale=0.5))
    am = _w32_mask(kind, B, S)
    mask = o.MaskInfo(am.to(DEV))
    slopes = alibi_slopes(nh).to(DEV)
    qd, god = qkv.reshape(B * S, 3 * H).to(DEV).to(td), go.reshape(B * S, H).to(DEV).to(td)

    def run(path_fwd, path_bwd):
        desc = o.fused_qkv_desc(B, S, nh, hd, causal=True)
        desc.future_fill = fill
        out = torch.empty((B * S, H), dtype=td, device=DEV)
        o.set_attn_path(path_fwd)
        sm, sl = o.attn_fwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, desc, slopes, mask)
        dq = torch.zeros_like(qd)
        o.set_attn_path(path_bwd)
        o.attn_bwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, god, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask)
        torch.cuda.synchronize()
        return out.float().cpu(), sm.cpu(), sl.cpu(), dq.float().cpu()

    try:
        o_old, m_old, l_old, g_old = run(0, 0)
        o_new, m_new, l_new, g_new = run(3, 3)
        o_x, _, _, g_x = run(1, 0)                                            # new forward (and its statistics) into the general backward
    finally:
        o.set_attn_path(3)
    fin = m_old > FMIN / 2
    assert torch.equal(m_new <= FMIN, m_old <= FMIN)                         # all-masked rows carry the same sentinel
    assert float((m_new - m_old)[fin].abs().max()) < 1e-3 and float(((l_new - l_old).abs() / l_old).max()) < 1e-3
    qr = qkv.clone().requires_grad_(True)
    if fill == 0.0:
        ref = _attn_oracle(qr, am, nh)                                        # modeling_bloom.py:84-116 (finfo.min fill)
    else:
        # GPT-2's replacement fill (modeling_gpt.py:88-93): oracle.gpt_ref.attention_core on the same heads, ALiBi as the score bias the
        # kernel adds before the replacement, the padding keys as the reference's additive finfo.min mask — in fp32, where
        # `score + finfo.min` is exactly finfo.min and a row whose whole causal window is padding attends uniformly to the FUTURE
        from oracle import gpt_ref as GR
        x = qr.view(B, S, nh, 3, hd)
        q, k, v = (x[..., i, :].transpose(1, 2) for i in range(3))
        bias = R.build_alibi(am, nh).view(B, nh, 1, S)
        add = ((1 - am).float() * FMIN).view(B, 1, 1, S)
        ref = GR.attention_core(q, k, v, add, score_bias=bias).transpose(1, 2).reshape(B, S, H)
    ref.backward(go)
    check("w32 out vs oracle", o_new.view(B, S, H), ref, 2e-2, 1e-2)
    check_norm("w32 out vs oracle (norms)", o_new.view(B, S, H), ref, 8e-3, 2e-2)
    # measured on MI355X (round 4): global 2.3e-3 - 2.5e-3, worst row 3.7e-3 of its norm: the bounds leave a factor 4 - 5
    check_norm("w32 dqkv vs oracle", g_new.view(B, S, 3 * H), qr.grad, 1e-2, 2e-2)
    check_norm("general dqkv vs oracle", g_old.view(B, S, 3 * H), qr.grad, 1e-2, 2e-2)
    check("w32 out vs general", o_new, o_old, 2e-2, 1e-2)
    check_norm("w32 dqkv vs general", g_new.view(B, S, 3 * H), g_old.view(B, S, 3 * H), 5e-3, 1.5e-2)
    check_norm("w32 fwd -> general bwd", g_x.view(B, S, 3 * H), g_old.view(B, S, 3 * H), 5e-3, 1.5e-2)
    assert torch.equal(o_x, o_new)


def test_attention_layer_golden_block():
    """One reference BloomAttentionLayer forward/backward incl. left padding (golden from the reference itself)."""
    o = ops()
    from cleantransformer_amd.models.modeling_bloom import alibi_slopes
    OPS = np.load(os.path.join(G, "ops.npz"))
    T = lambda k: torch.from_numpy(OPS[k])  # noqa: E731
    am = T("alibi_mask")
    hs, res, go = T("att_hs"), T("att_res"), T("att_go")
    B, S, H = hs.shape
    nh, hd = 8, H // 8
    wq, bq, wd, bd = T("att_p_query_key_value.weight"), T("att_p_query_key_value.bias"), T("att_p_dense.weight"), T("att_p_dense.bias")
    qkv = o.linear_fwd(hs.reshape(-1, H).to(DEV), wq.to(DEV), bq.to(DEV))
    mask = o.MaskInfo(am.to(DEV))
    slopes = alibi_slopes(nh).to(DEV)
    desc = o.fused_qkv_desc(B, S, nh, hd, causal=True)
    ctx = torch.empty((B * S, H), device=DEV)
    sm, sl = o.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ctx, desc, slopes, mask)
    out = o.linear_fwd(ctx, wd.to(DEV), bd.to(DEV), residual=res.reshape(-1, H).to(DEV))
    check("golden att_out", out.view(B, S, H), T("att_out"), 2e-5, 2e-6)
    g2 = go.reshape(-1, H).to(DEV)
    dctx = o.linear_dgrad(g2, wd.to(DEV))
    dqkv = torch.empty_like(qkv)
    o.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ctx, dctx, sm, sl, dqkv, dqkv[:, hd:], dqkv[:, 2 * hd:], desc, slopes, mask)
    dhs = o.linear_dgrad(dqkv, wq.to(DEV))
    check("golden att_ghs", dhs.view(B, S, H), T("att_ghs"), 1e-4, 2e-6)
    check("golden att_g_qkv_w", o.linear_wgrad(dqkv, hs.reshape(-1, H).to(DEV)), T("att_g_query_key_value.weight"), 1e-4, 2e-6)
    check("golden att_g_dense_w", o.linear_wgrad(g2, ctx), T("att_g_dense.weight"), 1e-4, 2e-6)
    check("golden att_g_dense_b", o.colsum(g2), T("att_g_dense.bias"), 1e-4, 2e-6)


def test_generic_attention_and_post_ln_block_golden():
    """transformer.py AttentionLayer / TransformerBlock vs goldens generated from the reference (dropout 0)."""
    from cleantransformer_amd.transformer import TransformerBlock
    OPS = np.load(os.path.join(G, "ops.npz"))

    class C:
        num_attention_heads = 4
        layer_norm_epsilong = 1e-5
        attention_probs_dropout_prob = 0.0
        hidden_size = 32
        hidden_dropout_prob = 0.0
    blk = TransformerBlock(C()).to(DEV)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            p.copy_(torch.from_numpy(OPS["blk_p_" + n]))
    x = torch.from_numpy(OPS["blk_x"]).to(DEV).requires_grad_(True)
    y = blk(x)
    check("golden blk_y", y, torch.from_numpy(OPS["blk_y"]), 2e-5, 2e-6)
    y.backward(torch.from_numpy(OPS["blk_go"]).to(DEV))
    check("golden blk_gx", x.grad, torch.from_numpy(OPS["blk_gx"]), 2e-4, 2e-6)
    for n, p in blk.named_parameters():
        check("golden blk_g_" + n, p.grad, torch.from_numpy(OPS["blk_g_" + n]), 2e-4, 4e-6)
    check("golden mha_y", blk.attention(x.detach()), torch.from_numpy(OPS["mha_y"]), 2e-5, 2e-6)
    check("golden mha_y_masked", blk.attention(x.detach(), attention_mask=torch.from_numpy(OPS["mha_addmask"]).to(DEV)),
          torch.from_numpy(OPS["mha_y_masked"]), 2e-5, 2e-6)


# ------------------------------------------------------------------------------------------------ CE / embedding
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-5), (torch.float16, 1e-5)])
@pytest.mark.parametrize("N,C,seq", [(37, 211, 37), (16, 1000, 8), (12, 250880, 6), (6, 77, 3)])
def test_cross_entropy_shifted(dtype, rtol, N, C, seq):
    o = ops()
    logits = rnd(N, C, seed=3) * 2
    ls = lo(dtype)(logits)
    labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(4))
    if seq == N:
        shift, rows, tgt = 0, torch.arange(N), labels
    else:
        shift = 1
        rows = torch.tensor([r for r in range(N) if (r % seq) + 1 < seq])
        tgt = labels[rows + 1]
    lr_ = ls.clone().requires_grad_(True)
    ref = R.cross_entropy(lr_[rows], tgt)
    ref.backward()
    ld = to_dev(logits, dtype)
    loss_out, row_lse = o.ce_fwd(ld, labels.to(DEV), seq=seq, shift=shift)
    check("ce.loss", loss_out[:1], ref.reshape(1), rtol, 0)
    check("ce.lse", row_lse, torch.logsumexp(ls.double(), -1), 2e-6, 1e-6)
    d = o.ce_bwd(ld, labels.to(DEV), row_lse, loss_out, torch.tensor([0.5], device=DEV), seq=seq, shift=shift)
    check("ce.dlogits", d.float(), 0.5 * lr_.grad, {torch.bfloat16: 1e-2, torch.float16: 2e-3}.get(dtype, 1e-5), 1e-9 if dtype == torch.float32 else 1e-6)
    if shift:
        dead = torch.tensor([r for r in range(N) if (r % seq) + 1 >= seq])
        assert float(d.float().cpu()[dead].abs().max()) == 0.0                      # rows without a target carry no gradient


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-5)])
def test_cross_entropy_odd_classes_padded_pitch(dtype, rtol):
    """GPT-2's V = 50257 in a buffer whose row pitch is padded to a multiple of 32 (LMHeadFn): the kernels run their vector
    body over the 8-divisible head and a scalar tail, on both the logits and the dlogits pitch."""
    o = ops()
    N, C, seq, Cp = 8, 50257, 4, 50272
    logits = rnd(N, C, seed=5) * 2
    ls = bf(logits) if dtype == torch.bfloat16 else logits
    labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(6))
    labels[3] = C - 1                                                       # a target inside the scalar tail
    rows = torch.tensor([r for r in range(N) if (r % seq) + 1 < seq])
    lr_ = ls.clone().requires_grad_(True)
    ref = R.cross_entropy(lr_[rows], labels[rows + 1])
    ref.backward()
    buf = torch.full((N, Cp), float("nan"), dtype=dtype, device=DEV)
    ld = buf[:, :C]
    ld.copy_(to_dev(logits, dtype))
    loss_out, row_lse = o.ce_fwd(ld, labels.to(DEV), seq=seq, shift=1)
    check("ce.loss", loss_out[:1], ref.reshape(1), rtol, 0)
    check("ce.lse", row_lse, torch.logsumexp(ls.double(), -1), 2e-6, 1e-6)
    dbuf = torch.zeros((N, Cp), dtype=dtype, device=DEV)
    d = o.ce_bwd(ld, labels.to(DEV), row_lse, loss_out, None, seq=seq, shift=1, out=dbuf[:, :C])
    check("ce.dlogits", d.float(), lr_.grad, 1e-2 if dtype == torch.bfloat16 else 1e-5, 1e-9 if dtype == torch.float32 else 1e-6)
    assert float(dbuf[:, C:].float().abs().max()) == 0.0                  # the pad columns are never written


def test_cross_entropy_module_golden_and_known_answers():
    import json
    from cleantransformer_amd.loss import CrossEntropyLoss
    OPS = np.load(os.path.join(G, "ops.npz"))
    lg = torch.from_numpy(OPS["ce_logits"]).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(OPS["ce_target"]).to(DEV)
    l = CrossEntropyLoss()(lg, tg)
    check("golden ce_repo_mean", l, torch.from_numpy(OPS["ce_repo_mean"]), 2e-6)
    check("golden ce_torch", l, torch.from_numpy(OPS["ce_torch"]), 2e-6)
    l.backward()
    check("golden ce_dlogits", lg.grad, torch.from_numpy(OPS["ce_dlogits"]), 1e-5, 1e-9)
    check("golden ce_repo_sum", CrossEntropyLoss('sum')(lg.detach(), tg), torch.from_numpy(OPS["ce_repo_sum"]), 2e-6)
    ka = json.load(open(os.path.join(G, "known_answers.json")))
    torch.manual_seed(999)                                                        # loss.py:76-100 self-check
    pred, gt = torch.rand(3, 4), torch.randint(0, 4, (3,))
    assert abs(float(CrossEntropyLoss()(pred.to(DEV), gt.to(DEV))) - ka["ce_index"]) < 2e-6


def test_embedding_gather_scatter_exact():
    o = ops()
    V, H, n = 211, 64, 300
    table = rnd(V, H, seed=5)
    ids = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(6))
    for dtype in (torch.float32, torch.bfloat16):
        td = to_dev(table, dtype)
        out = o.embed_fwd(td, ids.to(DEV))
        assert torch.equal(out.cpu(), td.cpu()[ids])                               # gather is bit-exact
    dout = rnd(n, H, seed=7)
    dt = torch.zeros(V, H, device=DEV)
    o.embed_bwd(dout.to(DEV), ids.to(DEV), dt)
    ref = torch.zeros(V, H).index_add_(0, ids, dout)
    check("embed.bwd", dt, ref, 1e-5, 1e-6)


# ------------------------------------------------------------------------------------------------ optimizers / utils
def _run_traj(make_opt, steps=50):
    OPS = np.load(os.path.join(G, "ops.npz"))
    w = torch.nn.Parameter(torch.from_numpy(OPS["opt_w0"]).clone().to(DEV))
    b = torch.nn.Parameter(torch.from_numpy(OPS["opt_b0"]).clone().to(DEV))
    opt = make_opt([w, b])
    gen = torch.Generator().manual_seed(13)
    for _ in range(steps):
        xin, tgt = torch.randn(4, 6, generator=gen), torch.randn(4, 5, generator=gen)
        wr, br = w.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
        ((xin @ wr + br - tgt) ** 2).sum().backward()                             # grads from the CPU problem definition
        opt.zero_grad()
        w.grad, b.grad = wr.grad.to(DEV), br.grad.to(DEV)
        opt.step()
    return w.detach().cpu(), b.detach().cpu()


@pytest.mark.parametrize("wd,tag", [(0.0, "wd0"), (0.01, "wd01")])
def test_adamw_trajectories_match_reference(wd, tag):
    from cleantransformer_amd.optimizer import AdamW
    OPS = np.load(os.path.join(G, "ops.npz"))
    w, b = _run_traj(lambda ps: AdamW(ps, lr=1e-2, weight_decay=wd))                           # repo AdamW == Adam + L2
    check("adam_repo_w", w, torch.from_numpy(OPS[f"adam_repo_{tag}_w"]), 2e-5, 2e-6)
    check("adam_repo_b", b, torch.from_numpy(OPS[f"adam_repo_{tag}_b"]), 2e-5, 2e-6)
    w, b = _run_traj(lambda ps: AdamW(ps, lr=1e-2, weight_decay=wd, decoupled=True))           # torch.optim.AdamW
    check("adam_torch_w", w, torch.from_numpy(OPS[f"adam_torch_{tag}_w"]), 2e-5, 2e-6)
    check("adam_torch_b", b, torch.from_numpy(OPS[f"adam_torch_{tag}_b"]), 2e-5, 2e-6)


def test_adamw_accepts_generator_and_many_tensors():
    from cleantransformer_amd.optimizer import AdamW
    ps = [torch.nn.Parameter(rnd(3 + i, 5, seed=i).to(DEV)) for i in range(60)]      # > CTMI_MT_MAX tensors, odd sizes
    ref = [p.detach().cpu().clone() for p in ps]
    opt = AdamW((p for p in ps), lr=1e-3, weight_decay=0.01, decoupled=True)         # a generator (reference bug Q2 not replicated)
    st = [(torch.zeros_like(r), torch.zeros_like(r)) for r in ref]
    for t in range(1, 4):
        for i, p in enumerate(ps):
            g = rnd(*p.shape, seed=100 * t + i)
            p.grad = g.to(DEV)
            R.adamw_update(ref[i], g.clone(), st[i][0], st[i][1], t, 1e-3, weight_decay=0.01, decoupled=True)
        opt.step()
    for i, p in enumerate(ps):
        check(f"adamw.many[{i}]", p, ref[i], 1e-5, 1e-7)


@pytest.mark.parametrize("decoupled", [True, False], ids=["decoupled", "l2"])
def test_adamw_chunk_balanced_launch_is_bit_identical_to_the_per_tensor_grid(decoupled):
    """Round 6: ctmi_adamw_step launches one workgroup per 16 Ki-element chunk, <= 64 tensors per launch (5 launches for Bloom-560M's 294
    tensors instead of 13).  Same arithmetic as the (stride loop, tensor) grid of rounds 1-5 (CTMI_OPT_LEGACY_GRID), no FMA contraction in
    either: parameters, both moments, the written-back gradient (L2 form) and the bf16 operand copies must agree BIT FOR BIT after three steps —
    on 70 tensors (two launches) that mix sizes below / at / above a chunk, sizes that are not a multiple of 4, a 4-byte-aligned view (scalar
    path) and one tensor with the 4 KiB / 8 KiB staggered state placement of optimizer._staggered."""
    o = ops()
    from cleantransformer_amd import optimizer as O
    g = torch.Generator().manual_seed(17)
    sizes = [1, 3, 4, 7, 1023, 4096, 16384, 16385, 16384 * 2 + 5, 70001, 1 << 18] + [129 + 17 * i for i in range(58)]
    base = torch.randn(50000, generator=g).to(DEV)

    def make():
        ps = [torch.randn(n, generator=torch.Generator().manual_seed(100 + i)).to(DEV) for i, n in enumerate(sizes)]
        ps.append(base.clone()[1:1 + 40003])                                              # 4-byte aligned, not 16: scalar path
        ms = [torch.zeros_like(x) for x in ps]
        vs = [torch.zeros_like(x) for x in ps]
        big = sizes.index(1 << 18)
        ms[big], vs[big] = O._staggered(ps[big], 1), O._staggered(ps[big], 2)
        assert ms[big].data_ptr() % (2 << 20) != vs[big].data_ptr() % (2 << 20)
        sh = [torch.zeros(x.numel(), dtype=torch.bfloat16, device=DEV) if i % 3 else None for i, x in enumerate(ps)]
        return ps, ms, vs, sh

    out = []
    for legacy in (False, True):
        ps, ms, vs, sh = make()
        gs_all = []
        for t in range(1, 4):
            gs = [torch.randn(x.numel(), generator=torch.Generator().manual_seed(1000 * t + i)).to(DEV) for i, x in enumerate(ps)]
            o.adamw_step(ps, gs, ms, vs, sh, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=t, decoupled=decoupled,
                         mutate_grad=not decoupled, grad_scale=0.5 if t == 2 else 1.0, legacy_grid=legacy)
            gs_all.append(gs)
        torch.cuda.synchronize()
        out.append((ps, ms, vs, sh, gs_all))
    a, b = out
    for i in range(len(a[0])):
        for k in range(3):
            assert torch.equal(a[k][i], b[k][i]), (i, "pmv"[k], int(a[k][i].numel()))
        if a[3][i] is not None:
            assert torch.equal(a[3][i], b[3][i]), (i, "shadow")
            assert torch.equal(a[3][i].float(), a[0][i].to(torch.bfloat16).float()), (i, "shadow is the rounded parameter")
        for t in range(3):
            assert torch.equal(a[4][t][i], b[4][t][i]), (i, "grad", t)
    # and against the oracle's update rule (fp64-free restatement of optimizer.py:75-95 / torch.optim.AdamW) on one tensor
    i = sizes.index(70001)
    ref_p = torch.randn(70001, generator=torch.Generator().manual_seed(100 + i))
    m0, v0 = torch.zeros(70001), torch.zeros(70001)
    for t in range(1, 4):
        gr = torch.randn(70001, generator=torch.Generator().manual_seed(1000 * t + i)) * (0.5 if t == 2 else 1.0)
        R.adamw_update(ref_p, gr, m0, v0, t, 1e-2, weight_decay=0.01, decoupled=decoupled)
    check("adamw.flat.vs_oracle", a[0][i], ref_p, 1e-5, 1e-6)


@pytest.mark.parametrize("cd", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_adamw_refreshes_bf16_shadow(cd):
    """the fused optimizers write the operand copy of a weight in the same pass — bf16, or IEEE half (CTMI_OPT_SHADOW_F16, round 5); a parameter
    whose copy changes dtype (an autocast context of another dtype) gets a new copy from the next compute_weight()"""
    o = ops()
    from cleantransformer_amd.optimizer import AdamW, SGD
    rd = lo(cd)
    p = torch.nn.Parameter(rnd(300, 40, seed=3).to(DEV))
    sh = o.compute_weight(p, cd)
    assert sh.dtype == cd and torch.equal(sh.float().cpu(), rd(p.detach().cpu()))
    opt = AdamW([p], lr=1e-1, decoupled=True)
    p.grad = rnd(300, 40, seed=4).to(DEV)
    opt.step()
    sh2 = o.compute_weight(p, cd)
    assert sh2.data_ptr() == sh.data_ptr()                                           # written in place by the fused kernel
    assert torch.equal(sh2.float().cpu(), rd(p.detach().cpu()))
    sgd = SGD([p], lr=1e-2)
    p.grad = rnd(300, 40, seed=5).to(DEV)
    sgd.step()
    assert torch.equal(o.compute_weight(p, cd).float().cpu(), rd(p.detach().cpu()))
    other = torch.float16 if cd == torch.bfloat16 else torch.bfloat16
    sh3 = o.compute_weight(p, other)
    assert sh3.dtype == other and torch.equal(sh3.float().cpu(), lo(other)(p.detach().cpu()))


def test_sgd_trajectory_matches_reference():
    from cleantransformer_amd.optimizer import SGD
    OPS = np.load(os.path.join(G, "ops.npz"))
    w, b = _run_traj(lambda ps: SGD(ps, lr=1e-2, momentum=0.9, weight_decay=0.01))
    check("sgd_w", w, torch.from_numpy(OPS["sgd_repo_w"]), 2e-5, 2e-6)
    check("sgd_b", b, torch.from_numpy(OPS["sgd_repo_b"]), 2e-5, 2e-6)


def test_utils_cast_sumsq_argmax():
    o = ops()
    x = rnd(1000, 37, seed=9)
    xd = x.to(DEV)
    assert torch.equal(o.cast(xd, torch.bfloat16).cpu(), x.to(torch.bfloat16))       # RNE conversion is bit-exact
    check("sumsq", o.sumsq(xd.reshape(-1)), x.double().pow(2).sum().reshape(1), 1e-12)
    for dtype in (torch.float32, torch.bfloat16):
        big = to_dev(rnd(5, 250880, seed=2), dtype)
        assert torch.equal(o.argmax_lastdim(big).cpu(), big.float().cpu().argmax(-1))   # token ids: bit-exact
    tie = torch.zeros(3, 50, device=DEV)
    tie[:, 7] = 1
    tie[:, 30] = 1
    assert o.argmax_lastdim(tie).tolist() == [7, 7, 7]                               # first maximal index, like torch.argmax


def test_scale_and_scale_copy_bit_exact():
    """DDP bucket fill: dst = s * src is one fp32 multiply per element (bit-exact vs torch), aligned and unaligned,
    out of place and in place (trainer/ddp.py: torch-DDP's divide-then-all-reduce order)."""
    o = ops()
    for n, off in ((1 << 20, 0), (12345, 0), (4099, 1), (7, 3)):
        base = rnd(n + 8, seed=n).to(DEV)
        src = base[off:off + n]
        dst = torch.empty(n + 8, device=DEV)[off:off + n]
        o.scale_copy(src, dst, 0.125)
        assert torch.equal(dst.cpu(), (src * 0.125).cpu())
        o.scale_copy(src, dst, 1.0 / 3.0)
        assert torch.equal(dst.cpu(), (src * torch.tensor(1.0 / 3.0, dtype=torch.float32)).cpu())
        want = (src * torch.tensor(1.0 / 7.0, dtype=torch.float32)).cpu()
        o.scale_copy(src, src, 1.0 / 7.0)                                              # aliasing: scaled where it is
        assert torch.equal(src.cpu(), want)
        y = rnd(n, seed=3).to(DEV)
        want = (y * 0.5).cpu()
        o.scale_(y, 0.5)
        assert torch.equal(y.cpu(), want)


@pytest.mark.parametrize("K", [192, 160, 224])
def test_gemm_more_tiles_than_slots(K):
    """Persistent launches: more output tiles than resident workgroups, so workgroups walk several tiles and the DMA stream, the
    cross-tile prefetch run across tile boundaries.  All three operand layouts,
    bf16, against torch on the same device in fp32.  K = 160 / 224: an ODD number of K-steps per tile (5 / 7), so on the tile that consumes
    K-steps in pairs the DMA stream (four stages ahead) changes work items in the middle of a pair and single steps alternate with pairs."""
    o = ops()
    M, N = 4096, 4352                                         # 16 x 17 = 272 tiles of 256 x 256 (> 256 CUs), 6 / 5 / 7 K-steps each
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    dy = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    y = o.linear_fwd(x, w, None)
    ref = x.float() @ w.float().t()
    assert torch.allclose(y.float(), ref, rtol=1.2e-2, atol=1.2e-2 * math.sqrt(K)), (y.float() - ref).abs().max()
    dx = o.linear_dgrad(dy, w)                                # [M, K]: small output, long K
    ref = dy.float() @ w.float()
    assert torch.allclose(dx.float(), ref, rtol=1.2e-2, atol=1.2e-2 * math.sqrt(N)), (dx.float() - ref).abs().max()
    x2 = (torch.randn(1024, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV)      # wgrad with a [4096, 4352] output: dy2^T x2
    dy2 = (torch.randn(1024, M, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    dw = o.linear_wgrad(dy2, x2)
    ref = dy2.float().t() @ x2.float()
    assert torch.allclose(dw.float(), ref, rtol=1e-3, atol=1e-2), (dw.float() - ref).abs().max()


STEP_GEMMS = [  # name, kind, rows T, out features, in features, epilogue            (BASELINE configs[1]: T = 8 x 1024, H = 1024)
    ("qkv", "fwd", 8192, 3072, 1024, "bias"), ("dense", "fwd", 8192, 1024, 1024, "bias+res"), ("h4h", "fwd", 8192, 4096, 1024, "gelu"),
    ("4hh", "fwd", 8192, 1024, 4096, "bias+res"), ("qkv", "dgrad", 8192, 3072, 1024, None), ("dense", "dgrad", 8192, 1024, 1024, None),
    ("h4h", "dgrad", 8192, 4096, 1024, None), ("4hh", "dgrad", 8192, 1024, 4096, "dgelu"), ("4hh", "dgrad", 8192, 1024, 4096, "mul"),
    ("qkv", "wgrad", 8192, 3072, 1024, None), ("dense", "wgrad", 8192, 1024, 1024, None), ("h4h", "wgrad", 8192, 4096, 1024, None),
    ("4hh", "wgrad", 8192, 1024, 4096, None), ("lm_head", "fwd", 8192, 250880, 1024, None), ("lm_head", "dgrad", 8192, 250880, 1024, None),
    ("lm_head", "wgrad", 8192, 250880, 1024, None),
    # Bloom-7B1 geometry on one GPU (BASELINE configs[4]: T = 2 x 2048, H = 4096): the [T,H] outputs — exactly ONE full round of 256x256 tiles with
    # K = 4096 / 12288 / 16384 — take the 256-row ping-pong tile since round 6 (pick_tile: t2 == 256 && K >= 2048), the residual forwards on its
    # non-prefetching epilogue variant
    ("7b1-dense", "fwd", 4096, 4096, 4096, "bias+res"), ("7b1-4hh", "fwd", 4096, 4096, 16384, "bias+res"), ("7b1-qkv", "dgrad", 4096, 12288, 4096, None),
    ("7b1-dense", "dgrad", 4096, 4096, 4096, None), ("7b1-h4h", "dgrad", 4096, 16384, 4096, None)]


@pytest.mark.parametrize("name,kind,T,Nout,Kin,epi", STEP_GEMMS, ids=lambda v: str(v))
def test_gemm_at_the_step_shapes_sampled_vs_fp64(name, kind, T, Nout, Kin, epi):
    """Every GEMM of the Bloom-560M step (modeling_bloom.py:79,121,256,267,220 and their autograd) at its REAL size — T = 8192 rows,
    K in {1024, 3072, 4096, 8192, 250 880}: the persistent ping-pong launches, their steady K-loop, the split-K and unsplit
    weight-gradient tiles, the logits-sized non-temporal epilogue — element by element on a 64 x 64 sample of the output (rows and
    columns drawn from every tile row / column region incl. the first and last) against an fp64 CPU product of the bf16-rounded
    operands.  Bound: bf16 outputs one rounding (2^-8 relative of the value) + fp32 accumulation noise (1e-5 of sqrt(K) * sigma^2);
    fp32 outputs (weight gradients) 2e-5 of the row scale."""
    o, L = ops(), lib()
    dev_g = torch.Generator(device=DEV).manual_seed(sum(map(ord, name + kind)) * 7919 + Nout)
    bfr = lambda *sh, sc=0.5: (torch.randn(*sh, generator=dev_g, device=DEV) * sc).to(torch.bfloat16)   # noqa: E731
    cpu_g = torch.Generator().manual_seed(7)

    def pick(n, k=64):
        edge = torch.tensor([0, 1, n - 2, n - 1, 255, 256, n // 2 - 1, n // 2])
        return torch.unique(torch.cat([edge.clamp(0, n - 1), torch.randint(0, n, (k - len(edge),), generator=cpu_g)]))
    d64 = lambda t: t.double().cpu()                                                                      # noqa: E731
    if kind == "fwd":
        x, w = bfr(T, Kin), bfr(Nout, Kin)
        bias = torch.randn(Nout, generator=dev_g, device=DEV) if epi else None
        res = bfr(T, Nout) if epi == "bias+res" else None
        u = torch.empty(T, Nout, dtype=torch.bfloat16, device=DEV) if epi == "gelu" else None
        y = o.linear_fwd(x, w, bias, residual=res, epilogue=L.EPI_GELU if epi == "gelu" else L.EPI_NONE, aux_out=u)
        r, c = pick(T), pick(Nout)
        ref = d64(x[r.to(DEV)]) @ d64(w[c.to(DEV)]).t()
        if bias is not None:
            ref = ref + d64(bias[c.to(DEV)])
        if epi == "gelu":
            got_u = d64(u[r.to(DEV)][:, c.to(DEV)])
            assert float(((got_u - ref).abs() / (ref.abs() + 1)).max()) < 6e-3
            ref = R.gelu_tanh(got_u)                                             # the activation sees the stored pre-activation
        if res is not None:
            ref = ref + d64(res[r.to(DEV)][:, c.to(DEV)])
        got = d64(y[r.to(DEV)][:, c.to(DEV)])
        scale, rt = math.sqrt(Kin) * 0.25, 4.5e-3
    elif kind == "dgrad":
        dy, w = bfr(T, Nout), bfr(Nout, Kin, sc=0.5 if Nout < 65536 else 0.05)
        aux = bfr(T, Kin) if epi in ("dgelu", "mul") else None
        dx = o.linear_dgrad(dy, w, epilogue={"dgelu": L.EPI_DGELU, "mul": L.EPI_MUL}.get(epi, L.EPI_NONE), aux_in=aux)
        r, c = pick(T), pick(Kin)
        ref = d64(dy[r.to(DEV)]) @ d64(w[:, c.to(DEV)])
        sig = 0.25 if Nout < 65536 else 0.025
        scale, rt = math.sqrt(Nout) * sig, 4.5e-3
        if aux is not None:
            a = d64(aux[r.to(DEV)][:, c.to(DEV)])
            ref = R.gelu_tanh_bwd(ref, a) if epi == "dgelu" else ref * a
            rt = 6e-3                                                            # + the v_exp / v_rcp form of the derivative
        got = d64(dx[r.to(DEV)][:, c.to(DEV)])
    else:
        dy, x = bfr(T, Nout), bfr(T, Kin)
        dw = o.linear_wgrad(dy, x)
        assert dw.dtype == torch.float32 and dw.shape == (Nout, Kin)
        r, c = pick(Nout), pick(Kin)
        ref = d64(dy[:, r.to(DEV)]).t() @ d64(x[:, c.to(DEV)])
        got = d64(dw[r.to(DEV)][:, c.to(DEV)])
        scale, rt = math.sqrt(T) * 0.25, 2e-5
    err = (got - ref).abs()
    bound = rt * ref.abs() + 2e-5 * scale
    if os.environ.get("CTMI_TEST_VERBOSE"):
        print(f"[step gemm] {name} {kind} {epi}: max err {float(err.max()):.3e}, max err/bound {float((err / bound).max()):.3f}, scale {scale:.2f}")
    assert torch.isfinite(got).all()
    assert bool((err <= bound).all()), f"{name} {kind}: {int((err > bound).sum())}/{err.numel()} sampled elements off, worst {float(err.max()):.3e} (bound there {float(bound.flatten()[err.argmax()]):.3e})"


@pytest.mark.parametrize("name,kind,T,Nout,Kin,epi", [c for c in STEP_GEMMS if not (c[0] == "lm_head" and c[1] == "fwd")], ids=lambda v: str(v))
def test_gemm_with_cold_operands_equals_warm_bit_for_bit(name, kind, T, Nout, Kin, epi):
    """A race detector for the LDS-DMA rings.  The counted waits of the K-loops (`s_waitcnt vmcnt(N)`) are hand-written: one that is too
    loose reads a ring stage before it has landed — and with L2-warm operands (every other GEMM test: operands just generated or used) the
    stage has ALWAYS landed by then, so the result is right.  Round 4 shipped such a wait for an hour (the first two-stages-per-phase build
    of the 128-row tile read stage 1 after waiting for stage 0 only): 82 GEMM parity tests green, NaN loss in the training step, where
    operands come from HBM.  The kernels are deterministic (split-K partials are reduced in a fixed order), so: the result with operands
    evicted from L2 and the memory-side cache (3 GiB written in between) must equal the warm result bit for bit, six times."""
    o, L = ops(), lib()
    dev_g = torch.Generator(device=DEV).manual_seed(sum(map(ord, name + kind)) * 104729 + Kin)
    bfr = lambda *sh, sc=0.5: (torch.randn(*sh, generator=dev_g, device=DEV) * sc).to(torch.bfloat16)   # noqa: E731
    if kind == "fwd":
        x, w = bfr(T, Kin), bfr(Nout, Kin)
        bias = torch.randn(Nout, generator=dev_g, device=DEV) if epi else None
        res = bfr(T, Nout) if epi == "bias+res" else None

        def run():
            u = torch.empty(T, Nout, dtype=torch.bfloat16, device=DEV) if epi == "gelu" else None
            y = o.linear_fwd(x, w, bias, residual=res, epilogue=L.EPI_GELU if epi == "gelu" else L.EPI_NONE, aux_out=u)
            return (y, u) if u is not None else (y,)
    elif kind == "dgrad":
        dy, w = bfr(T, Nout), bfr(Nout, Kin, sc=0.5 if Nout < 65536 else 0.05)
        aux = bfr(T, Kin) if epi in ("dgelu", "mul") else None

        def run():
            return (o.linear_dgrad(dy, w, epilogue={"dgelu": L.EPI_DGELU, "mul": L.EPI_MUL}.get(epi, L.EPI_NONE), aux_in=aux),)
    else:
        dy, x = bfr(T, Nout), bfr(T, Kin)

        def run():
            return (o.linear_wgrad(dy, x),)
    run()
    warm = run()
    assert all(bool(torch.isfinite(t.float()).all()) for t in warm)
    evict = torch.empty(3 << 28, dtype=torch.int32, device=DEV)               # 3 GiB of stores: > 256 MiB Infinity Cache + 32 MiB of L2s
    for i in range(6):
        evict.fill_(i)
        cold = run()
        for a, b in zip(cold, warm):
            assert torch.equal(a, b), f"{name} {kind} {epi}: run {i} with cold operands differs from the warm result in {int((a != b).sum())} elements"


@pytest.mark.parametrize("env", [{"CTMI_GEMM_TILE": "0"}, {"CTMI_GEMM_TILE": "1"}, {"CTMI_GEMM_TILE": "2"}, {"CTMI_GEMM_TILE": "3"},
                                 {"CTMI_GEMM_TILE": "4"}, {"CTMI_GEMM_SHARED": "1"}, {"CTMI_GEMM_PERSIST": "0"},
                                 {"CTMI_GEMM_TILE": "3", "CTMI_GEMM_SPLIT": "1"}, {"CTMI_GEMM_GLDS": "0"}],
                         ids=lambda e: ",".join(f"{k[10:]}={v}" for k, v in e.items()))
def test_gemm_every_tile_schedule_and_policy(env):
    """The launcher picks tile / schedule / split per shape and caches its environment overrides per process, so the
    default run exercises only what the picker chooses for the test shapes.  Re-run the GEMM parity tests in a fresh
    process under every override: 128x128, 256x128, 256x256 free-running, 256x256 and 128x256 ping-pong (LDS-shuffled
    epilogue), the shared-GPU policy, non-persistent launches, no split-K, and the register-staged v1 kernel."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "test_gemm_forward_dgrad_wgrad or test_gemm_transpose_detecting_identity or test_linear or test_gemm_more_tiles_than_slots"],
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float32, 1e-5, 1e-6), (torch.bfloat16, 2e-2, 2e-3)])
def test_soft_target_cross_entropy_vs_reference_golden(dtype, rtol, atol):
    """Probability-target branch of CrossEntropyLoss (loss.py:43-46) through ctmi_ce_soft_fwd/bwd: the reference's own losses and
    input gradients (tests/golden/soft_ce.npz), both reductions, normalised and raw targets, its printed self-check value, and a
    wide row (C = 50257) against the oracle."""
    from cleantransformer_amd.loss import CrossEntropyLoss
    from oracle import bloom_ref as R
    SC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "soft_ce.npz"))
    x = torch.from_numpy(SC["x"])
    for name in ("norm", "raw"):
        t = torch.from_numpy(SC[f"t_{name}"]).to(DEV)
        for red in ("mean", "sum"):
            xi = x.to(dtype).to(DEV).requires_grad_(True)
            loss = CrossEntropyLoss(red)(xi, t)
            (loss * 3.0).backward()
            ref_l, ref_g = float(SC[f"loss_{name}_{red}"]), torch.from_numpy(SC[f"dx_{name}_{red}"])
            if dtype == torch.bfloat16:                                    # the kernel sees bf16-rounded logits: compare with the oracle on those
                xr = x.to(dtype).float().requires_grad_(True)
                lr = R.cross_entropy_repo(xr, t.cpu(), red)
                lr.backward()
                ref_l, ref_g = float(lr), xr.grad
            assert abs(float(loss) - ref_l) <= 1e-5 * abs(ref_l) + 1e-6, (name, red, float(loss), ref_l)
            check(f"soft_ce.dx[{name},{red}]", xi.grad.float() / 3.0, ref_g, rtol, atol)
    if dtype == torch.float32:
        k = CrossEntropyLoss('mean')(torch.from_numpy(SC["known_pred"]).to(DEV), torch.from_numpy(SC["known_t"]).to(DEV))
        assert abs(float(k) - 3.14231014) < 1e-6
    g = torch.Generator().manual_seed(8)
    xw = (torch.randn(5, 50257, generator=g) * 3).to(dtype)
    tw = torch.softmax(torch.randn(5, 50257, generator=g) * 2, dim=-1)
    xi = xw.to(DEV).requires_grad_(True)
    loss = CrossEntropyLoss('mean')(xi, tw.to(DEV))
    loss.backward()
    xr = xw.double().requires_grad_(True)
    lr = -(tw.double() * torch.log_softmax(xr, -1)).sum() / 5
    lr.backward()
    assert abs(float(loss) - float(lr)) <= 2e-6 * float(lr)
    check("soft_ce.wide.dx", xi.grad.float(), xr.grad.float(), rtol, max(atol * 1e-3, 1e-9) if dtype == torch.float32 else 1e-6)
