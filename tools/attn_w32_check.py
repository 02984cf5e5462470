#!/usr/bin/env python3
"""GPU check + timing of the 256-row / 32x32-MFMA attention kernels (csrc/attention_w32.hip) against the general kernels of
csrc/attention.hip in the same process (ctmi_attn_set_path) and against an fp32 torch restatement on the CPU.
Usage: python tools/attn_w32_check.py [check] [time]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cleantransformer_amd import ops
from cleantransformer_amd.models.modeling_bloom import alibi_slopes

DEV = "cuda:0"
BF = torch.bfloat16
FMIN = torch.finfo(torch.float32).min


def mask_of(kind, B, S):
    am = torch.ones(B, S, dtype=torch.long)
    if kind in ("right", "mixed") and B > 1:
        am[1, (S * 3) // 4:] = 0
    if kind in ("left", "mixed"):
        am[0, :max(1, S // 3)] = 0
    if kind == "holes":
        am[:, 5::7] = 0
        am[0, :3] = 0
    return am


def cpu_ref(qkv, go, am, nh, hd, fill):
    """fp32 restatement: scores = scale*q.k + slope*pos, padding keys -> finfo.min, causal future -> fill (padding stays finfo.min)."""
    B, S, _ = qkv.shape
    x = qkv.float().view(B, S, nh, 3, hd).clone().requires_grad_(True)
    q, k, v = x[:, :, :, 0].transpose(1, 2), x[:, :, :, 1].transpose(1, 2), x[:, :, :, 2].transpose(1, 2)
    pos = ((am.cumsum(-1) - 1) * am).float()
    slopes = alibi_slopes(nh).float()
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + slopes[None, :, None, None] * pos[:, None, None, :]
    pad = (am == 0)[:, None, None, :].expand(B, nh, S, S)
    fut = torch.ones(S, S, dtype=torch.bool).triu(1)[None, None].expand(B, nh, S, S)
    s = torch.where(fut & ~pad, torch.full_like(s, fill), s)
    s = torch.where(pad, torch.full_like(s, FMIN), s)
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B, S, nh * hd)
    o.backward(go.float())
    return o.detach(), x.grad.reshape(B, S, 3 * nh * hd)


def run(path, qd, god, B, S, nh, hd, mask, slopes, fill):
    ops.set_attn_path(path)
    H = nh * hd
    desc = ops.fused_qkv_desc(B, S, nh, hd, causal=True)
    desc.future_fill = fill
    out = torch.empty((B * S, H), dtype=BF, device=DEV)
    sm, sl = ops.attn_fwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, desc, slopes, mask)
    dq = torch.zeros_like(qd)
    ops.attn_bwd(qd, qd[:, hd:], qd[:, 2 * hd:], out, god, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask)
    torch.cuda.synchronize()
    return out.float().cpu(), sm.cpu(), sl.cpu(), dq.float().cpu()


def err(a, b):
    d = (a - b).abs()
    return float(d.max()), float(d.max() / (b.abs().max() + 1e-30))


def check():
    cases = [(2, 64, 2, 64, "ones", 0.0), (2, 128, 2, 64, "right", 0.0), (1, 192, 2, 64, "left", 0.0), (2, 256, 3, 64, "mixed", 0.0),
             (2, 320, 2, 64, "mixed", 0.0), (1, 512, 2, 64, "holes", 0.0), (2, 1024, 2, 64, "mixed", 0.0), (1, 576, 2, 64, "left", -1e4),
             (2, 128, 2, 128, "ones", 0.0), (2, 256, 2, 128, "mixed", 0.0), (1, 448, 2, 128, "left", 0.0), (1, 384, 2, 128, "holes", -1e4)]
    bad = 0
    for B, S, nh, hd, kind, fill in cases:
        torch.manual_seed(S + hd)
        H = nh * hd
        qkv = (torch.randn(B, S, 3 * H) * 0.7).to(BF)
        go = (torch.randn(B, S, H) * 0.5).to(BF)
        am = mask_of(kind, B, S)
        mask = ops.MaskInfo(am.to(DEV))
        slopes = alibi_slopes(nh).to(DEV)
        qd, god = qkv.reshape(B * S, 3 * H).to(DEV), go.reshape(B * S, H).to(DEV)
        o_ref, g_ref = cpu_ref(qkv, go, am, nh, hd, FMIN if fill == 0.0 else fill)
        o0, m0, l0, g0 = run(0, qd, god, B, S, nh, hd, mask, slopes, fill)
        o1, m1, l1, g1 = run(3, qd, god, B, S, nh, hd, mask, slopes, fill)
        # mixed: new forward statistics feeding the general backward and vice versa (they must be interchangeable)
        e_old = (err(o0.view(B, S, H), o_ref), err(g0.view(B, S, 3 * H), g_ref))
        e_new = (err(o1.view(B, S, H), o_ref), err(g1.view(B, S, 3 * H), g_ref))
        fin = m0 > FMIN / 2
        e_m = float((m1 - m0)[fin].abs().max()) if fin.any() else 0.0
        same_min = bool(((m1 <= FMIN) == (m0 <= FMIN)).all())
        e_l = float(((l1 - l0).abs() / l0).max())
        ok = e_new[0][1] <= max(2.5 * e_old[0][1], 8e-3) and e_new[1][1] <= max(2.5 * e_old[1][1], 1.6e-2) and e_m < 1e-3 and e_l < 1e-3 and same_min   # (l: both families sum the unrounded probabilities)
        ok = ok and bool(torch.isfinite(o1).all()) and bool(torch.isfinite(g1).all())
        bad += 0 if ok else 1
        print(f"{'OK ' if ok else 'BAD'} B={B} S={S} nh={nh} hd={hd} {kind:5s} fill={fill:g}: out rel err new {e_new[0][1]:.2e} old {e_old[0][1]:.2e} | "
              f"dqkv rel err new {e_new[1][1]:.2e} old {e_old[1][1]:.2e} | stat m {e_m:.1e} l {e_l:.1e} minrows {'same' if same_min else 'DIFF'}", flush=True)
        if not ok:
            d = (o1.view(B, S, H) - o_ref).abs().amax(-1)
            print("   worst out rows:", [(int(i // S), int(i % S)) for i in d.flatten().topk(6).indices])
            dg = (g1.view(B, S, nh, 3, hd) - g_ref.view(B, S, nh, 3, hd)).abs().amax(-1)
            print("   dq/dk/dv max err:", [float(dg[..., j].max()) for j in range(3)])
    print("attn_w32 check:", "ALL OK" if bad == 0 else f"{bad} BAD")
    return bad


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_():
    cases = ((8, 1024, 16, 64), (4, 2048, 16, 64), (4, 2048, 32, 128))
    if os.environ.get("W32_CASES"):
        cases = tuple(cases[int(i)] for i in os.environ["W32_CASES"].split(","))
    paths = tuple(int(x) for x in os.environ.get("W32_PATHS", "0,3,0,3").split(","))
    for B, S, nh, hd in cases:
        H, T = nh * hd, B * S
        qkv = (torch.randn(T, 3 * H, device=DEV) * 0.5).to(BF)
        go = (torch.randn(T, H, device=DEV) * 0.5).to(BF)
        mask = ops.MaskInfo(torch.ones(B, S, dtype=torch.long, device=DEV))
        slopes = alibi_slopes(nh).to(DEV)
        desc = ops.fused_qkv_desc(B, S, nh, hd, causal=True)
        out = torch.empty((T, H), dtype=BF, device=DEV)
        dq = torch.zeros_like(qkv)
        fl = 4.0 * B * nh * S * S * hd / 2
        for path in paths:
            ops.set_attn_path(path)
            sm, sl = ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask)
            tf = timeit(lambda: ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask))
            tb = timeit(lambda: ops.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, go, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask))
            print(f"B={B} S={S} nh={nh} hd={hd} path={path}: fwd {tf * 1e3:7.1f} us {fl / tf / 1e9:7.1f} TF/s | bwd {tb * 1e3:7.1f} us {2.5 * fl / tf / 1e9 * tf / tb:7.1f} TF/s (causal-half flops)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        time_()
    sys.exit(1 if rc else 0)
