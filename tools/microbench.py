#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the Bloom-560M C2 shapes (T = 8192 tokens): HIP-event timing, TFLOP/s or GB/s.
Usage: python tools/microbench.py [gemm] [wgroup] [attn] [ln] [ce] [adamw] [embed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cleantransformer_amd import _lib, ops
from cleantransformer_amd.models.modeling_bloom import alibi_slopes

DEV = "cuda:0"
BF = torch.bfloat16


def timeit(fn, iters=10, warm=2):
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


def rnd(*s, dtype=BF):
    return (torch.randn(*s, device=DEV) * 0.5).to(dtype)


def bench_gemm(T=8192):
    H, V = 1024, 250880
    print(f"--- GEMM bf16, T={T}")
    only = os.environ.get("MB_ONLY")
    for name, N, K in (("qkv", 3 * H, H), ("dense", H, H), ("h4h", 4 * H, H), ("4hh", H, 4 * H), ("lm_head", V, H)):
        if only and name not in only.split(","):
            continue
        x, w, dy = rnd(T, K), rnd(N, K), rnd(T, N)
        fl = 2.0 * T * N * K
        it = 3 if N == V else 10
        t = timeit(lambda: ops.linear_fwd(x, w, None), it)
        print(f"{name:8s} fwd   M={T} N={N} K={K}: {t:8.3f} ms  {fl / t / 1e9:8.1f} TF/s")
        if os.environ.get("MB_FWD_ONLY"):
            del x, w, dy
            continue
        t = timeit(lambda: ops.linear_dgrad(dy, w), it)
        print(f"{name:8s} dgrad M={T} N={K} K={N}: {t:8.3f} ms  {fl / t / 1e9:8.1f} TF/s")
        t = timeit(lambda: ops.linear_wgrad(dy, x), it)
        print(f"{name:8s} wgrad M={N} N={K} K={T}: {t:8.3f} ms  {fl / t / 1e9:8.1f} TF/s")
        del x, w, dy


def bench_wgroup(T=8192, H=1024):
    """the four weight gradients + two bias column sums of one Bloom block: grouped launch vs the per-product kernels (single stream, warm)"""
    print(f"--- layer weight gradients of one block, T={T} H={H}")
    shapes = [(H, 4 * H, False), (4 * H, H, True), (H, H, False), (3 * H, H, True)]
    probs = [(rnd(T, no), rnd(T, ni), db) for (no, ni, db) in shapes]
    fl = sum(2.0 * T * no * ni for no, ni, _ in shapes)

    def separate():
        for dy, x, db in probs:
            ops.linear_wgrad(dy, x)
            if db:
                ops.colsum(dy)
    t = timeit(separate, 20)
    print(f"per-product (4 GEMMs + split-K reduces + 2 column sums): {t * 1e3:8.1f} us  {fl / t / 1e9:8.1f} TF/s")
    try:
        t = timeit(lambda: ops.wgrad_grouped(probs), 20)
        print(f"grouped (one launch, CTMI_WGRAD_GROUP={os.environ.get('CTMI_WGRAD_GROUP', '1')}):            {t * 1e3:8.1f} us  {fl / t / 1e9:8.1f} TF/s")
    except Exception as ex:                                            # noqa: BLE001
        print("grouped: not available in this library:", str(ex)[:80])


def bench_attn(B=8, S=1024, nh=16, hd=64):
    H = nh * hd
    T = B * S
    qkv, go = rnd(T, 3 * H), rnd(T, H)
    out = torch.empty((T, H), dtype=BF, device=DEV)
    mask = ops.MaskInfo(torch.ones(B, S, dtype=torch.long, device=DEV))
    slopes = alibi_slopes(nh).to(DEV)
    desc = ops.fused_qkv_desc(B, S, nh, hd, True)
    fl = 4.0 * B * nh * S * S * hd / 2                                   # causal-half
    sm, sl = ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask)
    t = timeit(lambda: ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask))
    print(f"--- attention B={B} S={S} nh={nh} hd={hd}\nfwd: {t:8.3f} ms  {fl / t / 1e9:8.1f} TF/s (causal-half flops)")
    dq = torch.empty_like(qkv)
    t = timeit(lambda: ops.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, go, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask))
    print(f"bwd: {t:8.3f} ms  {2.5 * fl / t / 1e9:8.1f} TF/s (2.5x fwd flops)")
    if os.environ.get("MB_KTRACE"):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                ops.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, go, sm, sl, dq, dq[:, hd:], dq[:, 2 * hd:], desc, slopes, mask)
            torch.cuda.synchronize()
        for e in prof.key_averages():
            print(f"   {e.key[:50]:50s} {e.device_time / 1.0:9.1f} us avg")


def bench_ln(T=8192, H=1024):
    x, g = rnd(T, H), rnd(T, H)
    w, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-5)
    t = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-5), 20)
    print(f"--- LayerNorm [{T},{H}] bf16\nfwd: {t * 1e3:8.1f} us  {2 * T * H * 2 / t / 1e6:8.1f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd(g, x, w, mean, rstd, dres=g), 20)
    print(f"bwd: {t * 1e3:8.1f} us  {4 * T * H * 2 / t / 1e6:8.1f} GB/s")
    t = timeit(lambda: ops.colsum(g), 20)
    print(f"colsum [{T},{H}]: {t * 1e3:8.1f} us  {T * H * 2 / t / 1e6:8.1f} GB/s")
    g4 = rnd(T, 4 * H)
    t = timeit(lambda: ops.colsum(g4), 20)
    print(f"colsum [{T},{4 * H}]: {t * 1e3:8.1f} us  {T * 4 * H * 2 / t / 1e6:8.1f} GB/s")


def bench_ce(T=8192, V=250880, S=1024):
    lg = rnd(T, V)
    lab = torch.randint(0, V, (T,), device=DEV)
    loss_out, lse = ops.ce_fwd(lg, lab, seq=S, shift=1)
    t = timeit(lambda: ops.ce_fwd(lg, lab, seq=S, shift=1), 5)
    print(f"--- CE [{T},{V}] bf16\nfwd: {t:8.3f} ms  {T * V * 2 / t / 1e6:8.1f} GB/s")
    d = torch.empty_like(lg)
    t = timeit(lambda: ops.ce_bwd(lg, lab, lse, loss_out, None, seq=S, shift=1, out=d), 5)
    print(f"bwd: {t:8.3f} ms  {2 * T * V * 2 / t / 1e6:8.1f} GB/s")


def bench_adamw(n=559_214_592):
    p = torch.zeros(n, device=DEV)
    g = torch.ones(n, device=DEV) * 1e-3
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh = torch.empty(n, dtype=BF, device=DEV)
    kw = dict(lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, decoupled=True)
    t = timeit(lambda: ops.adamw_step([p], [g], [m], [v], [sh], **kw), 5)
    print(f"--- AdamW {n} params (+bf16 shadow)\nstep: {t:8.3f} ms  {30.0 * n / t / 1e6:8.1f} GB/s (30 B/param)")


def bench_epi(T=8192, H=1024):
    """The epilogue-carrying layer GEMMs: h->4h forward with GELU / GELUG (+ pre-activation or derivative out) and the 4h->h
    data gradient with dGELU / MUL (second [T,4H] operand in), next to their plain forms."""
    x, w1, b1 = rnd(T, H), rnd(4 * H, H), torch.randn(4 * H, device=DEV)
    u = torch.empty(T, 4 * H, dtype=BF, device=DEV)
    fl = 2.0 * T * 4 * H * H
    for name, epi in (("plain+bias", _lib.EPI_NONE), ("GELU", _lib.EPI_GELU), ("GELUG", _lib.EPI_GELUG)):
        t = timeit(lambda: ops.linear_fwd(x, w1, b1, epilogue=epi, aux_out=None if epi == _lib.EPI_NONE else u))
        print(f"h4h fwd  {name:10s}: {t * 1e3:7.1f} us {fl / t / 1e9:8.1f} TF/s")
    dy, w2 = rnd(T, H), rnd(H, 4 * H)
    for name, epi in (("plain", _lib.EPI_NONE), ("DGELU", _lib.EPI_DGELU), ("MUL", _lib.EPI_MUL)):
        t = timeit(lambda: ops.linear_dgrad(dy, w2, epilogue=epi, aux_in=None if epi == _lib.EPI_NONE else u))
        print(f"4hh dgrad {name:9s}: {t * 1e3:7.1f} us {fl / t / 1e9:8.1f} TF/s")
    res = rnd(T, H)
    att, wd, bd = rnd(T, H), rnd(H, H), torch.randn(H, device=DEV)
    t = timeit(lambda: ops.linear_fwd(att, wd, bd, residual=res))
    print(f"dense fwd +res     : {t * 1e3:7.1f} us {2.0 * T * H * H / t / 1e9:8.1f} TF/s")
    g = rnd(T, 4 * H)
    t = timeit(lambda: ops.linear_fwd(g, w2, bd, residual=res))
    print(f"4hh fwd +res       : {t * 1e3:7.1f} us {fl / t / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "ln", "ce", "adamw"]
    if "gemm" in which:
        bench_gemm()
    if "wgroup" in which:
        bench_wgroup()
    if "attn" in which:
        bench_attn()
    if "ln" in which:
        bench_ln()
    if "ce" in which:
        bench_ce()
    if "adamw" in which:
        bench_adamw()
    if "epi" in which:
        bench_epi()
