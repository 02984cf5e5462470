#!/usr/bin/env python3
"""Why is a layer GEMM ~9 us slower inside the training step than in tools/microbench.py?  (r03: qkv fwd 69.4 vs 60.0 us, +res
forwards 60.3 vs 50.5, GELU h->4h 84.7 vs 77.)  The microbenchmark re-launches ONE problem on the same buffers: operands sit in L2 /
Infinity Cache and the TLB is hot.  The step runs 24 layers with their own weights and activations.  This probe times each layer shape
as a sequence of L launches in four regimes:
    warm      same x, w, out every launch                         (= tools/microbench.py)
    cold_w    a different weight matrix per launch
    cold_a    a different input / output activation per launch
    cold      both (what the step does)
and the forward chain LN -> qkv -> dense(+res) -> LN -> h4h(+GELU) -> 4hh(+res) over L layers with per-layer buffers, bracketed by
HIP events.  Also prints the shader clock under MFMA load (ctmi_clock_probe) before and after.
Usage: python tools/chain_probe.py [L=24]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cleantransformer_amd import _lib, ops

DEV = "cuda:0"
BF = torch.bfloat16
T, H = 8192, 1024


def rnd(*s, dtype=BF):
    return (torch.randn(*s, device=DEV) * 0.5).to(dtype)


def clock_mhz(iters=20000):
    return ops.clock_probe(DEV, iters)


def time_seq(fns, reps=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))            # us per launch


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    print(f"shader clock under MFMA load, idle chip: {clock_mhz():.0f} MHz")
    shapes = (("qkv fwd", 3 * H, H, False), ("dense fwd+res", H, H, True), ("h4h fwd", 4 * H, H, False), ("4hh fwd+res", H, 4 * H, True))
    for name, N, K, res in shapes:
        xs = [rnd(T, K) for _ in range(L)]
        ws = [rnd(N, K) for _ in range(L)]
        outs = [torch.empty(T, N, dtype=BF, device=DEV) for _ in range(L)]
        rs = [rnd(T, N) for _ in range(L)] if res else [None] * L
        b = torch.randn(N, device=DEV)

        def mk(ix, iw, io):
            return lambda: ops.linear_fwd(xs[ix], ws[iw], b, residual=rs[io], out=outs[io])
        warm = time_seq([mk(0, 0, 0) for _ in range(L)])
        cold_w = time_seq([mk(0, i, 0) for i in range(L)])
        cold_a = time_seq([mk(i, 0, i) for i in range(L)])
        cold = time_seq([mk(i, i, i) for i in range(L)])
        fl = 2.0 * T * N * K
        print(f"{name:14s} us/launch: warm {warm:6.1f}  cold_w {cold_w:6.1f}  cold_a {cold_a:6.1f}  cold {cold:6.1f}   ({fl / warm / 1e6:6.0f} -> {fl / cold / 1e6:6.0f} TF/s)")
        del xs, ws, outs, rs
    # data-gradient shapes (K-major weight)
    for name, N, K in (("qkv dgrad", H, 3 * H), ("dense dgrad", H, H), ("h4h dgrad", H, 4 * H), ("4hh dgrad", 4 * H, H)):
        dys = [rnd(T, K) for _ in range(L)]
        ws = [rnd(K, N) for _ in range(L)]
        warm = time_seq([(lambda: ops.linear_dgrad(dys[0], ws[0])) for _ in range(L)])
        cold = time_seq([(lambda i=i: ops.linear_dgrad(dys[i], ws[i])) for i in range(L)])
        fl = 2.0 * T * N * K
        print(f"{name:14s} us/launch: warm {warm:6.1f}  cold {cold:6.1f}   ({fl / warm / 1e6:6.0f} -> {fl / cold / 1e6:6.0f} TF/s)")
        del dys, ws
    # the forward chain of a block, per-layer buffers
    lnw, lnb = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    P = [dict(wqkv=rnd(3 * H, H), wd=rnd(H, H), w1=rnd(4 * H, H), w2=rnd(H, 4 * H), bq=torch.randn(3 * H, device=DEV), bd=torch.randn(H, device=DEV),
              b1=torch.randn(4 * H, device=DEV), b2=torch.randn(H, device=DEV), qkv=torch.empty(T, 3 * H, dtype=BF, device=DEV),
              h1=torch.empty(T, H, dtype=BF, device=DEV), u=torch.empty(T, 4 * H, dtype=BF, device=DEV), g=torch.empty(T, 4 * H, dtype=BF, device=DEV),
              out=torch.empty(T, H, dtype=BF, device=DEV)) for _ in range(L)]
    x0 = rnd(T, H)

    def chain(gemm_only):
        x = x0
        for p in P:
            ln1 = x if gemm_only else ops.layernorm_fwd(x, lnw, lnb, 1e-5)[0]
            ops.linear_fwd(ln1, p["wqkv"], p["bq"], out=p["qkv"])
            ops.linear_fwd(ln1, p["wd"], p["bd"], residual=x, out=p["h1"])              # (ln1 stands in for the attention output)
            ln2 = p["h1"] if gemm_only else ops.layernorm_fwd(p["h1"], lnw, lnb, 1e-5)[0]
            ops.linear_fwd(ln2, p["w1"], p["b1"], epilogue=_lib.EPI_GELU, aux_out=p["u"], out=p["g"])
            ops.linear_fwd(p["g"], p["w2"], p["b2"], residual=p["h1"], out=p["out"])
            x = p["out"]
    for gemm_only in (True, False):
        chain(gemm_only)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            chain(gemm_only)
        e1.record()
        torch.cuda.synchronize()
        print(f"forward chain of {L} blocks ({'GEMMs only' if gemm_only else 'LayerNorm + GEMMs'}), per-layer buffers: {e0.elapsed_time(e1) / 3:7.3f} ms  = {e0.elapsed_time(e1) * 1e3 / 3 / L:6.1f} us per block")
    print(f"shader clock under MFMA load, right after the chain: {clock_mhz():.0f} MHz")


if __name__ == "__main__":
    main()
