#!/usr/bin/env python3
"""Per-workgroup cycle anatomy of the LDS-DMA GEMM kernels (variant build -DCTMI_GEMM_TIMING=1): prologue (kernel entry -> first
K-step), K-loops, epilogues, whole lifetime — s_memtime deltas of wave 0 — next to the HIP-event duration of the launch.
Round 5: the instrumentation is no longer in csrc/gemm.hip — apply tools/experiments/gemm_experiment_branches_r04.patch (README.md there says
to which commit) and build the variant in the build container first:
    python -c "from cleantransformer_amd import _build; _build.build_variant('gemmtiming', ['-DCTMI_GEMM_TIMING=1'])"
Usage: python tools/gemm_anatomy.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTMI_LIB_PATH"] = os.path.join(ROOT, "cleantransformer_amd", "lib", "variants", os.environ.get("GEMM_VARIANT", "gemmtiming"), "libctmi355.so")
import torch

from cleantransformer_amd import _lib, ops

DEV, BF = "cuda:0", torch.bfloat16
lib = _lib.load()


def ticks(kind, n):
    fn = getattr(lib, "ctmi_gemm_debug_ticks_" + kind)
    fn.argtypes, fn.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
    buf = (C.c_ulonglong * (n * 4))()
    assert fn(buf, n * 4) == 0
    return torch.tensor(list(buf), dtype=torch.float64).view(n, 4)


def rnd(*s):
    return (torch.randn(*s, device=DEV) * 0.5).to(BF)


def run(name, kind, fn, flops):
    """fn: one callable (warm: the same buffers every launch) or a list of callables over different buffers (cold: what the 24 layers of a
    training step look like to the caches); the tick counters hold the LAST launch."""
    fns = fn if isinstance(fn, (list, tuple)) else [fn]
    for f in fns[:3] if len(fns) > 1 else fns * 3:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10 if len(fns) == 1 else len(fns)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    t = ticks(kind, 256)
    live = t[:, 3] > 0
    t = t[live]
    life = float(t[:, 3].mean())
    clk = float(t[:, 3].max()) / us if us > 0 else 0.0      # lower bound on ticks per us (longest workgroup / launch time)
    fn2 = getattr(lib, "ctmi_gemm_debug_ticks_" + kind + "_phase")
    fn2.argtypes, fn2.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
    pb = (C.c_ulonglong * 2048)()
    fn2(pb, 2048)
    ph = torch.tensor(list(pb), dtype=torch.float64).view(128, 2, 8)
    steps = float(t[:, 1].mean()) and max(1.0, flops / (2.0 * 256 * 256 * 32) / 256)
    print(f"{name:34s} {us:7.1f} us {flops / us / 1e6:7.1f} TF/s | workgroups {int(live.sum()):4d}  cycles: prologue {float(t[:, 0].mean()):7.0f}  K-loops {float(t[:, 1].mean()):8.0f}  "
          f"epilogues {float(t[:, 2].mean()):7.0f}  lifetime mean {life:8.0f} max {float(t[:, 3].max()):8.0f}  (>= {clk:5.0f} ticks/us)", flush=True)
    for w, nm in ((0, "leading group, wave 0 "), (1, "lagging group, last wave")):
        v = ph[:, w, :5].mean(0)
        tot = float(v.sum())
        if tot > 0:
            print(f"      {nm}: reads+DMA issue {float(v[0]):8.0f}  vmcnt/lgkm wait {float(v[1]):8.0f}  barrier-1 {float(v[2]):8.0f}  MFMA phase {float(v[3]):8.0f}  barrier-2 {float(v[4]):8.0f}   (sum {tot:8.0f})")


T, H = 8192, 1024
x, dy4, dy3, dyh = rnd(T, H), rnd(T, 4 * H), rnd(T, 3 * H), rnd(T, H)
wq, wd, w1, w2 = rnd(3 * H, H), rnd(H, H), rnd(4 * H, H), rnd(H, 4 * H)
b1 = torch.randn(4 * H, device=DEV)
u = torch.empty(T, 4 * H, dtype=BF, device=DEV)
res = rnd(T, H)
run("qkv fwd   [T,1024]x[3072,1024]^T", "nt", lambda: ops.linear_fwd(x, wq, None), 2.0 * T * 3 * H * H)
run("dense fwd [T,1024]x[1024,1024]^T +res", "nt", lambda: ops.linear_fwd(x, wd, None, residual=res), 2.0 * T * H * H)
run("h4h fwd + GELU", "nt", lambda: ops.linear_fwd(x, w1, b1, epilogue=_lib.EPI_GELU, aux_out=u), 2.0 * T * 4 * H * H)
run("h4h fwd plain", "nt", lambda: ops.linear_fwd(x, w1, b1), 2.0 * T * 4 * H * H)
run("4hh fwd K=4096 +res", "nt", lambda: ops.linear_fwd(dy4, w2, None, residual=res), 2.0 * T * 4 * H * H)
run("4hh dgrad DGELU [T,1024]x[1024,4096]", "nn", lambda: ops.linear_dgrad(dyh, w2, epilogue=_lib.EPI_DGELU, aux_in=u), 2.0 * T * 4 * H * H)
run("4hh dgrad plain", "nn", lambda: ops.linear_dgrad(dyh, w2), 2.0 * T * 4 * H * H)
run("h4h dgrad K=4096", "nn", lambda: ops.linear_dgrad(dy4, w1), 2.0 * T * 4 * H * H)
run("dense dgrad", "nn", lambda: ops.linear_dgrad(dyh, wd), 2.0 * T * H * H)
run("qkv dgrad K=3072", "nn", lambda: ops.linear_dgrad(dy3, wq), 2.0 * T * 3 * H * H)
V = 250880
wv = rnd(V, H)
run("lm_head fwd", "nt", lambda: ops.linear_fwd(x, wv, None), 2.0 * T * V * H)

if os.environ.get("ANATOMY_COLD", "1") != "0":
    L = 24
    print("--- cold: a different input, weight and output buffer per launch (24 sets), counters of the last launch")
    xs, outs3, outs4, outs1 = [rnd(T, H) for _ in range(L)], [torch.empty(T, 3 * H, dtype=BF, device=DEV) for _ in range(L)], \
        [torch.empty(T, 4 * H, dtype=BF, device=DEV) for _ in range(L)], [torch.empty(T, H, dtype=BF, device=DEV) for _ in range(L)]
    wqs, w1s, w2s = [rnd(3 * H, H) for _ in range(L)], [rnd(4 * H, H) for _ in range(L)], [rnd(H, 4 * H) for _ in range(L)]
    g4s = [rnd(T, 4 * H) for _ in range(L)]
    run("qkv fwd (cold)", "nt", [(lambda i=i: ops.linear_fwd(xs[i], wqs[i], None, out=outs3[i])) for i in range(L)], 2.0 * T * 3 * H * H)
    run("h4h fwd plain (cold)", "nt", [(lambda i=i: ops.linear_fwd(xs[i], w1s[i], b1, out=outs4[i])) for i in range(L)], 2.0 * T * 4 * H * H)
    run("4hh fwd K=4096 +res (cold)", "nt", [(lambda i=i: ops.linear_fwd(g4s[i], w2s[i], None, residual=xs[i], out=outs1[i])) for i in range(L)], 2.0 * T * 4 * H * H)
    run("4hh dgrad plain (cold)", "nn", [(lambda i=i: ops.linear_dgrad(xs[i], w2s[i])) for i in range(L)], 2.0 * T * 4 * H * H)
    run("h4h dgrad K=4096 (cold)", "nn", [(lambda i=i: ops.linear_dgrad(g4s[i], w1s[i])) for i in range(L)], 2.0 * T * 4 * H * H)
