#!/usr/bin/env python3
"""Joules per launch (and per useful TFLOP / GB) of the hot kernels, from package power sampled beside a steady loop of launches.

Why (round 4): the training step runs at ~1.3 kW of a 1.4 kW cap and the firmware pays for it with the shader clock (~2.09 of 2.4 GHz;
profiles/r04_power_samples.txt) — on this workload a kernel is worth what it costs in ENERGY, not only in cycles: a variant that is 3 % faster
alone and draws 5 % more gives the step nothing.  This tool puts a number on that: each workload is launched back to back for SECONDS
(default 1.5 s) while a thread samples `rocm-smi --showpower`; energy per launch = mean package power over the samples taken inside the
loop x time per launch; "dynamic" subtracts the idle draw measured first.  Operand values matter (switching activity): operands are
N(0, 0.5) in bf16 like the microbenchmarks'; `--zeros` repeats every workload on all-zero operands as the floor.

    python tools/energy_probe.py [gemm] [vendor] [wgrad] [wgroup] [attn] [hbm] [--seconds 1.5] [--zeros]

gemm: the step's GEMM shapes through the library (tile choice as in the step)      vendor: the same products through torch.matmul (hipBLASLt)
attn: attention forward / backward at B=8 S=1024 nh=16 hd=64                        hbm: AdamW, fused cross entropy, LayerNorm backward
"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from cleantransformer_amd import ops

DEV = "cuda:0"
BF = torch.bfloat16
SMI = "/opt/rocm/bin/rocm-smi"


def smi_power():
    try:
        r = subprocess.run([SMI, "-d", "0", "--showpower"], capture_output=True, text=True, timeout=10)
        m = re.search(r"Package Power \(W\): ([0-9.]+)", r.stdout)
        return float(m.group(1)) if m else None
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            p = smi_power()
            if p is not None:
                self.samples.append((time.perf_counter(), p))


def measure(name, fn, seconds, work=None, unit="TFLOP", idle_w=0.0):
    """fn() enqueues one launch (or a fixed group); returns a dict and prints a line."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t_est = e0.elapsed_time(e1) / 5 * 1e-3
    n = max(10, int(seconds / t_est))
    s = Sampler()
    s.start()
    time.sleep(0.15)
    t0 = time.perf_counter()
    e0.record()
    done = 0
    while done < n:                                                     # enqueue in slices so that the queue never runs dry nor a minute ahead
        k = min(n - done, max(1, int(0.05 / t_est)))
        for _ in range(k):
            fn()
        done += k
        if done * t_est - (time.perf_counter() - t0) > 0.3:
            time.sleep(0.1)
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s.stop = True
    s.join()
    per = e0.elapsed_time(e1) / n * 1e-3
    inside = [p for (t, p) in s.samples if t0 + 0.3 <= t <= t1 - 0.05]    # SMI averages over a window: skip the ramp
    if len(inside) < 3:
        print(f"{name:34s} {per * 1e6:9.1f} us/launch   (only {len(inside)} power samples: run longer)")
        return None
    pw = sum(inside) / len(inside)
    j, jd = pw * per, (pw - idle_w) * per
    line = f"{name:34s} {per * 1e6:9.1f} us  {pw:6.0f} W ({len(inside):2d} samples)  {j * 1e3:8.2f} mJ/launch  dynamic {jd * 1e3:8.2f} mJ"
    if work:
        line += f"   {work / per / 1e12 if unit == 'TFLOP' else work / per / 1e12:7.1f} {'TF/s' if unit == 'TFLOP' else 'TB/s'}   {j / (work / 1e12):6.2f} J/{unit}  dynamic {jd / (work / 1e12):6.2f}"
    print(line, flush=True)
    return {"name": name, "us": per * 1e6, "watts": pw, "mJ": j * 1e3}


def rnd(*s, zeros=False, dtype=BF):
    return torch.zeros(*s, device=DEV, dtype=dtype) if zeros else (torch.randn(*s, device=DEV) * 0.5).to(dtype)


def main(argv):
    seconds = 1.5
    if "--seconds" in argv:
        seconds = float(argv[argv.index("--seconds") + 1])
    which = [a for a in argv if not a.startswith("--") and not a.replace(".", "").isdigit()] or ["gemm", "vendor", "attn", "hbm"]
    modes = [False, True] if "--zeros" in argv else [False]
    torch.cuda.synchronize()
    time.sleep(1.0)
    idle = [smi_power() for _ in range(5)]
    idle = [p for p in idle if p is not None]
    idle_w = sum(idle) / len(idle) if idle else 0.0
    print(f"idle package power: {idle_w:.0f} W   (loop length {seconds} s per workload)")
    T, H, V = 8192, 1024, 250880
    for zeros in modes:
        tag = " [zero operands]" if zeros else ""
        if "gemm" in which or "vendor" in which or "wgrad" in which:
            only = os.environ.get("EP_ONLY")                               # e.g. EP_ONLY=4hh,h4h
            for name, N, K in (("qkv", 3 * H, H), ("h4h", 4 * H, H), ("4hh", H, 4 * H), ("lm_head", V, H)):
                if only and name not in only.split(","):
                    continue
                x, w, dy = rnd(T, K, zeros=zeros), rnd(N, K, zeros=zeros), rnd(T, N, zeros=zeros)
                fl = 2.0 * T * N * K
                if "wgrad" in which:                                     # weight gradients only (tile / split rules come from the environment: CTMI_WGRAD_RULE, ...)
                    if name != "lm_head":
                        measure(f"{name} wgrad{tag}", lambda: ops.linear_wgrad(dy, x), seconds, fl, idle_w=idle_w)
                if "gemm" in which:
                    measure(f"{name} fwd{tag}", lambda: ops.linear_fwd(x, w, None), seconds, fl, idle_w=idle_w)
                    measure(f"{name} dgrad{tag}", lambda: ops.linear_dgrad(dy, w), seconds, fl, idle_w=idle_w)
                    measure(f"{name} wgrad{tag}", lambda: ops.linear_wgrad(dy, x), seconds, fl, idle_w=idle_w)
                if "vendor" in which:
                    wt = w.t()
                    out = torch.empty(T, N, device=DEV, dtype=BF)
                    measure(f"{name} fwd  hipBLASLt{tag}", lambda: torch.matmul(x, wt, out=out), seconds, fl, idle_w=idle_w)
                    dx = torch.empty(T, K, device=DEV, dtype=BF)
                    measure(f"{name} dgrad hipBLASLt{tag}", lambda: torch.matmul(dy, w, out=dx), seconds, fl, idle_w=idle_w)
                    del out, dx
                del x, w, dy
        if "wgroup" in which:                                           # the four weight gradients + two bias column sums of one block (round 5)
            shapes = [(H, 4 * H, False), (4 * H, H, True), (H, H, False), (3 * H, H, True)]
            probs = [(rnd(T, no, zeros=zeros), rnd(T, ni, zeros=zeros), db) for (no, ni, db) in shapes]
            fl = sum(2.0 * T * no * ni for no, ni, _ in shapes)

            def separate():
                for dy, x, db in probs:
                    ops.linear_wgrad(dy, x)
                    if db:
                        ops.colsum(dy)
            measure(f"block wgrads, per product{tag}", separate, seconds, fl, idle_w=idle_w)
            measure(f"block wgrads, grouped launch{tag}", lambda: ops.wgrad_grouped(probs), seconds, fl, idle_w=idle_w)
            del probs
        if "attn" in which:
            from cleantransformer_amd.models.modeling_bloom import alibi_slopes
            B, S, nh, hd = 8, 1024, 16, 64
            qkv, go = rnd(B * S, 3 * nh * hd, zeros=zeros), rnd(B * S, nh * hd, zeros=zeros)
            att = torch.empty(B * S, nh * hd, dtype=BF, device=DEV)
            desc = ops.fused_qkv_desc(B, S, nh, hd, True)
            sl = alibi_slopes(nh).to(DEV)
            mask = ops.MaskInfo(torch.ones(B, S, dtype=torch.long, device=DEV))
            m, l = ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], att, desc, sl, mask)
            fl = 4.0 * B * nh * S * S * hd / 2
            measure(f"attention fwd{tag}", lambda: ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], att, desc, sl, mask), seconds, fl, idle_w=idle_w)
            dqkv = torch.empty_like(qkv)
            measure(f"attention bwd{tag}", lambda: ops.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], att, go, m, l, dqkv, dqkv[:, hd:], dqkv[:, 2 * hd:], desc, sl, mask),
                    seconds, 2.5 * fl, idle_w=idle_w)
        if "hbm" in which and not zeros:
            n = 559214592 // 4
            p, g = torch.randn(n, device=DEV), torch.randn(n, device=DEV) * 1e-3
            mm, vv, sh = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.empty(n, device=DEV, dtype=BF)
            measure("AdamW (140 M params)", lambda: ops.adamw_step([p], [g], [mm], [vv], [sh], lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01,
                                                                   step=3, decoupled=True), seconds, n * 30.0, unit="TB", idle_w=idle_w)
            del p, g, mm, vv, sh
            xx, gg = rnd(T, H), rnd(T, H)
            w, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
            y, mean, rstd = ops.layernorm_fwd(xx, w, b, 1e-5)
            measure("LayerNorm bwd [8192,1024]", lambda: ops.layernorm_bwd(gg, xx, w, mean, rstd, dres=gg), seconds, 4.0 * T * H * 2, unit="TB", idle_w=idle_w)


if __name__ == "__main__":
    main(sys.argv[1:])
