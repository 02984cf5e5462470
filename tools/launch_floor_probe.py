#!/usr/bin/env python3
"""What does one DEPENDENT launch cost on the GPU's side?  A chain of N launches of a kernel that does (almost) nothing, in one stream — the queue
is kept full (the host runs ahead), so HIP-event time / N is the device-side floor per launch: completion of the previous kernel, cache
write-back / invalidate, dispatch of the next.  Same chain replayed from a hipGraph (torch.cuda.CUDAGraph).  And the library's LayerNorm forward
on [8192,1024] bf16 as a real small kernel: time per launch in a chain vs the same rows in ONE launch of 16x the rows.
usage: python tools/launch_floor_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleantransformer_amd import ops

DEV = "cuda:0"


def ev_time(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    N = 2000
    x = torch.zeros(64, device=DEV)

    def chain():
        for _ in range(N):
            x.add_(1.0)
    t = ev_time(chain)
    print(f"torch x.add_(1) on 64 floats, {N} launches in one stream: {t / N * 1e3:.2f} us per launch (host-bound if the host cannot keep up)")
    # the library's own trivial launch through ctypes
    s = torch.ones(1, device=DEV)
    y = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)

    def chain2():
        for _ in range(N):
            ops.scale_if_(y, s, 1.0)
    t = ev_time(chain2)
    print(f"ctmi_scale_if (returns after one scalar load), {N} launches: {t / N * 1e3:.2f} us per launch")
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        chain()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        chain()
    t = ev_time(lambda: g.replay())
    print(f"the same {N} torch launches replayed from a hipGraph: {t / N * 1e3:.2f} us per launch")
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        chain2()
    t = ev_time(lambda: g2.replay())
    print(f"the same {N} ctmi_scale_if launches replayed from a hipGraph: {t / N * 1e3:.2f} us per launch")
    # a real small kernel
    rows, H = 8192, 1024
    xs = [torch.randn(rows, H, device=DEV).to(torch.bfloat16) for _ in range(16)]
    w, b = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    big = torch.cat(xs)

    def ln_chain():
        for xi in xs:
            ops.layernorm_fwd(xi, w, b, 1e-5)
    t16 = ev_time(ln_chain)
    t1 = ev_time(lambda: ops.layernorm_fwd(big, w, b, 1e-5))
    print(f"LayerNorm fwd [8192,1024] bf16: 16 launches {t16 / 16 * 1e3:.2f} us each; ONE launch over 16x the rows {t1 * 1e3:.1f} us = {t1 / 16 * 1e3:.2f} us per 8192 rows "
          f"-> per-launch overhead ~{(t16 - t1) / 16 * 1e3:.2f} us")
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        ln_chain()
    t16g = ev_time(lambda: g3.replay())
    print(f"the 16 LayerNorm launches from a hipGraph: {t16g / 16 * 1e3:.2f} us each")


if __name__ == "__main__":
    main()
