#!/usr/bin/env python3
"""Per-tile overhead of the 256x256 forward tile: time(K) = a + b*K at a fixed [T, N] output (LM-head-like, bf16 logits).
a / tiles-per-CU = what a tile costs outside its K-loop (epilogue, ring refill); 1/b = the K-loop's own rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleantransformer_amd import ops
from tools.microbench import timeit, rnd

T = 8192
NS = tuple(int(v) for v in os.environ['KS_N'].split(',')) if os.environ.get('KS_N') else ((250880,) if os.environ.get('KS_LM') else (250880, 4096))
for N in NS:
    pts = []
    for K in ((1024, 4096) if os.environ.get('KS_LM') else (256, 512, 1024, 2048, 4096)):
        x, w = rnd(T, K), rnd(N, K)
        t = timeit(lambda: ops.linear_fwd(x, w, None), 5 if N > 100000 else 20)
        fl = 2.0 * T * N * K
        pts.append((K, t))
        print(f"N={N} K={K}: {t * 1e3:9.1f} us  {fl / t / 1e9:8.1f} TF/s")
        del x, w
    (k0, t0), (k1, t1) = pts[0], pts[-1]
    b = (t1 - t0) / (k1 - k0); a = t0 - b * k0
    tiles = (T // 256) * ((N + 255) // 256)
    print(f"N={N}: a = {a * 1e3:.1f} us, K-loop rate = {2.0 * T * N / b / 1e9:.0f} TF/s, tiles per CU = {tiles / 256:.1f}, overhead per tile = {a * 1e3 / (tiles / 256):.2f} us")
