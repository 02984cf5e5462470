#!/usr/bin/env python3
"""Reference point only (NOT used by the product): what the vendor GEMM library (hipBLASLt through torch.matmul) reaches on the
Bloom-560M shapes, next to tools/microbench.py's numbers for the hand-written kernels."""
import torch
DEV = "cuda:0"
BF = torch.bfloat16

def timeit(fn, iters=10, warm=3):
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

T, H, V = 8192, 1024, 250880
for name, N, K in (("qkv", 3 * H, H), ("dense", H, H), ("h4h", 4 * H, H), ("4hh", H, 4 * H), ("lm_head", V, H)):
    x = (torch.randn(T, K, device=DEV) * 0.5).to(BF)
    w = (torch.randn(N, K, device=DEV) * 0.5).to(BF)
    dy = (torch.randn(T, N, device=DEV) * 0.5).to(BF)
    fl = 2.0 * T * N * K
    it = 3 if N == V else 10
    t = timeit(lambda: torch.matmul(x, w.t()), it)
    print(f"{name:8s} fwd   {t:8.3f} ms {fl / t / 1e9:8.1f} TF/s")
    t = timeit(lambda: torch.matmul(dy, w), it)
    print(f"{name:8s} dgrad {t:8.3f} ms {fl / t / 1e9:8.1f} TF/s")
    t = timeit(lambda: torch.matmul(dy.t(), x), it)
    print(f"{name:8s} wgrad {t:8.3f} ms {fl / t / 1e9:8.1f} TF/s (bf16 out)")
    del x, w, dy
