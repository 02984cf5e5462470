#!/usr/bin/env python3
"""AdamW HBM-rate probe: does the RELATIVE placement of the four fp32 streams (p, g, m, v) of one large tensor matter?  torch's caching allocator
hands out 2 MiB-aligned blocks, so element i of all four arrays sits at the same offset inside its block — the same channel and bank of the
HBM address map, four different rows.  This probe carves the arrays out of one slab with chosen byte offsets between them and times the fused
optimizer (ctmi_adamw_step) on the tied-table size of Bloom-560M (250880 x 1024).
usage: python tools/adamw_probe.py [n_elements]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleantransformer_amd import ops

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250880 * 1024
SLACK = 64 << 20


def timeit(fn, iters=8, warm=2):
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


def carve(slab, idx, off_bytes):
    per = N * 4 + SLACK
    start = idx * per + off_bytes
    assert start % 16 == 0
    return slab[start:start + N * 4].view(torch.float32)


def main():
    slab = torch.zeros(4 * (N * 4 + SLACK), dtype=torch.uint8, device=DEV)
    sh_slab = torch.zeros(N * 2 + SLACK, dtype=torch.uint8, device=DEV)
    print(f"N = {N} ({N * 4 / 2**30:.2f} GiB per fp32 stream); slab base % 2MiB = {slab.data_ptr() % (2 << 20)}")
    KB, MB = 1 << 10, 1 << 20
    variants = [("aligned (0,0,0,0)", (0, 0, 0, 0)), ("2 KiB steps", (0, 2 * KB, 4 * KB, 6 * KB)), ("4 KiB steps", (0, 4 * KB, 8 * KB, 12 * KB)),
                ("8 KiB steps", (0, 8 * KB, 16 * KB, 24 * KB)), ("p,g aligned; m +4K, v +8K", (0, 0, 4 * KB, 8 * KB)),
                ("p,g aligned; m +4K, v +12K", (0, 0, 4 * KB, 12 * KB)), ("p 0, g +8K, m +4K, v +12K", (0, 8 * KB, 4 * KB, 12 * KB)),
                ("4 KiB steps, shadow +2K", (0, 4 * KB, 8 * KB, 12 * KB, 2 * KB)), ("4 KiB steps, shadow +16K", (0, 4 * KB, 8 * KB, 12 * KB, 16 * KB)),
                ("1 MiB + 4 KiB steps", (0, MB + 4 * KB, 2 * MB + 8 * KB, 3 * MB + 12 * KB)), ("aligned again", (0, 0, 0, 0))]
    for name, offs in variants:
        p, g, m, v = (carve(slab, i, o) for i, o in enumerate(offs[:4]))
        sho = offs[4] if len(offs) > 4 else 0
        g.normal_(); p.normal_(std=0.02); m.zero_(); v.zero_()
        sh = sh_slab[sho:sho + N * 2].view(torch.bfloat16)
        step = [0]

        def fn():
            step[0] += 1
            ops.adamw_step([p], [g], [m], [v], [sh], lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=step[0], decoupled=True)
        ms = timeit(fn)
        print(f"{name:38s} {ms * 1e3:8.1f} us   {N * 30 / ms / 1e9:7.1f} TB/s", flush=True)


if __name__ == "__main__":
    main()
