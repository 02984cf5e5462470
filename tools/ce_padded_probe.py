#!/usr/bin/env python3
"""fused loss at a padded row pitch (GPT-2: 50257 classes in rows of 50272) vs the two-pass form: time of each piece"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleantransformer_amd import ops

dev = "cuda:0"
N, V, S = 8192, 50257, 2048
Vp = ops.pad_rows(V)
buf = (torch.randn(N, Vp, device=dev) * 2).to(torch.bfloat16)
l2 = buf[:, :V]
lab = torch.randint(0, V, (N // S, S), device=dev)


def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


print("ce_fused_ok:", ops.ce_fused_ok(l2), "pitch", l2.stride(0))
print("fused      : %.3f ms" % t(lambda: ops.ce_fwd_bwd(l2, lab, seq=S, shift=1)))
lo, lse = ops.ce_fwd(l2, lab, seq=S, shift=1, ignore_index=-100, denom_mode=0)
g = torch.ones(1, device=dev)
print("fwd        : %.3f ms" % t(lambda: ops.ce_fwd(l2, lab, seq=S, shift=1, ignore_index=-100, denom_mode=0)))
print("bwd        : %.3f ms" % t(lambda: ops.ce_bwd(l2, lab, lse, lo, g, seq=S, shift=1, ignore_index=-100)))
d = torch.empty(N, Vp, dtype=torch.bfloat16, device=dev)
print("pad zero   : %.3f ms" % t(lambda: d[:, V:].zero_()))
dense = l2.contiguous()[:, :50256]
print("fused dense 50256: %.3f ms" % t(lambda: ops.ce_fwd_bwd(dense.contiguous(), lab, seq=S, shift=1)))
