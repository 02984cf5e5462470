#!/usr/bin/env python3
"""Per-wave cycle anatomy of the 256-row attention forward (variant build -DCTMI_W32_TIMING=1): matrix segments, vector segments and
the barrier waits behind each, from s_memtime deltas the kernel dumps over stat_l.  Usage: python tools/attn_w32_timing.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the binding reads CTMI_LIB_PATH when the package is imported: point it at the variant first (built by
#   python -c "from cleantransformer_amd import _build; _build.build_variant('w32timing', ['-DCTMI_W32_TIMING=1'])"  in the build container)
os.environ["CTMI_LIB_PATH"] = os.path.join(ROOT, "cleantransformer_amd", "lib", "variants", os.environ.get("W32_VARIANT", "w32timing"), "libctmi355.so")
import torch

from cleantransformer_amd import ops
from cleantransformer_amd.models.modeling_bloom import alibi_slopes

DEV, BF = "cuda:0", torch.bfloat16
B, S, nh, hd = 8, 1024, 16, 64
H, T = nh * hd, B * S
qkv = (torch.randn(T, 3 * H, device=DEV) * 0.5).to(BF)
mask = ops.MaskInfo(torch.ones(B, S, dtype=torch.long, device=DEV))
slopes = alibi_slopes(nh).to(DEV)
desc = ops.fused_qkv_desc(B, S, nh, hd, causal=True)
out = torch.empty((T, H), dtype=BF, device=DEV)
ops.set_attn_path(3)
for _ in range(3):
    sm, sl = ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask)
torch.cuda.synchronize()
v = sl.view(B * nh, S // 32, 32)[:, :, :8].cpu()          # [bh, 32-row block, 8 values]
names = ["X (matrix seg)", "Y (vector seg)", "barrier after X", "barrier after Y", "prologue", "total", "ntiles", "my tiles"]
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, desc, slopes, mask)
torch.cuda.synchronize()
print(f"variant {os.environ.get('W32_VARIANT', 'w32timing')}: forward {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch (instrumented)")
if os.environ.get("W32_BRIEF"):
    tot = v.mean(0); work = tot[:, 7].sum()
    print("per own tile: X %.0f  Y %.0f  barX %.0f  barY %.0f" % tuple(float(tot[:, i].sum() / work) for i in range(4)))
    sys.exit(0)
print(f"B={B} S={S} nh={nh} hd={hd}; cycles per wave (s_memtime), by 32-row block of the sequence (mean over batch x heads)")
print("block " + " ".join(f"{n:>16s}" for n in names))
for rb in range(S // 32):
    print(f"{rb:5d} " + " ".join(f"{float(v[:, rb, i].mean()):16.0f}" for i in range(8)))
tot = v.mean(0)
work = tot[:, 7].sum()
print("per own tile: X %.0f  Y %.0f  barX %.0f  barY %.0f" % tuple(float(tot[:, i].sum() / work) for i in range(4)))
print("sum over blocks / 32 blocks: " + " ".join(f"{float(tot[:, i].mean()):.0f}" for i in range(6)))
