#!/usr/bin/env python3
"""Randomised parity sweep of the 128-row attention kernels (csrc/attention_w32.hip) against the general kernels and the fp32 CPU restatement
of tools/attn_w32_check.py: random batch / heads / sequence length (multiples of 64) / head size / padding pattern / fill kind.
usage: python tools/attn_w32_fuzz.py [cases=40] [seed=0]"""
import os
import sys
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import attn_w32_check as C
from cleantransformer_amd import ops
from cleantransformer_amd.models.modeling_bloom import alibi_slopes

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
bad = 0
for case in range(n):
    B, nh, hd = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3]), rnd.choice([64, 64, 128])
    S = 64 * rnd.randint(1, 12 if hd == 64 else 6)
    fill = rnd.choice([0.0, 0.0, -1e4])
    am = torch.ones(B, S, dtype=torch.long)
    for b in range(B):
        kind = rnd.choice(["none", "left", "right", "both", "holes", "onlylast"])
        if kind in ("left", "both"):
            am[b, :rnd.randint(1, max(1, S - 2))] = 0
        if kind in ("right", "both"):
            am[b, S - rnd.randint(1, S // 2):] = 0
        if kind == "holes":
            am[b, rnd.randint(0, 6)::rnd.randint(2, 9)] = 0
        if kind == "onlylast":
            am[b, :S - 1] = 0
        if int(am[b].sum()) == 0:
            am[b, rnd.randint(0, S - 1)] = 1
    torch.manual_seed(seed * 1000 + case)
    H = nh * hd
    qkv = (torch.randn(B, S, 3 * H) * rnd.choice([0.3, 0.7, 1.5])).to(C.BF)
    go = (torch.randn(B, S, H) * 0.5).to(C.BF)
    mask = ops.MaskInfo(am.to(C.DEV))
    slopes = alibi_slopes(nh).to(C.DEV)
    qd, god = qkv.reshape(B * S, 3 * H).to(C.DEV), go.reshape(B * S, H).to(C.DEV)
    o_ref, g_ref = C.cpu_ref(qkv, go, am, nh, hd, C.FMIN if fill == 0.0 else fill)
    o0, m0, l0, g0 = C.run(0, qd, god, B, S, nh, hd, mask, slopes, fill)
    o1, m1, l1, g1 = C.run(3, qd, god, B, S, nh, hd, mask, slopes, fill)
    e_old = (C.err(o0.view(B, S, H), o_ref), C.err(g0.view(B, S, 3 * H), g_ref))
    e_new = (C.err(o1.view(B, S, H), o_ref), C.err(g1.view(B, S, 3 * H), g_ref))
    fin = m0 > C.FMIN / 2
    e_m = float((m1 - m0)[fin].abs().max()) if fin.any() else 0.0
    same_min = bool(((m1 <= C.FMIN) == (m0 <= C.FMIN)).all())
    e_l = float(((l1 - l0).abs() / l0).max())
    ok = (e_new[0][1] <= max(2.5 * e_old[0][1], 8e-3) and e_new[1][1] <= max(2.5 * e_old[1][1], 1.6e-2) and e_m < 2e-3 and e_l < 2e-3 and same_min
          and bool(torch.isfinite(o1).all()) and bool(torch.isfinite(g1).all()))
    bad += 0 if ok else 1
    print(f"{'OK ' if ok else 'BAD'} B={B} S={S} nh={nh} hd={hd} fill={fill:g} valid={[int(v) for v in am.sum(1)]}: out {e_new[0][1]:.2e}/{e_old[0][1]:.2e} "
          f"dqkv {e_new[1][1]:.2e}/{e_old[1][1]:.2e} m {e_m:.1e} l {e_l:.1e} {'same' if same_min else 'DIFF'}", flush=True)
print("attn_w32 fuzz:", "ALL OK" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
