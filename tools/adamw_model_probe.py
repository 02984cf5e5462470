#!/usr/bin/env python3
"""AdamW on the REAL parameter set of Bloom-560M (294 tensors, 559 M elements): the optimizer class as the step uses it — state placement
(optimizer._staggered), launch form (CTMI_ADAMW_FLAT) — timed alone with HIP events.  usage: python tools/adamw_model_probe.py [stagger 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleantransformer_amd import optimizer as O
from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM

if len(sys.argv) > 1 and sys.argv[1] == "0":
    O._STAGGER_MIN = 1 << 62
torch.manual_seed(0)
cfg = BloomConfig(vocab_size=250880, hidden_size=1024, n_layer=24, num_attention_heads=16, compute_dtype="bf16")
m = BloomForCausalLM(cfg)
m._tie_weight()
m = m.to("cuda:0")
ps = [p for p in m.parameters()]
for p in ps:
    p.grad = torch.randn_like(p) * 1e-3
opt = O.AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
n = sum(p.numel() for p in ps)
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); opt.step(); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
t = ts[len(ts) // 2]
has_shadow = sum(p.numel() for p in ps if getattr(p, "_ct_shadow", None) is not None)
print(f"AdamW {len(ps)} tensors, {n} elements: median {t:.3f} ms (min {ts[0]:.3f})  {28.0 * n / t / 1e9:.2f} TB/s at 28 B/param "
      f"(stagger {'off' if O._STAGGER_MIN > 1 << 60 else 'on'}, CTMI_ADAMW_FLAT={os.environ.get('CTMI_ADAMW_FLAT', '1')})")
