#!/usr/bin/env python3
"""How much of the step does the power cap cost?  The same Bloom-560M SFT step as bench.py (B=8, S=1024, bf16) timed twice in one process:
with bench.py's random-init weights, and with EVERY parameter zero (activations, logits and gradients are then zero too: the kernels execute
the same instructions on operands that do not switch; the loss is ln V, finite).  The second number is what this schedule of kernels would take
if data switching were free — the ceiling for energy-side optimisation at this instruction count (profiles/r04_power_samples.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from cleantransformer_amd.optimizer import AdamW


def run(zero, steps=20, warm=5):
    dev = torch.device("cuda:0")
    m = bench.build_model(dev, "bf16")
    if zero:
        with torch.no_grad():
            for p in m.parameters():
                p.zero_()
    opt = AdamW(m.parameters(), lr=0.0 if zero else 1e-5, weight_decay=0.0 if zero else 0.01, decoupled=True)
    ids = torch.randint(0, bench.V, (8, 1024), generator=torch.Generator(device=dev).manual_seed(999), device=dev)
    am = torch.ones(8, 1024, dtype=torch.long, device=dev)

    def step():
        out, _ = m(input_ids=ids, attention_mask=am, labels=ids)
        opt.zero_grad()
        out[0].backward()
        opt.step()
        return out[0]
    for _ in range(warm):
        loss = step()
    torch.cuda.synchronize()
    s = bench._PowerSampler().start()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        loss = step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    s.stop()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))[steps // 2]
    print(f"{'all-zero parameters' if zero else 'random-init parameters':24s} {ms:7.3f} ms/step   loss {float(loss.detach()):.4f}   power {s.summary()}", flush=True)
    del m, opt
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run(False)
    run(True)
