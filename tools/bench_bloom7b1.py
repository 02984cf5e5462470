"""Secondary measurement (BASELINE configs[4] geometry on ONE GPU): Bloom-7B1 (30 L, H 4096, 32 heads -> head_dim 128, V 250880)
SFT step — forward, loss, backward, fused AdamW — bf16 compute with fp32 master weights / grads / Adam state (~127 GB of the
288 GB HBM + activations), B*S = 4096 tokens at S = 2048.  One JSON line; bench.py (Bloom-560M) stays the headline metric."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=30)
    args = ap.parse_args()
    bench.V, bench.H, bench.L, bench.NH = 250880, 4096, args.layers, 32
    dev = torch.device("cuda:0")
    from cleantransformer_amd import ops
    from cleantransformer_amd.optimizer import AdamW
    m = bench.build_model(dev, "bf16")
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    B, S = args.batch, args.seq
    ids = torch.randint(0, bench.V, (B, S), generator=torch.Generator(device=dev).manual_seed(999), device=dev)
    am = torch.ones(B, S, dtype=torch.long, device=dev)

    def step():
        outputs, _ = m(input_ids=ids, attention_mask=am, labels=ids)
        opt.zero_grad()
        outputs[0].backward()
        opt.step()
        return outputs[0]

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    import math
    if not math.isfinite(float(loss.detach())):                          # a timing of garbage is not a measurement (bench.py has the long version)
        raise SystemExit(f"{__file__}: the loss after the timed steps is {float(loss.detach())}; refusing to report a throughput for it")
    Hh, Ll, Vv = bench.H, bench.L, bench.V
    n_mm = Ll * 12 * Hh * Hh + Vv * Hh
    f_tok = 6.0 * n_mm + 6.0 * Ll * S * Hh
    tok_s = B * S * args.steps / dt
    print(json.dumps({
        "metric": "SFT tokens/sec/step Bloom-7B1 bf16 (1 GPU)", "value": round(tok_s, 1), "unit": "tokens/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Bloom-7B1 geometry ({Ll}L, H=4096, nh=32, V=250880) SFT step, B={B} S={S}, fp32 master/grads/Adam"},
        "final_loss": round(float(loss.detach()), 4), "hbm_gb_allocated": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        "roofline": {"bound": "mfma", "peak": 2500.0, "unit": "TFLOP/s", "step_achieved": round(tok_s * f_tok / 1e12, 1),
                     "step_frac": round(tok_s * f_tok / 1e12 / 2500.0, 4), "flops_per_token": f_tok}}))


if __name__ == "__main__":
    main()
