"""Secondary measurement (BASELINE configs[3]): GPT-2-medium (24 L, n_embd 1024, 16 heads, V 50257, n_positions 2048) LM-head
training step — forward, shifted cross-entropy, backward, fused AdamW — bf16 compute, B*S = 8192 tokens with S = 2048, on one
MI355X.  Prints one JSON line (same fields as bench.py where they apply).  bench.py (Bloom-560M) stays the headline metric."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
V, H, L, NH, P = 50257, 1024, 24, 16, 2048
PEAK = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seq", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from cleantransformer_amd import ops
    from cleantransformer_amd.models.modeling_gpt import GPTConfig, GPTLMHeadModel
    from cleantransformer_amd.optimizer import AdamW
    cfg = GPTConfig(vocab_size=V, n_embd=H, n_positions=P, n_layer=L, n_head=NH, n_ctx=8, embd_pdrop=0.0, attn_pdrop=0.0,
                    resid_pdrop=0.0, compute_dtype="bf16")      # n_ctx only sizes the unused tril buffer
    m = GPTLMHeadModel(cfg, version="gpt2").to(dev).train()
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02, generator=g)
            elif ("norm" in n or "ln_f" in n) and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    for blk in m.gpt.blocks:
        blk.mlp[3].p = 0.0
    opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    B, S = args.batch, args.seq
    ids = torch.randint(0, V, (B, S), generator=torch.Generator(device=dev).manual_seed(999), device=dev)
    am = torch.ones(B, S, dtype=torch.long, device=dev)

    def step():
        (loss, _, _), _ = m(ids, attention_mask=am, labels=ids)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    timer = ops.KernelTimer(["lm_head_fwd"])
    ops.set_timer(timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    import math
    if not math.isfinite(float(loss.detach())):                          # a timing of garbage is not a measurement (bench.py has the long version)
        raise SystemExit(f"{__file__}: the loss after the timed steps is {float(loss.detach())}; refusing to report a throughput for it")
    n_mm = L * 12 * H * H + V * H
    f_tok = 6.0 * n_mm + 6.0 * L * S * H                                  # attention counted causal-half, as SURVEY §8(d)
    tok_s = B * S * args.steps / dt
    head = timer.ms("lm_head_fwd")
    head_avg = sum(head) / max(1, len(head))
    print(json.dumps({
        "metric": "LM training tokens/sec/step GPT-2-medium bf16", "value": round(tok_s, 1), "unit": "tokens/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"GPT-2-medium (24L, n_embd=1024, nh=16, V=50257) LM-head training step, B={B} S={S} (BASELINE configs[3])"},
        "final_loss": round(float(loss.detach()), 4),
        "roofline": {"bound": "mfma", "kernel": "LM-head forward [T,1024]x[50257,1024]^T", "achieved": round(2.0 * B * S * H * V / (head_avg * 1e-3) / 1e12, 1),
                     "peak": PEAK, "unit": "TFLOP/s", "avg_launch_ms": round(head_avg, 4),
                     "step_achieved": round(tok_s * f_tok / 1e12, 1), "step_frac": round(tok_s * f_tok / 1e12 / PEAK, 4), "flops_per_token": f_tok}}))


if __name__ == "__main__":
    main()
