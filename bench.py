#!/usr/bin/env python3
"""bench.py — SFT tokens/sec of the Bloom-560M step on MI355X (BASELINE.json metric, configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = the reference's SFT step (examples/ft_bloom.py:84-90): forward (24-layer Bloom-560M, tied LM head,
shifted cross entropy) -> zero_grad -> backward -> AdamW(lr=1e-5, wd=0.01), on a synthetic batch B=8, S=1024 per GPU
(weak scaling), bf16 compute with fp32 master weights / gradients / optimizer state.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

V, H, L, NH = 250880, 1024, 24, 16
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def flops_per_token(S: int) -> float:
    """6*N_mm + 6*L*S*H (attention counted causal-half)."""
    n_mm = L * 12 * H * H + V * H
    return 6.0 * n_mm + 6.0 * L * S * H


def _profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (profiles/r01_lmhead_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied); None if the profile is absent."""
    try:
        return float(json.load(open(os.path.join(ROOT, "profiles", "r01_lmhead_traffic.json")))["traffic_bytes_per_launch"])
    except Exception:
        return None


def build_model(device, compute_dtype):
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    cfg = BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=NH, compute_dtype=compute_dtype)
    with torch.device("meta"):
        m = BloomForCausalLM(cfg)
    m = m.to_empty(device=device)
    m._tie_weight()
    g = torch.Generator(device=device).manual_seed(1234)
    with torch.no_grad():                                  # random init of the real architecture (no checkpoints offline)
        for n, p in m.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02, generator=g)
            elif n.endswith("layernorm.weight") or n.endswith("ln_f.weight"):
                p.fill_(1.0)
            else:
                p.zero_()
    return m.train()


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement, validated against the reference) timed on this box's host cores on a bounded sample:
    Bloom-560M 24 layers, B=2, S=128, fp32, full SFT step."""
    from oracle import bloom_ref as R
    B, S = 2, 128
    sh = R.BloomShape(V, H, L, NH)
    g = torch.Generator().manual_seed(5)
    p = {}
    for n in R.param_names(sh):
        shp = R.param_shape(sh, n)
        p[n] = (torch.randn(shp, generator=g) * 0.02) if len(shp) > 1 else (torch.ones(shp) if "layernorm.weight" in n or "ln_f.weight" in n else torch.zeros(shp))
    ids = torch.randint(0, V, (B, S), generator=g)
    am = torch.ones(B, S, dtype=torch.long)
    st = R.AdamState(p)
    t_all = []
    t_start = time.time()
    for i in range(4):
        t0 = time.time()
        R.train_step(p, sh, ids, am, st)
        t_all.append(time.time() - t0)
        if time.time() - t_start > seconds_budget and i >= 1:
            break
    timed = t_all[1:] if len(t_all) > 1 else t_all
    dt = sorted(timed)[len(timed) // 2]
    return {"value": round(B * S / dt, 2), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (CPU restatement) Bloom-560M 24L fp32 SFT step, B={B} S={S}, {len(timed)} timed step(s) after 1 warm-up, "
                      f"{dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if os.environ.get("CTMI_BENCH_ONE_DEVICE"):            # plumbing test only: all ranks share cuda:0 (with gloo)
        local_rank = 0
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("CTMI_GEMM_SHARED", "1")     # RCCL kernels share the CUs under backward
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("CTMI_DIST_BACKEND", "nccl"))       # "nccl" == RCCL on ROCm

    from cleantransformer_amd import ops
    from cleantransformer_amd.optimizer import AdamW
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP

    B, S = args.batch, args.seq
    model = build_model(device, args.dtype)
    net = DDP(model, device_ids=[local_rank]) if world > 1 else model
    opt = AdamW(net.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)      # == torch.optim.AdamW(lr=1e-5), ft_bloom.py:70
    g = torch.Generator(device=device).manual_seed(999 + rank)                       # per-rank data seed
    ids = torch.randint(0, V, (B, S), generator=g, device=device)
    am = torch.ones(B, S, dtype=torch.long, device=device)
    labels = ids.clone()

    def step():
        outputs, _ = net(input_ids=ids, attention_mask=am, labels=labels)
        loss = outputs[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    timer = ops.KernelTimer(["lm_head_fwd"])
    ops.set_timer(timer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host-side launch time per step (GPU runs behind)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    final_loss = float(loss.detach())
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    tokens_per_s = world * B * S * args.steps / dt

    if rank == 0:
        f_tok = flops_per_token(S)
        step_tflops = tokens_per_s / world * f_tok / 1e12               # per GPU
        head_ms = timer.ms("lm_head_fwd")
        head_avg = sum(head_ms) / max(1, len(head_ms))
        head_flops = 2.0 * B * S * H * V                                 # algorithmic FLOPs of one LM-head GEMM launch
        head_tflops = head_flops / (head_avg * 1e-3) / 1e12 if head_avg > 0 else 0.0
        out = {
            "metric": "SFT tokens/sec/step Bloom-560M bf16", "value": round(tokens_per_s, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Bloom-560M (24L, H=1024, nh=16, V=250880) SFT step fwd+bwd+AdamW, B={B} S={S} per GPU "
                                   f"(BASELINE configs[1]), random-init weights, fp32 master/grads/Adam state",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 2),
            "roofline": {"bound": "mfma", "kernel": "gemm_glds_kernel<bf16,NT,256x256 ping-pong> LM-head forward [T,1024]x[250880,1024]^T",
                         "achieved": round(head_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(head_tflops / PEAK_BF16_TFLOPS, 4), "traffic": _profiled_traffic(),
                         "avg_launch_ms": round(head_avg, 4), "launches": len(head_ms),
                         "step_achieved": round(step_tflops, 1), "step_frac": round(step_tflops / PEAK_BF16_TFLOPS, 4),
                         "flops_per_token": f_tok},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
