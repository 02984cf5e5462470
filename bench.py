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
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (profiles/rNN_lmhead_traffic.json, newest
    round: separate FETCH_SIZE / WRITE_SIZE passes of the same command, gfx950 correction applied — the counters cannot be
    collected inside a timed run); None if no profile is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_lmhead_traffic.json")), reverse=True):
        try:
            return float(json.load(open(f))["traffic_bytes_per_launch"])
        except Exception:
            continue
    return None


def _profiled_step_traffic():
    """HBM bytes per STEP, all kernels, from the committed PMC passes over this very command (profiles/rNN_step_traffic.json, newest
    round first; tools/collect_profiles.sh + tools/summarize_profiles.py).  None if no such file exists."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")), reverse=True):
        try:
            return float(json.load(open(f))["traffic_bytes_per_step"])
        except Exception:
            continue
    return None


def _smi_snapshot():
    """Power cap / draw and current clocks of GPU 0 as rocm-smi reports them (best effort; None when the tool is missing or fails): with the
    shader-clock probes this is what explains a 3-4 % spread of ms_per_step between boxes of the pool without prose."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "-d", "0", "--showpower", "--showmaxpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("power", "sclk", "mclk", "fclk", "temperature (sensor junction)", "temperature (sensor memory)")):
                keep[k] = v
        return keep or None
    except Exception:
        return None


class _PowerSampler:
    """rocm-smi package power / shader clock of GPU 0, sampled by a thread (one rocm-smi process per sample, ~10 per second) while the steps run.
    Round 4 found the step power-limited — the package at ~1.3 kW of its 1.4 kW cap, the firmware paying with the shader clock
    (profiles/r04_power_samples.txt) — so the line carries the evidence itself.  Best effort: `summary()` is None without the tool."""

    def __init__(self):
        import shutil
        import threading
        self.exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        self.power, self.sclk, self._stop = [], [], False
        self.thread = threading.Thread(target=self._run, daemon=True) if os.path.exists(self.exe) else None

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                r = subprocess.run([self.exe, "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10)
                m = re.search(r"Package Power \(W\): ([0-9.]+)", r.stdout)
                c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r.stdout)
                if m:
                    self.power.append(float(m.group(1)))
                if c:
                    self.sclk.append(int(c.group(1)))
            except Exception:
                return

    def start(self):
        if self.thread is not None:
            self.thread.start()
        return self

    def stop(self):
        self._stop = True
        if self.thread is not None:
            self.thread.join(timeout=15)

    def summary(self):
        if not self.power:
            return None
        busy = [p for p in self.power[1:]] or self.power                    # the first sample may predate the first step
        med = lambda v: sorted(v)[len(v) // 2]                                # noqa: E731
        return {"samples": len(busy), "package_power_w_median": med(busy), "package_power_w_max": max(busy),
                "sclk_mhz_median": med(self.sclk) if self.sclk else None,
                "note": "rocm-smi sampled ~10 x per second during the extra steps that follow the timed region (its figure is a moving average: the first samples still see the idle gap before them; cap: smi_before)"}


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


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def _cpu_sample(L_, B, S, warm, timed, budget_s):
    """One shape of BASELINE.md §3: the oracle's full SFT step (forward -> zero_grad -> backward -> AdamW), `warm` warm-up steps
    then up to `timed` timed ones (median), stopping early once `budget_s` seconds are spent."""
    from oracle import bloom_ref as R
    sh = R.BloomShape(V, H, L_, NH)
    g = torch.Generator().manual_seed(5)
    p = {}
    for n in R.param_names(sh):
        shp = R.param_shape(sh, n)
        p[n] = (torch.randn(shp, generator=g) * 0.02) if len(shp) > 1 else (torch.ones(shp) if "layernorm.weight" in n or "ln_f.weight" in n else torch.zeros(shp))
    ids = torch.randint(0, V, (B, S), generator=g)
    am = torch.ones(B, S, dtype=torch.long)
    st = R.AdamState(p)
    t_all, t_start = [], time.time()
    for i in range(warm + timed):
        t0 = time.time()
        R.train_step(p, sh, ids, am, st)
        t_all.append(time.time() - t0)
        if time.time() - t_start > budget_s and i >= warm:
            break
    done_timed = t_all[warm:] if len(t_all) > warm else t_all[-1:]
    dt = sorted(done_timed)[len(done_timed) // 2]
    return {"layers": L_, "batch": B, "seq": S, "tokens_per_s": round(B * S / dt, 2), "s_per_step": round(dt, 3),
            "warmup_steps": min(warm, len(t_all) - len(done_timed)), "timed_steps": len(done_timed)}


def cpu_baseline(mode="full"):
    """the oracle (CPU restatement of the reference path, pinned to the reference through the committed golden
    vectors) timed on THIS box's host cores, fp32, plain torch CPU ops, on a bounded sample of the workload:
      C1 (BASELINE configs[0]: 2-layer slice, B=2, S=128), Bloom-560M 24 layers B=2 S=128, and — mode "full" — 24 layers B=2
      S=1024 (one timed step after one warm-up: a step takes tens of seconds).
    `value` is the 24-layer S=128 sample (the configuration closest to the GPU workload that fits the time bound); all samples
    are listed with thread count and CPU model."""
    # thread count: the oracle's fp32 CPU kernels peak around 32 threads on the GPU boxes' 128-core / 256-thread hosts (measured:
    # 1024 tokens of the 24-layer model take 5.4 s at 32 threads, 29 s at torch's default of 128) — the baseline uses the fast setting
    old_threads = torch.get_num_threads()
    threads = max(1, min(32, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    samples = [_cpu_sample(2, 2, 128, 2, 3, 12.0), _cpu_sample(L, 2, 128, 1, 3, 25.0)]
    if mode == "full":
        samples.append(_cpu_sample(L, 2, 1024, 1, 1, 45.0))
    head = samples[1]
    torch.set_num_threads(old_threads)
    return {"value": head["tokens_per_s"], "unit": "tokens/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
            "host_logical_cpus": os.cpu_count(),
            "sample": f"oracle (CPU restatement) fp32 full SFT step (fwd, zero_grad, bwd, AdamW), torch.set_num_threads({threads}); value = "
                      f"Bloom-560M 24L B=2 S=128 median of {head['timed_steps']} timed step(s) after {head['warmup_steps']} warm-up, "
                      f"{head['s_per_step']} s/step",
            "samples": samples}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp16"],
                    help="compute dtype; bf16 is the metric's (BASELINE.json); fp16 runs the reference's autocast loop (GradScaler: scale, unscale + inf check, step, "
                         "update); fp32 parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "short"], help="short: skip the 24-layer S=1024 CPU sample")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: wire dtype of the gradient buckets (bf16 halves the xGMI bytes; gradients then carry bf16 rounding)")
    ap.add_argument("--ddp-backend", default=os.environ.get("CTMI_DDP_BACKEND", "torch"), choices=["torch", "rccl"],
                    help="N > 1: gradient collectives through torch.distributed (default) or the library's own RCCL communicator (ctmi_ddp_*)")
    ap.add_argument("--no-comm-probe", action="store_true",
                    help="N > 1: do not time the backend x launch-policy candidates during warm-up; run --ddp-backend / CTMI_DDP_LAUNCH_POLICY as given")
    ap.add_argument("--probe-rccl", action="store_true", help="--gpus N > 1: also time the library's own RCCL communicator (csrc/comm.hip) among the candidates")
    ap.add_argument("--no-power-sampler", action="store_true", help="do not sample rocm-smi power / clocks beside the timed steps")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-class HIP-event pass after the timed region")
    ap.add_argument("--no-padded-sample", action="store_true", help="skip the secondary sample with 25 %% of every row right-padded")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: replay the captured step from a hipGraph (cleantransformer_amd/graph.py) instead of issuing it launch by launch; measured "
                         "equal on the GPU-bound headline shape (profiles/r06_launch_floor.txt), so the default stays the reference's eager loop")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("CTMI_GEMM_SHARED", "2")     # RCCL kernels share the CUs under backward (DistributedDataParallel arms its policy itself)
        # RCCL runs one workgroup per channel: bound the CUs the collectives may hold (and, with CTMI_DDP_LAUNCH_POLICY=reserve, keep
        # exactly that many out of the persistent GEMM launches — trainer/ddp.py); read when the communicator is created
        os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("CTMI_DDP_COMM_CUS", "16"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")                    # "nccl" == RCCL on ROCm

    from cleantransformer_amd import ops
    from cleantransformer_amd.models import modeling_bloom
    from cleantransformer_amd.optimizer import AdamW
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP

    B, S = args.batch, args.seq
    model = build_model(device, args.dtype)
    comm_dtype = torch.bfloat16 if args.comm_dtype == "bf16" else None
    os.environ["CTMI_DDP_BACKEND"] = args.ddp_backend
    opt = AdamW(model.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)    # == torch.optim.AdamW(lr=1e-5), ft_bloom.py:70
    g = torch.Generator(device=device).manual_seed(999 + rank)                       # per-rank data seed
    ids = torch.randint(0, V, (B, S), generator=g, device=device)
    am = torch.ones(B, S, dtype=torch.long, device=device)
    labels = ids.clone()
    batch = {"ids": ids, "am": am, "labels": labels}
    holder = {"net": model}

    scaler = None
    if args.dtype == "fp16":
        # half-precision gradients need loss scaling — the reference's --use_torch_amp loop (ft_bloom_DDP.py:121-127): without it the softmax part of
        # dlogits underflows, the backward operands are mostly exact zeros, the matrix pipe draws less power and the step looks ~1 ms FASTER than it is
        # (measured: profiles/r05_dtype_step.txt; tools/probes/mfma_energy.hip has the all-zero-operand rates)
        from cleantransformer_amd.amp import GradScaler
        scaler = GradScaler()

    # --graph (N = 1): the step of the reference loop, captured once and replayed as ONE hipGraph launch per step (cleantransformer_amd/graph.py: the
    # same ~460 launches in the same order).  A replay removes the HOST side of every launch (6.7 ms of enqueue per step); on this GPU-bound shape the
    # device-side cost of a dependent launch stays what it is and the step time does not move (profiles/r06_launch_floor.txt: 36.50 eager vs 36.55
    # replayed, three interleaved pairs), so the default is the reference's own eager loop.  The first two warm-up steps run eagerly, the third
    # captures; the eager step is timed as well, after the timed region (timing.eager_ms_per_step).
    graphed = None
    if world == 1 and scaler is None and args.graph and args.warmup >= 3:
        from cleantransformer_amd.graph import GraphedStep
        graphed = GraphedStep(model, opt, warmup=2)

    def step():
        if graphed is not None:                                          # (enabled = False: it drops its graph and runs the eager loop itself)
            return graphed(batch["ids"], batch["am"], batch["labels"])
        outputs, _ = holder["net"](input_ids=batch["ids"], attention_mask=batch["am"], labels=batch["labels"])
        loss = outputs[0]
        opt.zero_grad()
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            return loss
        loss.backward()
        opt.step()
        return loss

    # ---- N > 1: which collectives backend and which GEMM launch policy?  No multi-GPU box is available while building, so the bench
    # decides on the box it runs on: every candidate (torch.distributed vs the library's own RCCL communicator) x (shared vs reserve)
    # wraps the SAME model, runs 1 + 3 steps during warm-up (median, max over ranks), the fastest is re-armed for the timed region and
    # all four timings are reported in config.comm.candidates — one SCALE run yields a decision, not a single unexplained point.
    comm_candidates = None
    if world > 1:
        def wrap(backend, policy):
            os.environ["CTMI_DDP_BACKEND"] = backend
            os.environ["CTMI_DDP_LAUNCH_POLICY"] = policy
            return DDP(model, device_ids=[local_rank], comm_dtype=comm_dtype)
        chosen = (args.ddp_backend, os.environ.get("CTMI_DDP_LAUNCH_POLICY", "flow"))
        if not args.no_comm_probe:
            comm_candidates = []
            # (round 5) the library's own RCCL communicator is only probed on request (--probe-rccl): it has never run at world > 1, and a
            # candidate that HANGS takes the whole measurement with it; torch.distributed's RCCL backend x the two launch policies is the default
            for backend in (("torch", "rccl") if args.probe_rccl else ("torch",)):
                for policy in ("flow", "shared", "reserve", "persistent"):
                    rec = {"ddp_backend": backend, "launch_policy": policy}
                    # phase 1: wrap.  Every rank reports whether ITS wrapper exists before any rank enters the wrapper's collectives — a
                    # candidate that cannot be built on one rank (no librccl there, a communicator error) is skipped by all (round-4 advisor)
                    try:
                        net = wrap(backend, policy)
                        werr = None
                    except Exception as e:                                   # noqa: BLE001
                        net, werr = None, f"{type(e).__name__}: {e}"[:200]
                    wt = torch.tensor([0.0 if net is not None else 1.0], dtype=torch.float64, device=device)
                    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
                    if float(wt[0]) > 0.5:
                        if net is not None:
                            net.close()
                        rec["error"] = werr or "the wrapper could not be built on another rank"
                        rec["ms_per_step"] = None
                        comm_candidates.append(rec)
                        if rank == 0:
                            print(f"[bench] comm candidate {rec}", file=sys.stderr, flush=True)
                        continue
                    try:
                        holder["net"] = net
                        step()
                        torch.cuda.synchronize()
                        dist.barrier()
                        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                        evs[0].record()
                        for i in range(3):
                            step()
                            evs[i + 1].record()
                        torch.cuda.synchronize()
                        ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(3))[1]
                        ok = 1.0
                    except Exception as e:                                   # a candidate that cannot run here (e.g. no librccl) is recorded, not fatal
                        ms, ok = 1e9, 0.0
                        rec["error"] = f"{type(e).__name__}: {e}"[:200]
                    tt = torch.tensor([ms, -ok], dtype=torch.float64, device=device)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)                # slowest rank; failed anywhere = failed
                    rec["ms_per_step"] = round(float(tt[0]), 3) if float(tt[1]) < -0.5 else None
                    comm_candidates.append(rec)
                    if rank == 0:                                             # progress on stderr: survives in the log if a later candidate hangs
                        print(f"[bench] comm candidate {rec}", file=sys.stderr, flush=True)
                    if isinstance(holder["net"], DDP):
                        holder["net"].close()
                    holder["net"] = model
            good = [c for c in comm_candidates if c["ms_per_step"] is not None]
            if good:
                best = min(good, key=lambda c: c["ms_per_step"])                 # identical on every rank (all-reduced numbers)
                chosen = (best["ddp_backend"], best["launch_policy"])
        args.ddp_backend = chosen[0]
        holder["net"] = wrap(*chosen)

    def timed_steps(n):
        """n steps between barrier + synchronize brackets; per-step device time from HIP events on the compute stream.
        -> (wall seconds of the bracketed region, per-step milliseconds, host loop seconds, last loss)"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n):
            loss_ = step()
            ev[i + 1].record()
        host = time.perf_counter() - t0                                      # host loop time inside the timed region (back-pressured)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return wall, [ev[i].elapsed_time(ev[i + 1]) for i in range(n)], host, loss_

    for _ in range(args.warmup):
        loss = step()
    # host cost of enqueueing one step, measured with an EMPTY device queue (two steps right after a synchronize: the host is
    # never held back by the GPU).  Inside the timed region below the host runs ahead of the device until the HIP queue pushes
    # back, so its loop time there converges to the GPU step time and says nothing about the host.
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for _ in range(2):
        loss = step()
    host_enqueue_ms = (time.perf_counter() - h0) / 2 * 1e3
    smi_before = _smi_snapshot() if rank == 0 else None
    clock_before = ops.clock_probe(device)                                 # shader clock under MFMA load, chip warm from the warm-up steps
    timer = ops.KernelTimer(["lm_head_fwd"])
    if graphed is None:
        ops.set_timer(timer)
    dt, per_step_ms, host_loop_s, loss = timed_steps(args.steps)
    ops.set_timer(None)
    graph_info = None
    if graphed is not None:
        # the same step issued launch by launch, right after the timed region: the eager number beside the replayed one, and the HIP-event brackets
        # of the LM-head forward (roofline.largest_launch) — a replay runs no host code that could record them
        graph_info = {"enabled": graphed.graph is not None, "replays": graphed.replays, "fallback_reason": graphed.fallback_reason}
        graphed.enabled = False
        for _ in range(2):
            step()
        ops.set_timer(timer)
        _, eager_ms, _, _ = timed_steps(max(5, args.steps // 2))
        ops.set_timer(None)
        graph_info["eager_ms_per_step"] = round(sorted(eager_ms)[len(eager_ms) // 2], 3)
        graph_info["eager_steps"] = len(eager_ms)
        graphed.enabled = graph_info["enabled"]                          # (later samples replay again: two eager warm-up steps, then a new capture)
        if graphed.enabled:
            for _ in range(3):
                step()
    # package power / shader clock while stepping: sampled beside `steps` EXTRA steps right after the timed region, not inside it — ten rocm-smi
    # processes per second beside the timed steps cost 0.07 ms per step (same box, three interleaved pairs: 36.60 vs 36.54; round-4 advisor)
    sampler = _PowerSampler().start() if (rank == 0 and not args.no_power_sampler) else None
    if not args.no_power_sampler:
        for _ in range(max(args.steps, 10)):
            step()
        torch.cuda.synchronize()
    if sampler is not None:
        sampler.stop()
    clock_after = ops.clock_probe(device)                                  # ... and right after the timed steps
    smi_after = _smi_snapshot() if rank == 0 else None
    final_loss = float(loss.detach())
    bad_loss = not (math.isfinite(final_loss) and 0.0 < final_loss < 2.0 * math.log(V))
    if world > 1:                                                         # every rank leaves together (a lone exit would hang the others in the next all-reduce)
        bl = torch.tensor([1.0 if bad_loss else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(bl, op=dist.ReduceOp.MAX)
        bad_loss = bool(float(bl[0]) > 0.5) or bad_loss
    if bad_loss:
        # A timing of garbage is not a measurement — and it is not even conservative: MFMAs on NaN operands draw less power, the chip clocks
        # higher and EVERY kernel of the step runs ~8 % faster (measured in round 4 on a build with a data race: profiles/r04_k2_pairs.txt).
        raise SystemExit(f"bench.py: the loss after the timed steps is {final_loss} (random-init start: ~{math.log(V):.1f}) — "
                         f"the step computes garbage; refusing to report a throughput for it")
    med_ms = sorted(per_step_ms)[len(per_step_ms) // 2]
    if world > 1:
        tt = torch.tensor([dt, med_ms], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med_ms = float(tt[0]), float(tt[1])
    mean_ms = dt / args.steps * 1e3
    tokens_per_s = world * B * S / (med_ms * 1e-3)                       # median of the timed steps (max over ranks)

    # ---- world > 1: what the collectives cost that backward could not hide.  The same wrapped model under the same launch policy runs
    # `steps` more steps with every gradient collective SKIPPED (DistributedDataParallel.stub_collectives: bucket copies, pre-division, the
    # chunked tied-table gradient and its row scatter still run): exposed_ms = step with collectives - step without.  After the timed
    # region and after the loss check — those steps train on unaveraged gradients.
    comm_exposed = None
    if world > 1 and isinstance(holder["net"], DDP):
        holder["net"].stub_collectives = True
        try:
            for _ in range(2):
                step()
            _, st_ms, _, _ = timed_steps(max(5, args.steps))
        finally:
            holder["net"].stub_collectives = False
        sm = sorted(st_ms)[len(st_ms) // 2]
        ts_ = torch.tensor([sm], dtype=torch.float64, device=device)
        dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
        sm = float(ts_[0])
        comm_exposed = {"exposed_ms": round(med_ms - sm, 3), "compute_only_ms_per_step": round(sm, 3), "steps": len(st_ms),
                        "ranks": world, "backend": dist.get_backend(),
                        "note": "compute_only = the same wrapped model and launch policy with every gradient collective skipped "
                                "(median, max over ranks); exposed = ms_per_step - compute_only"}

    # ---- per-class device time of one step (HIP-event brackets inside the library; side stream off so that brackets do not overlap)
    breakdown = None
    if not args.no_breakdown:
        if graphed is not None:
            graphed.enabled = False                                      # the per-class brackets are recorded by host code: eager steps
        side = modeling_bloom._WGRAD_SIDE_STREAM
        modeling_bloom._WGRAD_SIDE_STREAM = False
        step()
        torch.cuda.synchronize()
        ops.profile_begin()
        nb = 3
        for _ in range(nb):
            step()
        prof = ops.profile_end()
        modeling_bloom._WGRAD_SIDE_STREAM = side
        breakdown = {"note": f"sum of HIP-event brackets per class over {nb} extra steps with the weight-gradient side stream off, ms per step; "
                             "brackets include launch gaps inside a library call; 'launches' = library calls per step",
                     "ms": {k: round(v[0] / nb, 3) for k, v in prof.items()}, "launches": {k: v[1] // nb for k, v in prof.items()}}
        breakdown["ms_total"] = round(sum(breakdown["ms"].values()), 3)
        if graphed is not None:
            graphed.enabled = graph_info["enabled"]
            if graphed.enabled:
                for _ in range(3):
                    step()

    # ---- secondary sample (SURVEY §8(d) "Synthetic inputs"): 25 % of every row right-padded
    padded = None
    if not args.no_padded_sample:
        am_p = am.clone()
        am_p[:, (S * 3) // 4:] = 0
        batch["am"] = am_p
        for _ in range(2):
            step()
        _, ps_ms, _, _ = timed_steps(max(5, args.steps // 2))
        batch["am"] = am
        pm = sorted(ps_ms)[len(ps_ms) // 2]
        if world > 1:
            tp = torch.tensor([pm], dtype=torch.float64, device=device)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            pm = float(tp[0])
        padded = {"mask": "last 25 % of every row padded (attention_mask = 0), labels unchanged", "ms_per_step": round(pm, 3),
                  "tokens_per_s": round(world * B * S / (pm * 1e-3), 1), "steps": len(ps_ms)}

    if rank == 0:
        f_tok = flops_per_token(S)
        step_tflops = tokens_per_s / world * f_tok / 1e12               # per GPU
        head_ms = timer.ms("lm_head_fwd")
        head_avg = sum(head_ms) / max(1, len(head_ms))
        head_flops = 2.0 * B * S * H * V                                 # algorithmic FLOPs of one LM-head GEMM launch
        head_tflops = head_flops / (head_avg * 1e-3) / 1e12 if head_avg > 0 else 0.0
        out = {
            "metric": "SFT tokens/sec/step Bloom-560M bf16", "value": round(tokens_per_s, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(med_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Bloom-560M ({L}L, H=1024, nh=16, V={V}) SFT step fwd+bwd+AdamW, B={B} S={S} per GPU "
                                   f"(BASELINE configs[1]), random-init weights, fp32 master/grads/Adam state",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}",
                       "comm_dtype": args.comm_dtype if world > 1 else None,
                       "comm": None if world == 1 else {"nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                                                        "launch_policy": os.environ.get("CTMI_DDP_LAUNCH_POLICY", "flow"),
                                                        "ddp_backend": args.ddp_backend, "candidates": comm_candidates,
                                                        "ranks": world, "exposed_ms": None if comm_exposed is None else comm_exposed["exposed_ms"],
                                                        "exposed": comm_exposed,
                                                        "tied_chunk_mb": os.environ.get("CTMI_DDP_TIED_CHUNK_MB", "64")},
                       "padded_sample": padded},
            "graph": graph_info if graph_info is not None else {"enabled": False, "why": "eager loop (default); --graph replays the captured step at N = 1"},
            "timing": {"value_from": "median of the per-step HIP-event times of the timed steps (max over ranks)",
                       "eager_ms_per_step": None if graph_info is None else graph_info["eager_ms_per_step"],
                       "ms_per_step_median": round(med_ms, 3), "ms_per_step_mean_wall": round(mean_ms, 3),
                       "ms_per_step_min": round(min(per_step_ms), 3), "ms_per_step_max": round(max(per_step_ms), 3),
                       "wall_s_timed_region": round(dt, 4),
                       "shader_clock_mhz_before": round(clock_before, 1), "shader_clock_mhz_after": round(clock_after, 1),
                       "shader_clock_note": "ctmi_clock_probe: s_memtime / s_memrealtime of one wave while 2048 workgroups issue bf16 MFMAs (~6 ms), "
                                            "launched right before / right after the timed region; the 2.5 PF peak assumes 2400 MHz",
                       "smi_before": smi_before, "smi_after": smi_after, "power_while_stepping": sampler.summary() if sampler is not None else None},
            "final_loss": round(final_loss, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 2),
            "host_loop_ms_per_step_in_timed_region": round(host_loop_s / args.steps * 1e3, 2),
            # the step-level fraction — algorithmic FLOPs of the whole step (6 N_mm + 6 L S H per token, attention
            # causal-half) over the step time, against the dense bf16 MFMA peak.  The largest single launch is a sub-entry.
            "roofline": {"bound": "mfma", "kernel": "whole SFT step (all kernels; F_tok = 6*N_mm + 6*L*S*H)",
                         "achieved": round(step_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4), "traffic": _profiled_step_traffic(),
                         "traffic_note": "HBM bytes per step, all kernels (separate --pmc FETCH_SIZE / WRITE_SIZE passes over this command, FETCH doubled per the gfx950 note; profiles/r*_step_traffic.json)",
                         "flops_per_token": f_tok,
                         "largest_launch": {"kernel": "gemm_glds_kernel<bf16,NT,256x256 ping-pong> LM-head forward [T,1024]x[250880,1024]^T",
                                            "achieved": round(head_tflops, 1), "frac": round(head_tflops / PEAK_BF16_TFLOPS, 4),
                                            "avg_launch_ms": round(head_avg, 4), "launches": len(head_ms),
                                            "traffic_bytes_per_launch": _profiled_traffic(),
                                            "algorithmic_bytes_per_launch": 2.0 * (B * S * H + V * H + B * S * V)},
                         "breakdown_ms_per_step": breakdown},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
