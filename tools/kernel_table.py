#!/usr/bin/env python3
"""ONE evidence table for the hot kernels of the Bloom-560M step (round-4 verdict item 9): time, algorithmic FLOP or bytes, fraction of the
chip peak, HBM-side traffic against the algorithmic bytes, matrix-pipe busy share, joules per launch.

    python tools/kernel_table.py gpurun_out/evidence profiles r05        (after tools/collect_profiles.sh)

Sources (all from ONE collection on one box): rocprofv3 --kernel-trace --stats over `bench.py --steps 10 --warmup 3` (15 steps traced);
separate counter-only passes over the same command for FETCH_SIZE, WRITE_SIZE (FETCH doubled: gfx950 tallies the 128-byte requests of wide
reads at 64 B — MI355X_MICROARCH.md) and for SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; tools/energy_probe.py for joules.
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the share of SIMD cycles in which the matrix pipe
is executing, at whatever shader clock the counted run had (GRBM_GUI_ACTIVE is summed over the eight XCDs — dividing by the raw sum, as
profiles/r04_pmc_lmhead_mfma.txt did, understates it eightfold).
Algorithmic work per step (Bloom-560M, B = 8, S = 1024, T = 8192, H = 1024, V = 250 880, 24 layers): see WORK below."""
import csv
import glob
import os
import re
import sys

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
T, H, V, L, S, B, NH, HD = 8192, 1024, 250880, 24, 1024, 8, 16, 64
GF = lambda n, k: 2.0 * T * n * k                                       # noqa: E731  one [T,k] x [k,n] product
ATT = 4.0 * B * NH * S * S * HD / 2                                      # causal-half attention forward of one layer
P = 559214592
# kernel-name pattern -> (label, unit, work per STEP, algorithmic HBM bytes per STEP or None, energy-probe line or None)
WORK = [
    (r"gemm_wgrad_grouped_kernel", "layer weight gradients, grouped (4 products + 2 bias sums per launch)", "TF", L * (GF(H, 4 * H) * 2 + GF(H, H) + GF(3 * H, H)),
     L * ((H + 4 * H + 4 * H + H + H + H + 3 * H + H) * T * 2 + 12 * H * H * 4), "block wgrads, grouped launch"),
    (r"wgrad_partials_reduce_k", "  + sum of the K-halves of the cut tiles", "TB", L * 128 * 128 * 256 * 4 * 3, L * 128 * 128 * 256 * 4 * 3, None),
    (r"gemm_glds_kernel<float, true, true, 0, 4, 4, true", "layer weight gradients, per product (round-4 form)", "TF", L * (GF(H, 4 * H) * 2 + GF(H, H) + GF(3 * H, H)), None, None),
    (r"gemm_glds_kernel<unsigned short, false, false, 0, 8, 4, true, false, false>", "LM-head forward (logits)", "TF", GF(V, H), T * H * 2 + V * H * 2 + T * V * 2, "lm_head fwd"),
    (r"gemm_glds_kernel<float, true, true, 0, 8, 4, true", "LM-head weight gradient ([V,H] fp32)", "TF", GF(V, H), T * H * 2 + T * V * 2 + V * H * 4, "lm_head wgrad"),
    (r"gemm_glds_kernel<unsigned short, false, true, 0, 8, 4, true", "LM-head data gradient (K = V, split 2)", "TF", GF(V, H), T * V * 2 + V * H * 2 + T * H * 2 * 3, "lm_head dgrad"),
    (r"gemm_glds_kernel<unsigned short, false, true, 0, 4, 4, true, false, false>", "data gradients qkv / dense / h->4h (128x256 tile)", "TF", L * (GF(H, 3 * H) + GF(H, H) + GF(H, 4 * H)),
     L * ((3 * H + H + 4 * H + 3 * H) * T * 2 + 8 * H * H * 2), "qkv dgrad"),
    (r"gemm_glds_kernel<unsigned short, false, false, 0, 4, 4, true, true, false>", "dense / 4h->h forward + residual (128x256 tile)", "TF", L * (GF(H, H) + GF(H, 4 * H)),
     L * ((H + 4 * H + 4 * H) * T * 2 + 5 * H * H * 2), "4hh fwd"),
    (r"gemm_glds_kernel<unsigned short, false, true, 2, [48], 4, true", "4h->h data gradient x gelu'(u) (dGELU epilogue; 256x256 tile since round 6)", "TF", L * GF(4 * H, H), L * ((H + 4 * H + 4 * H) * T * 2 + 4 * H * H * 2), "4hh dgrad"),
    (r"gemm_glds_kernel<unsigned short, false, false, 1, 8, 4, true, false, true>", "h->4h forward + GELU (256x256 tile, cross-lane epilogue)", "TF", L * GF(4 * H, H),
     L * ((H + 4 * H + 4 * H) * T * 2 + 4 * H * H * 2), "h4h fwd"),
    (r"gemm_glds_kernel<unsigned short, false, false, 0, 8, 4, true, false, true>", "QKV forward (256x256 tile, cross-lane epilogue)", "TF", L * GF(3 * H, H), L * ((H + 3 * H) * T * 2 + 3 * H * H * 2), "qkv fwd"),
    (r"attn32_fwd_kernel", "attention forward (hd = 64, causal)", "TF", L * ATT, L * 4 * T * H * 2, "attention fwd"),
    (r"attn32_dq_kernel", "attention backward: dQ", "TF", L * ATT * 2.5 * 3 / 7, L * 5 * T * H * 2, None),
    (r"attn32_dkdv_kernel", "attention backward: dK, dV", "TF", L * ATT * 2.5 * 4 / 7, L * 6 * T * H * 2, "attention bwd"),
    (r"adamw_(mt|flat)_k", "AdamW, fused (fp32 state + bf16 shadow)", "TB", P * 30.0, P * 30.0, "AdamW (140 M params)"),
    (r"ce_fused_k", "shifted cross entropy: loss + dlogits in one pass", "TB", T * V * 2 * 2.0, T * V * 2 * 2.0, None),
    (r"ln_fwd_vec", "LayerNorm forward", "TB", (2 * L + 2) * T * H * 2 * 2.0, (2 * L + 2) * T * H * 2 * 2.0, None),
    (r"ln_bwd_vec", "LayerNorm backward (+ residual-gradient add, bias column sums)", "TB", (2 * L + 2) * T * H * 2 * 4.0, (2 * L + 2) * T * H * 2 * 4.0, "LayerNorm bwd [8192,1024]"),
    (r"splitk_reduce", "split-K reduce", "TB", None, None, None),
    (r"colsum_part", "bias column sums (separate pass)", "TB", None, None, None),
    (r"reduce_jobs_k", "partial-row reductions of a block (LayerNorm affine, bias gradients)", "TB", None, None, None),
    (r"wgrad_tail_k", "block tail (round 6): sum of the K-halves of the cut tiles + the partial-row reductions in one launch", "TB", L * 128 * 128 * 256 * 4 * 3, L * 128 * 128 * 256 * 4 * 3, None),
]
PEAK = {"TF": 2500e12, "TB": 8e12}


def one(pattern):
    fs = glob.glob(os.path.join(src, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


def per_kernel_counter(run, counters):
    f = one(f"{run}/*/*_counter_collection.csv")
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in counters:
            d = out.setdefault(r["Kernel_Name"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def match(table, pat):
    tot = {}
    for k, d in table.items():
        if re.search(pat, k):
            for c, v in d.items():
                tot[c] = tot.get(c, 0.0) + v
    return tot


def energy_lines():
    f = os.path.join(src, f"{rnd}_energy_probe.txt")
    out = {}
    if os.path.exists(f):
        for ln in open(f):
            m = re.match(r"^(.{34})\s+([0-9.]+) us\s+([0-9.]+) W .*?([0-9.]+) mJ/launch", ln)
            if m and "[zero" not in ln:
                out[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    return out


LMH = "void gemm_glds_kernel<unsigned short, false, false, 0, 8, 4, true, false, false>"


def traced_steps(run_dir, csvname="kernel_trace"):
    """steps a run covers = launches of the LM-head forward (one per step, > 1 ms each)"""
    f = one(f"{run_dir}/*/*_{csvname}.csv")
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(LMH) and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) > 1e6:
            seen.add(r.get("Dispatch_Id") or r.get("Correlation_Id") or r["Start_Timestamp"])
    return max(len(seen), 1)


def table(run_dir, title):
    STEPS = traced_steps(run_dir)
    PSTEPS = traced_steps("pmc_step_FETCH_SIZE", "counter_collection") if one("pmc_step_FETCH_SIZE/*/*_counter_collection.csv") else STEPS
    stats = list(csv.DictReader(open(one(f"{run_dir}/*/*_kernel_stats.csv"))))
    fetch = per_kernel_counter("pmc_step_FETCH_SIZE", {"FETCH_SIZE"})
    write = per_kernel_counter("pmc_step_WRITE_SIZE", {"WRITE_SIZE"})
    mfma = per_kernel_counter("pmc_step_MFMA", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
    en = energy_lines()
    rows = []
    for pat, label, unit, work, alg, eline in WORK:
        rs = [r for r in stats if re.search(pat, r["Name"])]
        if not rs:
            continue
        ms = sum(float(r["TotalDurationNs"]) for r in rs) / 1e6 / STEPS
        calls = sum(int(r["Calls"]) for r in rs) / STEPS
        rate = (work / (ms * 1e-3)) if work else None
        fb = match(fetch, pat).get("FETCH_SIZE", 0.0) * 1024 * 2 / PSTEPS
        wb = match(write, pat).get("WRITE_SIZE", 0.0) * 1024 / PSTEPS
        mm = match(mfma, pat)
        busy = (mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (mm["GRBM_GUI_ACTIVE"] / 8 * 1024)) if mm.get("GRBM_GUI_ACTIVE") else None
        e = en.get(eline) if eline else None
        rows.append((ms, label, calls, unit, work, rate, fb + wb, alg, busy, e))
    rows.sort(reverse=True)
    tot = sum(float(r["TotalDurationNs"]) for r in stats) / 1e6 / STEPS
    out = [f"### {title}", "",
           f"Sum of kernel time {tot:.2f} ms per step; the rows below cover {sum(r[0] for r in rows):.2f} ms.", "",
           "| kernel (role in the step) | ms / step | launches / step | algorithmic work / step | achieved | of chip peak | FETCH + WRITE / step | vs algorithmic bytes | MFMA busy | energy probe: µs, W, mJ / launch |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for ms, label, calls, unit, work, rate, traffic, alg, busy, e in rows:
        w = "—" if not work else (f"{work / 1e12:.2f} TFLOP" if unit == "TF" else f"{work / 1e9:.1f} GB")
        a = "—" if not rate else (f"{rate / 1e12:.0f} TF/s" if unit == "TF" else f"{rate / 1e12:.2f} TB/s")
        fpk = "—" if not rate else f"{100 * rate / PEAK[unit]:.1f} %"
        tr = f"{traffic / 1e9:.2f} GB" if traffic else "—"
        ratio = f"{traffic / alg:.2f}×" if (traffic and alg) else "—"
        bz = f"{100 * busy:.0f} %" if busy is not None else "—"
        es = f"{e[0]:.0f} µs, {e[1]:.0f} W, {e[2]:.1f} mJ" if e else "—"
        out.append(f"| {label} | {ms:.3f} | {calls:.1f} | {w} | {a} | {fpk} | {tr} | {ratio} | {bz} | {es} |")
    return "\n".join(out) + "\n"


doc = [f"# {rnd}: per-kernel evidence table of the Bloom-560M SFT step (B = 8, S = 1024, bf16, one MI355X)", "",
       "Produced by `tools/kernel_table.py` from ONE `tools/collect_profiles.sh` collection (same box, same build).  Columns: kernel time from "
       "`rocprofv3 --kernel-trace --stats` (per-step = total ÷ the steps the trace covers = launches of the LM-head forward); algorithmic work from the model's shapes; *achieved* = work ÷ kernel time; "
       "*of chip peak* against 2.5 PFLOP/s dense bf16 or 8 TB/s; FETCH + WRITE from separate counter-only passes over the same command (FETCH doubled per the gfx950 note; "
       "Infinity-Cache hits are included — the counters sit at the L2 boundary), *vs algorithmic bytes* = that ÷ the bytes the kernel has to move at least; "
       "*MFMA busy* = `SQ_VALU_MFMA_BUSY_CYCLES ÷ (GRBM_GUI_ACTIVE / 8 × 1024)` (share of SIMD cycles with the matrix pipe executing, at the shader clock of the "
       "counted run); the energy column is `tools/energy_probe.py` (package power beside a ≥ 1 s back-to-back loop of ONE launch of that kind, warm operands).", ""]
doc.append(table("prof_bench", "Default build: grouped weight gradients, the whole step on one stream"))
if one("prof_bench_1stream/*/*_kernel_stats.csv"):
    doc.append(table("prof_bench_1stream", "CTMI_WGRAD_GROUP=0: four weight-gradient products per block on a side stream (the round-4 form; kernels overlap, so times are inflated by sharing)"))
open(os.path.join(dst, f"{rnd}_kernel_table.md"), "w").write("\n".join(doc))
print("wrote", os.path.join(dst, f"{rnd}_kernel_table.md"))
