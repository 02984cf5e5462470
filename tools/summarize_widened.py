"""Kernel summaries + HBM traffic of the widened configs (GPT-2 medium = BASELINE configs[3]; Bloom-7B1 geometry on one GPU = configs[4]) from the raw
rocprofv3 output of tools/batches/r6_batch7.sh.  usage: python tools/summarize_widened.py gpurun_out/r6b7 profiles r06"""
import collections
import csv
import glob
import json
import os
import sys

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]


def one(pat):
    return max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)


def last_step(trace_csv, marker):
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    # optimizer launches come in runs (one run per step): the last step is what lies between the end of the previous run and the end of the last one
    runs = []
    for i in idx:
        if runs and i - runs[-1][-1] <= 2:
            runs[-1].append(i)
        else:
            runs.append([i])
    s, e = runs[-2][-1] + 1, runs[-1][-1]
    return rows[s:e + 1], len(runs)


def counter_per_step(pat, counter, marker):
    rows = list(csv.DictReader(open(one(pat))))
    rows = [r for r in rows if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    runs = []
    for i in idx:
        if runs and i - runs[-1][-1] <= 2:
            runs[-1].append(i)
        else:
            runs.append([i])
    s, e = runs[-2][-1] + 1, runs[-1][-1]
    per = collections.defaultdict(float)
    for r in rows[s:e + 1]:
        per[r["Kernel_Name"][:110]] += float(r["Counter_Value"])
    return per


def table(tag, title, flops_per_step, marker="adamw_flat_k"):
    step, nsteps = last_step(one(f"prof_{tag}/*/*_kernel_trace.csv"), marker)
    span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        a = agg[r["Kernel_Name"][:110]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fetch = counter_per_step(f"pmc_{tag}_FETCH_SIZE/*/*_counter_collection.csv", "FETCH_SIZE", marker)
    write = counter_per_step(f"pmc_{tag}_WRITE_SIZE/*/*_counter_collection.csv", "WRITE_SIZE", marker)
    tf = sum(fetch.values()) * 1024 * 2
    tw = sum(write.values()) * 1024
    lines = [title,
             f"(rocprofv3 --kernel-trace; the LAST traced step of {nsteps}: from the end of the previous optimizer step to the end of this one; traffic from separate "
             f"--pmc FETCH_SIZE / WRITE_SIZE passes over the same command, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md)",
             f"step span {span:.2f} ms, sum of kernel time {busy:.2f} ms ({'kernels overlap: weight gradients on the side stream' if busy > 1.02 * span else 'one stream'}), "
             f"{len(step)} launches; algorithmic {flops_per_step / 1e12:.1f} TFLOP per step = {flops_per_step / span / 1e9:.0f} TF/s = {flops_per_step / span / 1e9 / 2500 * 100:.1f} % of 2.5 PF",
             f"HBM traffic of the step: FETCH {tf / 1e9:.1f} GB + WRITE {tw / 1e9:.1f} GB = {(tf + tw) / 1e9:.1f} GB = {(tf + tw) / span / 1e9:.2f} TB/s average", "",
             f"{'ms/step':>9} {'launches':>8} {'avg us':>9} {'FETCH GB':>9} {'WRITE GB':>9}  kernel"]
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:32]:
        lines.append(f"{t / 1e3:9.3f} {c:8d} {t / c:9.1f} {fetch.get(n, 0.0) * 2048 / 1e9:9.2f} {write.get(n, 0.0) * 1024 / 1e9:9.2f}  {n}")
    open(os.path.join(dst, f"{rnd}_{tag}_kernel_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:6]))


L, H, V, S, T = 24, 1024, 50257, 2048, 8192
table("gpt2", f"{rnd}: python tools/bench_gpt2.py — GPT-2 medium (24L, n_embd 1024, 16 heads, V 50257) LM training step, B=4 S=2048, bf16 (BASELINE configs[3])",
      (6.0 * (L * 12 * H * H + V * H) + 6.0 * L * S * H) * T)
L, H, V, S, T = 30, 4096, 250880, 2048, 4096
table("7b1", f"{rnd}: python tools/bench_bloom7b1.py — Bloom-7B1 geometry (30L, H 4096, 32 heads, V 250880) SFT step on ONE GPU, B=2 S=2048, bf16 (BASELINE configs[4] geometry)",
      (6.0 * (L * 12 * H * H + V * H) + 6.0 * L * S * H) * T)
