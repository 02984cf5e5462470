"""Turn the raw rocprofv3 CSVs of tools/collect_profiles.sh into the small summaries committed under profiles/.

usage: python tools/summarize_profiles.py gpurun_out/evidence profiles r01
"""
import csv
import glob
import json
import os
import sys

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
LM = "gemm_glds_kernel<unsigned short, false, false, 0, 8, 4, true, false, false>"     # bf16 NT, no epilogue, 256x256 ping-pong, LDS-patch epilogue


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)     # newest run if several were merged back


def kernel_table(run_dir, steps_traced, title, out_name):
    rows = list(csv.DictReader(open(one(f"{run_dir}/*/*_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [title, f"(rocprofv3 --kernel-trace --stats; {steps_traced} steps traced incl. warm-up; per-step = total / {steps_traced})",
             f"sum of kernel time: {tot / 1e6 / steps_traced:.2f} ms/step", ""]
    lines.append(f"{'ms/step':>8} {'calls/step':>10} {'avg us':>9}  kernel")
    for r in rows[:40]:
        lines.append(f"{float(r['TotalDurationNs']) / 1e6 / steps_traced:8.3f} {int(r['Calls']) / steps_traced:10.1f} {float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}")
    tr = list(csv.DictReader(open(one(f"{run_dir}/*/*_kernel_trace.csv"))))
    lm = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in tr if r["Kernel_Name"].startswith("void " + LM)]
    lm = [d for d in lm if d > 1e6]                                     # the [T,1024]x[250880,1024]^T launches (the QKV ones are ~65 us)
    lines += ["", f"LM-head forward launches of {LM}: {len(lm)} launches, average {sum(lm) / len(lm) / 1e6:.4f} ms "
              f"(= {2 * 8192 * 250880 * 1024 / (sum(lm) / len(lm)) / 1e3:.1f} TFLOP/s); bench.py's live HIP-event figure for the same run is in "
              f"{rnd}_bench_under_rocprof.json"]
    open(os.path.join(dst, out_name), "w").write("\n".join(lines) + "\n")
    return sum(lm) / len(lm) / 1e6


def pmc(counter):
    rows = list(csv.DictReader(open(one(f"pmc_{counter}/*/*_counter_collection.csv"))))
    v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith("void " + LM) and r["Counter_Name"] == counter]
    return v


def traced_steps(run_dir):
    """steps a trace covers = launches of the LM-head forward (one per step): 3 warm-up + 2 idle-queue host timing + 10 timed (+ 10 beside the
    power sampler since round 5)"""
    tr = csv.DictReader(open(one(f"{run_dir}/*/*_kernel_trace.csv")))
    n = sum(1 for r in tr if r["Kernel_Name"].startswith("void " + LM) and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) > 1e6)
    return max(n, 1)


steps = traced_steps("prof_bench")
a = kernel_table("prof_bench", steps, f"{rnd}: python bench.py --no-cpu-baseline (default since round 5: grouped weight gradients, the whole step on ONE stream: clean per-kernel durations)",
                 f"{rnd}_bench_kernel_summary.txt")
import shutil as _sh
_sh.copy(os.path.join(dst, f"{rnd}_bench_kernel_summary.txt"), os.path.join(dst, f"{rnd}_bench_1stream_kernel_summary.txt"))   # (the name rounds 1-4 used for the single-stream trace)
b = kernel_table("prof_bench_1stream", traced_steps("prof_bench_1stream"), f"{rnd}: CTMI_WGRAD_GROUP=0 python bench.py --no-cpu-baseline (the round-4 form: four weight-gradient products per block on a side stream, kernels overlap)",
                 f"{rnd}_bench_per_product_kernel_summary.txt")
fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
traffic = {
    "kernel": LM + " = LM-head forward [8192,1024]x[250880,1024]^T",
    "launches_profiled": [len(fetch), len(write)],
    "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk,
    "fetch_bytes_corrected": fk * 1024 * 2, "write_bytes": wk * 1024,
    "traffic_bytes_per_launch": fk * 1024 * 2 + wk * 1024,
    "algorithmic_bytes_per_launch": 8192 * 1024 * 2 + 250880 * 1024 * 2 + 8192 * 250880 * 2,
    "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) on MB_ONLY=lm_head MB_FWD_ONLY=1 tools/microbench.py gemm; "
            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B). "
            "Algorithmic bytes: A 16.8 MB + B 514 MB read, C 4.11 GB written (non-temporal). Reads exceed A+B because each of the 8 XCDs "
            "(private 4 MiB L2) streams the weight panel for each group of 4 tile rows; they are served by the 256 MiB Infinity Cache.",
}
json.dump(traffic, open(os.path.join(dst, f"{rnd}_lmhead_traffic.json"), "w"), indent=1)
# HBM bytes per step over ALL kernels (PMC passes over the bench command itself: 15 steps traced)
def step_counter(counter):
    rows = list(csv.DictReader(open(one(f"pmc_step_{counter}/*/*_counter_collection.csv"))))
    per = {}
    for r in rows:
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0][:90] if not r["Kernel_Name"].startswith("void (anonymous") else r["Kernel_Name"][:90]
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    return per
try:
    fs, ws = step_counter("FETCH_SIZE"), step_counter("WRITE_SIZE")
    fetch_b = sum(fs.values()) * 1024 * 2 / steps
    write_b = sum(ws.values()) * 1024 / steps
    top = sorted(((fs.get(k, 0.0) * 2048 + ws.get(k, 0.0) * 1024) / steps, k) for k in set(fs) | set(ws))[::-1][:12]
    json.dump({"command": f"python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3 ({steps} steps traced incl. warm-up and the power-sampling steps)",
               "fetch_bytes_per_step": fetch_b, "write_bytes_per_step": write_b, "traffic_bytes_per_step": fetch_b + write_b,
               "note": "separate --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B "
                       "requests of wide coalesced reads at 64 B); fetches served by the Infinity Cache are included (the counters sit at the L2 boundary)",
               "top_kernels_bytes_per_step": [{"kernel": k, "bytes": b} for b, k in top]},
              open(os.path.join(dst, f"{rnd}_step_traffic.json"), "w"), indent=1)
    print("step traffic: %.1f GB/step (fetch %.1f, write %.1f)" % ((fetch_b + write_b) / 1e9, fetch_b / 1e9, write_b / 1e9))
except Exception as e:                                                   # (older collections have no step passes)
    print("no step-traffic passes in this collection:", e)

# the plain result files of the collection travel as they are
import shutil
for name in (f"{rnd}_bench_default.json", f"{rnd}_bench_under_rocprof.json", f"{rnd}_bench_per_product_under_rocprof.json", f"{rnd}_energy_probe.txt", f"{rnd}_chain_probe.txt", f"{rnd}_pmc_lmhead_mfma.txt",
             f"{rnd}_microbench.txt", f"{rnd}_vendor_gemm_reference.txt", f"{rnd}_microbench_epilogues.txt", f"{rnd}_attention_paths.txt",
             f"{rnd}_gpt2_medium_bench.txt", f"{rnd}_bloom7b1_1gpu_bench.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
shutil.copy(one("prof_bench/*/*_kernel_stats.csv"), os.path.join(dst, f"{rnd}_bench_kernel_stats.csv"))
print("LM-head fwd avg ms: default run %.4f, per-product two-stream run %.4f; traffic %.2f GB/launch" % (a, b, traffic["traffic_bytes_per_launch"] / 1e9))
