#!/usr/bin/env python3
"""ISA lint for the LDS-DMA kernels: no compiler-made `s_waitcnt vmcnt(0)` inside a steady K-loop.

Why (round 4): the LDS-DMA pieces (`global_load_lds_dwordx4`) are inline asm, so hipcc neither counts nor waits for them — the kernels carry
their own counted `s_waitcnt vmcnt(N)`.  But the compiler's wait-count pass still protects ITS OWN vector-memory operations (epilogue loads),
and when it believes one of them may still be pending at a loop head it inserts `s_waitcnt vmcnt(0)` there — which also waits for every
LDS-DMA piece in flight.  Rounds 1-3 shipped that wait at the head of the steady K-loop of every ping-pong kernel with the cross-lane epilogue
(all 128-row tiles, the XLANE 256-row tiles — ~190 of the 290 GEMM launches of a step): every K-step drained the three-stage ring.  It is
invisible in the source and in any parity test; this tool makes it visible.

For every kernel of csrc/gemm.hip (six translation-unit parts: three bf16 families, three fp16 families) and csrc/attention_w32.hip: compile to assembly (device only), find the
innermost loops that contain >= 8 MFMAs and >= 1 LDS-DMA instruction in <= 400 lines (the steady loops), and report every `s_waitcnt` with
`vmcnt(0)` in them that is NOT inside an inline-asm region — and every hand-written (inline-asm) `vmcnt(N)` of a steady loop whose N is not a
multiple of the pieces the loop issues per trip (a miscounted immediate).  Exit status 1 if there is any.

    python tools/kernel_isa_scan.py            # all kernels
    python tools/kernel_isa_scan.py --one "bf16_t, false, true, 0, 4, 4, true, false, false" [-DFLAG=1 ...]   # one GEMM instantiation (seconds)
    python tools/kernel_isa_scan.py --file some.s                                                              # an assembly file that already exists
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cleantransformer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value", "--offload-device-only", "-S"]


def hipcc():
    import shutil
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise SystemExit("hipcc not found")


def compile_s(src, out, extra):
    r = subprocess.run([hipcc(), *FLAGS, *extra, os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("hipcc failed on " + src + ":\n" + r.stderr[-3000:])
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")


def scan(path):
    """-> [(kernel, [(loop start line, loop length, mfmas, dma pieces, [offending line numbers])])] for the steady loops of every kernel."""
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = demangle([n for _, n in starts])
    starts.append((len(lines), None))
    out = []
    for k, ((a, _), (b, _)) in enumerate(zip(starts, starts[1:])):
        labels = {}
        for i in range(a, b):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                labels[m.group(1)] = i
        loops = []
        for i in range(a, b):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        res = []
        for (s, e) in loops:
            if e - s > 400 or any(s < s2 and e2 < e for (s2, e2) in loops):          # not compact / not innermost
                continue
            body = lines[s:e + 1]
            nm = sum("v_mfma" in x for x in body)
            nd = sum(("global_load_lds" in x) or ("buffer_load" in x and " lds" in x) for x in body)
            if nm < 8 or nd < 1:
                continue
            in_asm, bad, counted = False, [], []
            for j, x in enumerate(body):
                if "ASMSTART" in x:
                    in_asm = True
                elif "ASMEND" in x:
                    in_asm = False
                elif not in_asm and re.search(r"s_waitcnt\b.*vmcnt\(0\)", x):
                    bad.append(s + j + 1)
                elif in_asm:
                    m = re.search(r"s_waitcnt\b.*vmcnt\((\d+)\)", x)
                    if m:
                        counted.append(int(m.group(1)))
            res.append((s + 1, e - s, nm, nd, bad, counted))
        out.append((names[k].replace("unsigned short", "bf16"), res))
    return out


def main(argv):
    tmp = tempfile.mkdtemp(prefix="ctmi_isa_")
    if argv and argv[0] == "--file":
        jobs = None
        files = list(argv[1:])
    elif argv and argv[0] == "--one":
        extra = ["-DCTMI_GEMM_PART=9", f"-DCTMI_ONE_KERNEL={argv[1]}", *argv[2:]]
        jobs = [("gemm.hip", os.path.join(tmp, "one.s"), extra)]
    else:
        jobs = [("gemm.hip", os.path.join(tmp, f"gemm_p{p}.s"), [f"-DCTMI_GEMM_PART={p}", *argv]) for p in (1, 2, 3, 4, 5, 6)]
        jobs.append(("attention_w32.hip", os.path.join(tmp, "attention_w32.s"), ["-fno-slp-vectorize", *argv]))
    if jobs is not None:
        with ThreadPoolExecutor(max_workers=4) as ex:
            files = list(ex.map(lambda j: compile_s(*j), jobs))
    # Loops of <= 200 lines are the steady K-loops proper (ping-pong: ~100-135 lines).  The free-running schedule keeps its work-item switch
    # (64-bit address set-up, run once per output tile) inside the K-loop's body (~380 lines): a wait there drains the ring once per tile, not
    # once per K-step — reported as a note, not an error.
    nbad = nnote = nloops = 0
    seen = set()
    for f in files:
        for kern, loops in scan(f):
            for (s, n, nm, nd, bad, counted) in loops:
                nloops += 1
                # the hand-counted wait of a steady loop leaves a WHOLE number of trips' pieces in flight (each trip issues `nd` pieces): an
                # immediate that is not a multiple of nd is a miscount (the side-piece experiments, built with their flags, are the exception)
                odd = [c for c in counted if n <= 200 and c > 0 and c % nd]
                if odd and (f, s) not in seen:
                    seen.add((f, s))
                    nbad += 1
                    print(f"BAD  {kern}: steady loop at {os.path.basename(f)}:{s} issues {nd} LDS-DMA pieces per trip but waits with vmcnt{odd}")
                key = (f, tuple(bad))
                if bad and key not in seen:
                    seen.add(key)
                    steady = n <= 200
                    nbad += steady
                    nnote += not steady
                    print(f"{'BAD ' if steady else 'note'} {kern}: {'steady loop' if steady else 'K-loop with in-body tile switch'} at {os.path.basename(f)}:{s} "
                          f"({n} lines, {nm} MFMAs, {nd} LDS-DMA pieces) has compiler-made vmcnt(0) at lines {bad}")
    print(f"{nloops} loops scanned in {len(files)} assembly files ({tmp}): {nbad} steady loops flagged (compiler-made s_waitcnt vmcnt(0) / miscounted hand-written wait), {nnote} notes")
    return 1 if nbad or not nloops else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
