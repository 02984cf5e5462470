"""Build libctmi355.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache, no fallbacks."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libctmi355.so")
# (source, object, extra flags): gemm.hip is built as seven translation units (fp32 + entry point, three bf16 families, three fp16 families) so its template instantiations compile in parallel
SOURCES = [("elementwise.hip", "elementwise.o", []), ("attention.hip", "attention.o", []), ("attention_w32.hip", "attention_w32.o", ["-fno-slp-vectorize"]), ("probe.hip", "probe.o", []), ("prof.hip", "prof.o", []), ("comm.hip", "comm.o", []), ("block.hip", "block.o", [])] + \
          [("gemm.hip", f"gemm_p{i}.o", [f"-DCTMI_GEMM_PART={i}"]) for i in range(7)]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ctmi355 kernels can only be built with the ROCm toolchain")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, extra_flags, verbose: bool = True) -> str:
    """A/B builds for kernel experiments (tools/): the same sources with extra -D flags, into lib/variants/<name>/.  The product
    never loads these unless CTMI_LIB_PATH points at one (see _lib.py)."""
    vdir = os.path.join(LIB_DIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for src, obj, extra in SOURCES:
        o = os.path.join(vdir, obj)
        objs.append(o)
        jobs.append([hipcc, *FLAGS, *extra, *extra_flags, "-c", os.path.join(CSRC, src), "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    out = os.path.join(vdir, "libctmi355.so")
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    for o in objs:
        os.remove(o)
    if verbose:
        print("[ctmi355 variant]", name, extra_flags, "->", out)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "ctmi355.h"))
    objs, jobs = [], []
    for src, obj, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, obj)
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[ctmi355 build]", " ".join(cmd[-5:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv))
