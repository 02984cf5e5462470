#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in libctmi355.so, read from the code objects' AMDGPU metadata (no GPU needed).

usage: python tools/kernel_resources.py [pattern ...]        (demangled-name substrings; default: the hot kernels)

The numbers decide occupancy on gfx950 (512 VGPRs per SIMD lane: <= 168 -> 3 waves/SIMD, <= 256 -> 2, above -> 1), which is what
several of the A/B results in DESIGN.md hinge on (the attention kernels' third wave, the ping-pong GEMM's two).  tests/
test_kernel_resources.py pins the budgets of the kernels on the measured path so a change that costs a wave fails on the CPU box.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
          "private_segment_fixed_size", "max_flat_workgroup_size")


def have_tools() -> bool:
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")) and shutil.which("c++filt") is not None


def read(lib_path: str):
    """-> {demangled kernel name: {field: int}} for every kernel of every gfx950 code object bundled in lib_path."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if line.lstrip().startswith("- .") and k == "agpr_count" and cur.get("name"):
                    out[cur["name"]] = cur                                    # (a kernel record starts with "- .agpr_count")
                    cur = {}
                if k == "name":
                    cur["name"] = v
                elif k in FIELDS:
                    cur[k] = int(v)
            if cur.get("name"):
                out[cur["name"]] = cur
    names = list(out)
    dem = subprocess.run([shutil.which("c++filt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return {d: out[n] for n, d in zip(names, dem)}


def waves_per_simd(k) -> int:
    regs = k.get("vgpr_count", 0)                                             # on gfx950 vgpr_count is the unified (arch + acc) allocation
    regs = (regs + 7) // 8 * 8
    return max(1, min(8, 512 // max(regs, 1)))


def main():
    pats = sys.argv[1:] or ["gemm_glds_kernel<unsigned short", "gemm_glds_kernel<float, true, true", "attn_", "ce_fused", "ln_bwd_vec", "ln_fwd_vec", "adamw"]
    from cleantransformer_amd import _lib
    ks = read(_lib.LIB_PATH)
    print(f"{'VGPR':>5} {'AGPR':>5} {'w/SIMD':>6} {'spillV':>6} {'spillS':>6} {'LDS':>7}  kernel")
    for name in sorted(ks):
        if not any(p in name for p in pats):
            continue
        k = ks[name]
        print(f"{k.get('vgpr_count', 0):5d} {k.get('agpr_count', 0):5d} {waves_per_simd(k):6d} {k.get('vgpr_spill_count', 0):6d} "
              f"{k.get('sgpr_spill_count', 0):6d} {k.get('group_segment_fixed_size', 0):7d}  {name[:150]}")


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
