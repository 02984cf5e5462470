"""CPU (no GPU needed; hipcc cross-compiles): ISA lint of the LDS-DMA kernels — tools/kernel_isa_scan.py.

No steady K-loop may contain a compiler-made `s_waitcnt vmcnt(0)`: the LDS-DMA pieces are inline asm that hipcc does not count, so a wait it
inserts for one of ITS loads also drains every DMA piece in flight.  Rounds 1-3 shipped exactly that in every ping-pong kernel with the
cross-lane epilogue (the three-stage ring was one stage deep); no parity test can see it.

The two self-tests feed the scanner hand-written assembly with the two defect classes it exists for (round 5: the source no longer carries
the `-DCTMI_PP_FAST_NONPLAIN=1` / `-DCTMI_PP_SIDE_LDS=1` builds that used to serve as the bad examples — tools/experiments/ has them as a patch)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, "tools", "kernel_isa_scan.py")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_compiler_made_vmcnt0_in_a_steady_k_loop():
    r = subprocess.run([sys.executable, SCAN], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 steady loops flagged" in r.stdout, r.stdout[-2000:]
    # the grouped weight-gradient instantiation (round 5) is among the kernels that were scanned
    assert "loops scanned" in r.stdout


def _steady_loop(wait_lines, pieces=3, mfmas=16):
    """assembly text of one kernel with one steady loop: `pieces` LDS-DMA instructions in an inline-asm region, `mfmas` matrix instructions,
    and the given wait lines (each a (text, inside_inline_asm) pair) in between"""
    out = ["\t.text", "_Z9fake_gemmv:", "\ts_load_dwordx2 s[0:1], s[4:5], 0x0", ".LBB0_1:"]
    out += ["\tds_read_b128 v[0:3], v100"] * 8
    out += ["\t;;#ASMSTART"] + ["\tglobal_load_lds_dwordx4 v[10:11], off"] * pieces + ["\t;;#ASMEND"]
    for text, in_asm in wait_lines:
        out += (["\t;;#ASMSTART", "\t" + text, "\t;;#ASMEND"] if in_asm else ["\t" + text])
    out += ["\ts_barrier"] + ["\tv_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[4:7], v[20:23]"] * mfmas + ["\ts_barrier"]
    out += ["\ts_add_i32 s2, s2, -1", "\ts_cmp_lg_u32 s2, 0", "\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm", ""]
    return "\n".join(out)


def _scan_text(tmp_path, text):
    f = tmp_path / "fake.s"
    f.write_text(text)
    return subprocess.run([sys.executable, SCAN, "--file", str(f)], capture_output=True, text=True, timeout=120)


def test_the_lint_passes_a_correct_steady_loop(tmp_path):
    r = _scan_text(tmp_path, _steady_loop([("s_waitcnt vmcnt(6)", True), ("s_waitcnt lgkmcnt(0)", True)]))
    assert r.returncode == 0 and " 0 steady loops flagged" in r.stdout, r.stdout + r.stderr


def test_the_lint_sees_a_compiler_made_drain(tmp_path):
    """the round-1-3 structure: hipcc's own `s_waitcnt vmcnt(0)` (outside any inline-asm region) in the steady loop — the lint is not vacuous"""
    r = _scan_text(tmp_path, _steady_loop([("s_waitcnt vmcnt(0)", False), ("s_waitcnt vmcnt(6)", True)]))
    assert r.returncode == 1 and "BAD " in r.stdout and "compiler-made vmcnt(0)" in r.stdout, r.stdout + r.stderr


def test_the_lint_sees_a_wait_that_is_not_a_whole_number_of_trips(tmp_path):
    """second class (round 4): a steady loop's hand-written `vmcnt(N)` must leave a whole number of trips' LDS-DMA pieces in flight (the
    side-input-through-LDS experiment — one extra piece every third K-step, waits spelled vmcnt(7) — was the build that broke the rule on purpose)"""
    r = _scan_text(tmp_path, _steady_loop([("s_waitcnt vmcnt(7)", True)]))
    assert r.returncode == 1 and "pieces per trip but waits with vmcnt" in r.stdout, r.stdout + r.stderr
