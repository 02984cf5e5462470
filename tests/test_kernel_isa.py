"""CPU (no GPU needed; hipcc cross-compiles): ISA lint of the LDS-DMA kernels — tools/kernel_isa_scan.py.

No steady K-loop may contain a compiler-made `s_waitcnt vmcnt(0)`: the LDS-DMA pieces are inline asm that hipcc does not count, so a wait it
inserts for one of ITS loads also drains every DMA piece in flight.  Rounds 1-3 shipped exactly that in every ping-pong kernel with the
cross-lane epilogue (the three-stage ring was one stage deep); no parity test can see it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_compiler_made_vmcnt0_in_a_steady_k_loop():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_isa_scan.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 steady loops flagged" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_lint_sees_the_round3_structure():
    """the same kernel built the way rounds 1-3 built it (-DCTMI_PP_FAST_NONPLAIN=1) must be flagged: the lint is not vacuous"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_isa_scan.py"), "--one", "bf16_t, false, true, 0, 4, 4, true, false, false",
                        "-DCTMI_PP_FAST_NONPLAIN=1", "-DCTMI_PP_SIDE_LDS=0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "BAD " in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_lint_sees_a_wait_that_is_not_a_whole_number_of_trips():
    """second class (round 4): a steady loop's hand-written `vmcnt(N)` must leave a whole number of trips' LDS-DMA pieces in flight.  The
    side-input-through-LDS experiment (one extra piece every third K-step, waits spelled vmcnt(7)) is the build that breaks the rule on purpose."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_isa_scan.py"), "--one", "bf16_t, false, true, 2, 4, 4, true, false, false",
                        "-DCTMI_PP_SIDE_LDS=1", "-DCTMI_PP_K2=0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "pieces per trip but waits with vmcnt" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
