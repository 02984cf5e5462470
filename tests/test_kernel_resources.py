"""Register budgets of the kernels on the measured path, read from the built library's code-object metadata (no GPU needed).

Several results in DESIGN.md hinge on occupancy: the attention kernels run three waves per SIMD only up to 168 VGPRs (the mask-free
backward body and the scalar-origin tile loads were adopted / rejected per kernel on exactly that), the ping-pong GEMM tiles need
two waves per SIMD (<= 256), the 128 x 128 GEMM tile three.  A change that silently costs a wave, or a kernel whose accumulators
start spilling (the rejected 128 x 128-wave-tile GEMM: 350 - 480 spilled VGPRs), shows up here on the CPU box at build time.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    from cleantransformer_amd import _lib
    if not KR.have_tools():
        pytest.skip("llvm-objdump / llvm-readelf / c++filt not available")
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libctmi355.so not built")
    return KR.read(_lib.LIB_PATH)


def pick(kernels, prefix):
    out = {n: k for n, k in kernels.items() if n.startswith(prefix)}
    assert out, f"no kernel named {prefix}..."
    return out


@pytest.mark.parametrize("kernel", ["attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkdv_kernel"])
def test_attention_training_kernels_keep_three_waves_per_simd(kernels, kernel):
    # bf16, head_dim 64, no additive mask, FAST (aligned) instantiation, no dropout: what Bloom-560M / GPT-2-medium training launches
    (name, k), = pick(kernels, f"void {kernel}<unsigned short, 64, false, true, false>").items()
    assert k["vgpr_count"] <= 168, (name, k["vgpr_count"])
    assert k["vgpr_spill_count"] == 0, (name, k["vgpr_spill_count"])


def test_attention_128row_kernels_keep_their_occupancy_without_scratch(kernels):
    """csrc/attention_w32.hip (the measured path since round 3): forward and dQ at three workgroups per CU (<= 168 VGPRs), dK/dV and
    the head_dim-128 forward at two (<= 256), and NO scratch anywhere — a scratch reload in the tile loop comes with
    s_waitcnt vmcnt(0) and drains the LDS-DMA prefetch (the first head_dim-64 forward did exactly that with 5 spilled dwords)."""
    w32 = {n: k for n, k in kernels.items() if "::attn32_" in n}          # (anonymous namespace)
    assert len(w32) >= 6, sorted(w32)
    for name, k in w32.items():
        cap = 168 if ("attn32_dq_kernel<64" in name or "attn32_fwd_kernel<64" in name) else 256
        assert k["vgpr_count"] <= cap, (name, k["vgpr_count"])
        assert k["vgpr_spill_count"] == 0 and k.get("private_segment_fixed_size", 0) == 0, (name, k)


def test_attention_head_dim_128_fast_kernels_do_not_spill(kernels):
    for kernel in ("attn_fwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkdv_kernel"):
        (name, k), = pick(kernels, f"void {kernel}<unsigned short, 128, false, true, false>").items()
        assert k["vgpr_count"] <= 256 and k["vgpr_spill_count"] == 0, (name, k)


def test_gemm_kernels_fit_their_occupancy(kernels):
    """Round 5: with the ablation / timing / experiment branches out of csrc/gemm.hip NO LDS-DMA GEMM kernel spills (round 4 shipped the
    256-row ping-pong tiles at 256 VGPRs + up to 31 spilled), and every tile keeps headroom below its occupancy step."""
    gemms = pick(kernels, "void gemm_glds_kernel<")
    gemms.update(pick(kernels, "void gemm_glds_kernel_f16<"))                  # (round 5: the fp16 twins of the same schedules)
    gemms.update(pick(kernels, "gemm_wgrad_grouped_kernel"))
    for name, k in gemms.items():
        # two waves per SIMD (eight-wave tiles: one workgroup per CU).  Round 6: the 256-row ping-pong tile of the dGELU data gradient prefetches
        # its side input through one register slot — 244, still without scratch (two slots, as on the 128-row tile, needed 260 = 256 + 4 spilled)
        cap = 248 if ", true, 2, 8, 4, true, false, false>" in name else 240
        assert k["vgpr_count"] <= cap, (name, k["vgpr_count"])
        assert k["agpr_count"] == 0, (name, k["agpr_count"])
        assert k["vgpr_spill_count"] == 0 and k.get("private_segment_fixed_size", 0) == 0, (name, k)
    # the 128 x 128 free-running tile: three workgroups per CU
    for name, k in gemms.items():
        if ", 4, 2, false, false, false" in name:
            assert k["vgpr_count"] <= 168, (name, k)
    # the 128-row ping-pong tile (two ring stages per phase): the forward / data-gradient / weight-gradient instantiations
    for name, k in gemms.items():
        if ", 4, 4, true, " in name:
            assert k["vgpr_count"] <= 200, (name, k)


def test_streaming_kernels_do_not_spill(kernels):
    for prefix in ("void ce_fused_k<unsigned short>", "adamw_mt_k", "void ln_fwd_vec<unsigned short", "void ln_bwd_vec<unsigned short",
                   "reduce_jobs_k", "void colsum_part<unsigned short>", "void embed_bwd_k<unsigned short>"):
        for name, k in pick(kernels, prefix).items():
            assert k["vgpr_spill_count"] == 0 and k.get("private_segment_fixed_size", 0) == 0, (name, k)


def _disassemble(lib_path, wanted):
    """{demangled name: disassembly text} of the kernels whose demangled name starts with one of `wanted` (llvm-objdump on the gfx950 code objects)"""
    import shutil
    import subprocess
    import tempfile
    llvm = KR.LLVM
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-s", "-W", os.path.join(tmp, f)], capture_output=True, text=True).stdout.split("\n")
            mangled = [ln.split()[-1] for ln in syms if " FUNC " in ln]
            if not mangled:
                continue
            dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.split("\n")
            for m, d in zip(mangled, dem):
                if any(d.startswith(w) for w in wanted):
                    r = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", f"--disassemble-symbols={m}", os.path.join(tmp, f)],
                                       capture_output=True, text=True)
                    out[d] = r.stdout
    return out


def test_streamed_bf16_outputs_are_written_through_and_fp32_gradients_are_not():
    """Round 6 (csrc/common.h st_wt16; profiles/r06_boundary_dirty.txt): the 16-byte stores of LayerNorm, of the 128-row attention kernels and of the
    bf16 GEMM epilogues carry `sc1` (the dependent launch does not wait for the producer's dirty L2 lines); the fp32 weight gradients of the grouped
    launch are plain stores (written through they were 9 % slower), the GELU pre-activation and the logits non-temporal.  A build flag or a refactor
    that silently drops the policy shows up here on the CPU box."""
    from cleantransformer_amd import _lib
    if not KR.have_tools() or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("tools / library not available")
    wanted = ["void ln_fwd_vec<unsigned short, 2>", "void ln_bwd_vec<unsigned short, 2, 4, true, 1>", "void (anonymous namespace)::attn32_fwd_kernel<64, 4, false, false>",
              "void gemm_glds_kernel<unsigned short, false, false, 0, 8, 4, true, false, true>", "void gemm_glds_kernel<unsigned short, false, false, 1, 8, 4, true, false, true>",
              "gemm_wgrad_grouped_kernel(GroupedArgs)"]
    dis = _disassemble(_lib.LIB_PATH, wanted)
    assert len(dis) == len(wanted), sorted(dis)

    def stores(name, flag):
        txt = next(v for k, v in dis.items() if k.startswith(name))
        return sum(1 for ln in txt.split("\n") if "global_store_dwordx4" in ln and (flag in ln.split("//")[0].split()))
    for name in wanted[:5]:
        assert stores(name, "sc1") > 0, name
    assert stores("gemm_wgrad_grouped_kernel", "sc1") == 0
    assert stores("void gemm_glds_kernel<unsigned short, false, false, 1, 8, 4", "nt") > 0                  # the GELU pre-activation stays non-temporal
