"""GPU (-m gpu): the grouped weight-gradient launch (include/ctmi355.h ctmi_wgrad_grouped, ABI v13; csrc/gemm.hip gemm_wgrad_grouped_kernel) —
the autograd of the four Linears of a block (modeling_bloom.py:79,121,256,267) in one persistent launch, bias column sums included.

Checked against (a) an fp64 product of the bf16 operands (sampled where the matrices are large), (b) the per-product kernels, (c) itself:
bit-identical from run to run — the K-halves of the last partial round go through partial slabs and are added in a fixed order —
and bit-identical with operands evicted from every cache (the LDS-DMA ring's race detector, as for the single-problem GEMMs)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from cleantransformer_amd import ops as o
    return o


def lib():
    from cleantransformer_amd import _lib
    return _lib


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def _problems(T, shapes, seed, scale=0.5, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    bfr = lambda *sh: (torch.randn(*sh, generator=g, device=DEV) * scale).to(dtype)   # noqa: E731
    return [(bfr(T, n_out), bfr(T, n_in), want_db) for (n_out, n_in, want_db) in shapes]


def _check_fp64(name, dy, x, dw, db, in_out=False, sample=48):
    """dw (and db) against fp64 on a sample of rows / columns that covers every tile row / column boundary region"""
    T, n_out = dy.shape
    n_in = x.shape[1]
    g = torch.Generator().manual_seed(n_out * 31 + n_in)

    def pick(n):
        if n <= sample:
            return torch.arange(n)
        edge = torch.tensor([0, 1, 127, 128, 255, 256, n - 257, n - 129, n - 128, n - 1]).clamp_(0, n - 1)
        return torch.unique(torch.cat([edge, torch.randint(0, n, (sample,), generator=g)]))
    ro, ci = pick(n_out).to(DEV), pick(n_in).to(DEV)
    ref = dy[:, ro].double().t() @ x[:, ci].double()
    got = (dw.t() if in_out else dw)[ro][:, ci].double()
    # fp32 accumulation of T products of size ~sigma^2: error ~ 1e-7 * sqrt(T) * sigma^2 per element; bound 2e-5 of the row scale (as the single-problem test)
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max())
    assert err <= 2e-5 * scale * max(1.0, (T / 8192) ** 0.5) + 1e-6, (name, err, scale)
    if db is not None:
        rb = dy.double().sum(0)
        eb = float((db.double() - rb).abs().max())
        assert eb <= 2e-5 * (float(rb.abs().max()) + 1e-30) + 1e-5, (name, "bias gradient", eb)


BLOOM = [(1024, 4096, False), (4096, 1024, True), (1024, 1024, False), (3072, 1024, True)]       # dw2, dw1 (+db1), dwd, dwqkv (+dbqkv) at H = 1024


@pytest.mark.parametrize("T,shapes", [
    (8192, BLOOM),                                                    # the measured step: 384 tiles = one whole round + 128 tiles cut in two
    (64, BLOOM), (96, BLOOM), (160, BLOOM), (4096 + 32, BLOOM),       # 2 / 3 / 5 K-steps (single steps, odd halves), an odd number of pairs
    (512, [(256, 256, True)]),                                        # one problem, two tiles: both cut
    (256, [(128, 256, True), (256, 512, False), (384, 256, True)]),   # three problems, 1 + 4 + 3 tiles
    (1024, [(2048, 4096, True), (4096, 2048, False)]),                # 256 + 256 tiles: two whole rounds, nothing cut, column sums in a whole tile
    (2048, [(4096, 2048, True), (2048, 4096, True)]),                 # 512 tiles
], ids=lambda v: str(v) if isinstance(v, int) else f"{len(v)}p")
def test_grouped_weight_gradients_vs_fp64_and_per_product(T, shapes):
    o = ops()
    probs = _problems(T, shapes, seed=T * 7 + len(shapes))
    outs = o.wgrad_grouped(probs)
    torch.cuda.synchronize()
    for i, ((dy, x, want), (dw, db)) in enumerate(zip(probs, outs)):
        assert dw.shape == (dy.shape[1], x.shape[1]) and bool(torch.isfinite(dw).all())
        _check_fp64(f"problem {i} T={T}", dy, x, dw, db)
        single = o.linear_wgrad(dy, x)                                # the per-product kernel: other summation order, same operands
        rel = float((dw.double() - single.double()).norm() / (single.double().norm() + 1e-30))
        assert rel < 2e-6, (i, rel)
    # run-to-run: same bits
    again = o.wgrad_grouped(probs)
    torch.cuda.synchronize()
    for (a, ab), (b, bb) in zip(outs, again):
        assert torch.equal(a, b)
        assert (ab is None and bb is None) or torch.equal(ab, bb)


def test_grouped_weight_gradients_on_half_operands():
    """the fp16 twin of the kernel (round 5: v_mfma_f32_16x16x32_f16, the column sums against a fragment of HALF ones): the measured step's problem
    set and a small one with odd K-steps, against fp64 of the same half-rounded operands and against the per-product fp16 kernels"""
    o = ops()
    for T, shapes in ((8192, BLOOM), (96, [(128, 256, True), (256, 512, False), (384, 256, True)])):
        probs = _problems(T, shapes, seed=T + 3, dtype=torch.float16)
        outs = o.wgrad_grouped(probs)
        torch.cuda.synchronize()
        for i, ((dy, x, want), (dw, db)) in enumerate(zip(probs, outs)):
            _check_fp64(f"fp16 problem {i} T={T}", dy, x, dw, db)
            single = o.linear_wgrad(dy, x)
            assert float((dw.double() - single.double()).norm() / (single.double().norm() + 1e-30)) < 2e-6, i


GPT2 = [(1024, 4096, False), (4096, 1024, True), (1024, 1024, False), (3072, 1024, True)]        # the same four products through Conv1D ([in,out]) weights


@pytest.mark.parametrize("T,shapes", [
    (256, [(256, 512, False), (768, 256, False)]),                    # no bias gradients (the round-5 case)
    (256, [(256, 512, True), (768, 256, True)]),                      # bias gradients = column sums of the launch's B operand (round 6)
    (8192, GPT2),                                                     # GPT-2-medium's block: one whole round + 128 tiles cut in two (the sums of the cut tiles take the K-halves path)
    (96, GPT2), (160, GPT2),                                          # 3 / 5 K-steps: single steps and odd halves
    (512, [(256, 128, True)]),                                        # one tile [128 in, 256 out], cut in two
    (1024, [(2048, 4096, True), (4096, 2048, True)]),                 # whole rounds only, several tile rows (only the first sums)
], ids=lambda v: str(v) if isinstance(v, int) else f"{len(v)}p{sum(1 for t in v if t[2])}b")
def test_grouped_weight_gradients_in_out_layout(T, shapes):
    """[in,out] gradients (Conv1D weights, modeling_gpt.py:32-46): dy is the B operand of the launch; round 6: its column sums (the bias gradient)
    come from the tiles of the first tile row — MFMAs of the B fragments against a fragment of ones (its own kernel instantiation: the Bloom
    launch keeps its code).  Until then such steps ran a separate column-sum pass per bias."""
    o = ops()
    probs = _problems(T, shapes, seed=5)
    outs = o.wgrad_grouped(probs, in_out=True)
    torch.cuda.synchronize()
    for (dy, x, want), (dw, db) in zip(probs, outs):
        assert dw.shape == (x.shape[1], dy.shape[1]) and (db is not None) == want
        _check_fp64("in_out", dy, x, dw, db, in_out=True)
    # bias gradients of the two launch forms agree to the last bit of the MFMA's fp32 accumulation order per K-half: compare with [out,in] sums loosely
    if any(x.shape[1] % 256 or dy.shape[1] % 128 for dy, x, _ in probs):
        return                                                        # (the [out,in] launch tiles the transposed gradient: not every shape fits both)
    outs2 = o.wgrad_grouped(probs, in_out=False)
    torch.cuda.synchronize()
    for (dw, db), (dw2, db2) in zip(outs, outs2):
        assert relerr(dw.t(), dw2) < 1e-5
        if db is not None:
            assert relerr(db, db2) < 1e-5


def test_grouped_weight_gradients_refuse_what_the_tiling_cannot_take():
    o = ops()
    for T, shapes in ((32, [(256, 256, False)]), (100, [(256, 256, False)]), (256, [(192, 256, False)]), (256, [(256, 128, False)]),
                      (256, [(256, 256, False)] * 5), (64, [(8192, 4096, False)])):            # (the last: >= 32 Mi elements — left to the 256x256 tiles)
        with pytest.raises(lib().CtmiError):
            o.wgrad_grouped(_problems(T, shapes, seed=1))
    with pytest.raises(lib().CtmiError):                             # fp32 operands: parity mode keeps the per-product kernels
        g = torch.Generator(device=DEV).manual_seed(2)
        o.wgrad_grouped([(torch.randn(256, 256, generator=g, device=DEV), torch.randn(256, 256, generator=g, device=DEV), False)])


def test_grouped_weight_gradients_with_cold_operands_equal_warm_bit_for_bit():
    """the LDS-DMA ring's race detector (see test_gemm_with_cold_operands_equals_warm_bit_for_bit) for the grouped launch: whole tiles, halves,
    the work-item switches between problems with different leading dimensions — operands from HBM, six times"""
    o = ops()
    probs = _problems(8192, BLOOM, seed=11)
    o.wgrad_grouped(probs)
    warm = o.wgrad_grouped(probs)
    torch.cuda.synchronize()
    evict = torch.empty(3 << 28, dtype=torch.int32, device=DEV)               # 3 GiB of stores: > 256 MiB Infinity Cache + 32 MiB of L2s
    for i in range(6):
        evict.fill_(i)
        cold = o.wgrad_grouped(probs)
        torch.cuda.synchronize()
        for k, ((a, ab), (b, bb)) in enumerate(zip(cold, warm)):
            assert torch.equal(a, b), f"run {i}, problem {k}: {int((a != b).sum())} elements differ with cold operands"
            assert (ab is None) or torch.equal(ab, bb), f"run {i}, problem {k}: bias gradient differs with cold operands"


@pytest.mark.parametrize("mode", ["2", "3"])
def test_grouped_weight_gradients_other_split_modes(mode):
    """CTMI_WGRAD_GROUP=2 (every tile cut in two) and 3 (nothing cut) in a fresh process: the environment is read once"""
    e = dict(os.environ)
    e["CTMI_WGRAD_GROUP"] = mode
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "vs_fp64_and_per_product and (8192 or 96 or 3p)"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


def test_block_backward_uses_the_grouped_launch_and_matches_the_per_product_form():
    """ctmi_bloom_block_bwd at a Bloom-560M layer (T = 2048 here): the default (grouped) backward against the same call in a process with
    CTMI_WGRAD_GROUP=0 — every parameter gradient to fp32 summation order, dx bit for bit (the data-gradient chain does not change)"""
    code = r'''
import sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
from cleantransformer_amd import ops as o, _lib
from tests.test_gpu_block import _block_inputs
B, S, H, nh = 2, 1024, 1024, 16
x, dout, params, mask, slopes, _, _ = _block_inputs(B, S, H, nh, torch.bfloat16, seed=77, pad="right")
acts = o.bloom_block_fwd(x, params, mask, slopes, 1e-5, False, B, S, nh)
dx, grads = o.bloom_block_bwd(acts, x, params, mask, slopes, 1e-5, False, dout, use_side_stream=True)
torch.cuda.synchronize()
torch.save([dx.cpu()] + [g.cpu() for g in grads], sys.argv[1])
''' % ((os.path.dirname(os.path.dirname(os.path.abspath(__file__))),) * 2)
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("1", "0"):
            e = dict(os.environ)
            e["CTMI_WGRAD_GROUP"] = mode
            out = os.path.join(td, f"g{mode}.pt")
            r = subprocess.run([sys.executable, "-c", code, out], env=e, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            res[mode] = torch.load(out)
    assert torch.equal(res["1"][0], res["0"][0]), "dx"
    names = lib().BLK_PARAMS
    for n, a, b in zip(names, res["1"][1:], res["0"][1:]):
        rel = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        assert rel < 5e-6, (n, rel)
