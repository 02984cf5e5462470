"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/ctmi355.h declares
(no compute calls — there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ctmi355.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|uint32_t|const char\*)\s+(ctmi_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(2)] = n
    return decls


def test_header_declares_the_path():
    d = _declared()
    for must in ("ctmi_layernorm_fwd", "ctmi_layernorm_bwd", "ctmi_gemm", "ctmi_attn_fwd", "ctmi_attn_bwd", "ctmi_ce_fwd",
                 "ctmi_ce_bwd", "ctmi_embed_fwd", "ctmi_embed_bwd", "ctmi_adamw_step", "ctmi_sgd_step", "ctmi_mask_prep"):
        assert must in d, must


def test_library_builds_and_exports_every_declared_symbol():
    from cleantransformer_amd import _build, _lib
    path = _build.build(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    decl = _declared()
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in ctmi355.h but not exported"
    # the ctypes binding mirrors the header: same names, same arity
    assert set(_lib.PROTOTYPES) == set(decl), set(_lib.PROTOTYPES) ^ set(decl)
    for name, (_, args) in _lib.PROTOTYPES.items():
        assert len(args) == decl[name], (name, len(args), decl[name])
    assert _lib.load().ctmi_abi_version() == _lib.ABI_VERSION


def test_product_path_has_no_cpu_fallback():
    """ops on CPU tensors must raise, not silently compute."""
    import torch
    from cleantransformer_amd import _lib
    from cleantransformer_amd.transformer import LayerNorm
    ln = LayerNorm(8)
    with pytest.raises(_lib.CtmiError):
        ln(torch.randn(2, 8))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cleantransformer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
