"""ctypes binding of libctmi355.so — the C-ABI boundary declared in include/ctmi355.h.

The product path has no CPU fallback: if the HIP library is missing or an entry point fails,
an exception is raised (``CtmiError``)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must come first: libctmi355.so has to bind to the HIP runtime torch already loaded — two
#                              HIP runtimes in one process cannot both initialise the device)

from ._build import LIB_PATH as _DEFAULT_LIB_PATH

# CTMI_LIB_PATH: load another build of the same sources (kernel A/B experiments under tools/; see _build.build_variant)
LIB_PATH = os.environ.get("CTMI_LIB_PATH") or _DEFAULT_LIB_PATH

F32, BF16, F16 = 0, 1, 2
OPT_LEGACY_GRID = 4                   # OR-ed into mutate_grad of ctmi_adamw_step: the (stride loop, tensor) grid of ABI <= 14
OPT_SHADOW_F16 = 2                    # OR-ed into the mutate_grad / first_step argument of the fused optimizers: the shadows are IEEE half
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_RELU, EPI_DRELU, EPI_GELUG, EPI_MUL = 0, 1, 2, 3, 4, 5, 6
MT_MAX = 24
PROF_CLASSES = ("gemm_fwd", "gemm_dgrad", "gemm_wgrad", "lm_head", "attn_fwd", "attn_bwd", "layernorm", "loss", "optimizer", "reduce", "other")
ABI_VERSION = 15

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class CtmiError(RuntimeError):
    pass


class AttnDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("B", "nh", "Sq", "Sk", "hd",
                                   "q_bs", "q_hs", "q_rs", "k_bs", "k_hs", "k_rs",
                                   "v_bs", "v_hs", "v_rs", "o_bs", "o_hs", "o_rs",
                                   "am_b", "am_h", "am_q", "am_k")] + [("scale", f32), ("causal", i32), ("future_fill", f32), ("dropout_seed", C.c_uint32), ("dropout_p", f32), ("reserved_", i32)]


class ReduceJob(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("n", i64), ("part_stride", i64), ("nparts", i32), ("accumulate", i32), ("alpha", f32), ("pad_", i32)]


class WgradProblem(C.Structure):
    """ctmi_wgrad_problem (include/ctmi355.h)."""
    _fields_ = [("dy", vp), ("x", vp), ("dw", vp), ("db", vp), ("n_out", i64), ("n_in", i64), ("in_out", i32), ("pad_", i32)]


BLK_SLOTS = ("ln1", "mean1", "rstd1", "qkv", "att", "stat_m", "stat_l", "h1", "mean2", "rstd2", "ln2", "u", "g", "out")
BLK_QKV_BLOCKED, BLK_WGRAD_IN_OUT, BLK_W_IN_OUT = 1, 2, 4
BLK_PARAMS = ("ln1_w", "ln1_b", "wqkv", "bqkv", "wd", "bd", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")


class BloomBlock(C.Structure):
    """ctmi_bloom_block (include/ctmi355.h)."""
    _fields_ = [("B", i64), ("S", i64), ("H", i64), ("nh", i64), ("eps", f32), ("post_ln_res", i32), ("dtype", i32), ("flags", i32),
                ("attn_scale", f32), ("future_fill", f32)] + \
               [(n, vp) for n in BLK_PARAMS] + [(n, vp) for n in ("slopes", "kpos", "kvalid", "first_valid", "x", "slab")]


class BloomBlockGrads(C.Structure):
    """ctmi_bloom_block_grads (include/ctmi355.h)."""
    _fields_ = [("dout", vp), ("dx", vp)] + [("d" + n, vp) for n in BLK_PARAMS] + \
               [("ws", vp), ("ws_bytes", i64), ("splitk_ws", vp), ("splitk_ws_bytes", i64), ("side_stream", vp),
                ("side_splitk_ws", vp), ("side_splitk_ws_bytes", i64), ("defer_join", i32)]


# name -> (restype, argtypes); must mirror include/ctmi355.h exactly (checked by tests/test_abi.py)
PROTOTYPES = {
    "ctmi_abi_version": (i32, []),
    "ctmi_last_error": (C.c_char_p, []),
    "ctmi_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, f32, i32, vp]),
    "ctmi_layernorm_bwd_ws": (i64, [i64, i64]),
    "ctmi_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i64, i64, i32, vp]),
    "ctmi_gemm": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, i64, i64, i64, f32, i32, vp, vp, i32, vp, vp, i32, i32, vp, i64, vp]),
    "ctmi_set_launch_policy": (i32, [i32, i32]),
    "ctmi_get_launch_policy": (i32, [C.POINTER(i32), C.POINTER(i32)]),
    "ctmi_colsum": (i32, [vp, i64, vp, i32, vp, i64, i64, i32, vp]),
    "ctmi_colsum_ws": (i64, [i64, i64]),
    "ctmi_attn_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(AttnDesc), i32, vp]),
    "ctmi_attn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(AttnDesc), i32, vp]),
    "ctmi_attn_set_path": (i32, [i32]),
    "ctmi_mask_prep": (i32, [vp, vp, vp, vp, i64, i64, vp]),
    "ctmi_embed_fwd": (i32, [vp, vp, vp, i64, i64, i64, i32, vp, vp]),
    "ctmi_embed_bwd": (i32, [vp, vp, vp, i64, i64, i64, i32, f32, vp]),
    "ctmi_ce_fwd": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, i64, i32, vp]),
    "ctmi_ce_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, vp]),
    "ctmi_ce_fwd_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, i64, f32, vp, i32, vp]),
    "ctmi_scale_if": (i32, [vp, i64, i64, i64, vp, f32, vp, i32, vp]),
    "ctmi_scale_if_passes": (i64, []),
    "ctmi_reduce_jobs": (i32, [C.POINTER(ReduceJob), i32, vp]),
    "ctmi_wgrad_grouped": (i32, [C.POINTER(WgradProblem), i32, i64, i32, vp, i64, vp]),
    "ctmi_bloom_block_layout": (i64, [i64, i64, i64, i64, i32, C.POINTER(i64)]),
    "ctmi_bloom_block_fwd": (i32, [C.POINTER(BloomBlock), vp]),
    "ctmi_bloom_block_bwd_ws": (i64, [i64, i64, i64, i64, i32]),
    "ctmi_bloom_block_wgrad_grouped": (i32, [i64, i64, i64, i32, i32]),
    "ctmi_bloom_block_bwd": (i32, [C.POINTER(BloomBlock), C.POINTER(BloomBlockGrads), vp]),
    "ctmi_ce_soft_fwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, i64, i32, i64, i32, vp]),
    "ctmi_ce_soft_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
    "ctmi_ddp_unique_id": (C.c_int, [C.c_void_p]),
    "ctmi_ddp_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ctmi_ddp_destroy": (C.c_int, [C.c_void_p]),
    "ctmi_ddp_all_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ctmi_ddp_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ctmi_ddp_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "ctmi_ddp_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ctmi_dropout_hash": (C.c_uint32, [C.c_uint32]),
    "ctmi_dropout_keep_hash": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "ctmi_dropout_threshold": (C.c_uint32, [f32]),
    "ctmi_dropout": (i32, [vp, vp, vp, i64, f32, C.c_uint32, i32, vp]),
    "ctmi_adamw_step": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64),
                              i32, f32, f32, f32, f32, f32, i32, i32, i32, f32, vp]),
    "ctmi_adamw_set_hyper": (i32, [vp, f32, f32, f32, f32, f32, i32, i32, i32, f32, vp]),
    "ctmi_adamw_step_dev": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "ctmi_sgd_step": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64),
                            i32, f32, f32, f32, f32, i32, vp]),
    "ctmi_cast": (i32, [vp, i32, vp, i32, i64, vp]),
    "ctmi_transpose_cast": (i32, [vp, vp, i32, i64, i64, vp]),
    "ctmi_sumsq": (i32, [vp, i64, vp, i32, vp]),
    "ctmi_scale": (i32, [vp, i64, f32, vp, vp]),
    "ctmi_scale_copy": (i32, [vp, vp, i64, f32, vp]),
    "ctmi_amp_unscale": (i32, [C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "ctmi_amp_update": (i32, [vp, f32, f32, i32, vp]),
    "ctmi_argmax": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "ctmi_row_lse": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "ctmi_group_topk": (i32, [vp, i64, vp, vp, f32, vp, vp, i64, i32, i64, i32, i32, vp]),
    "ctmi_scores_filter": (i32, [vp, i64, f32, vp, i64, f32, vp, i64, i64, i64, vp]),
    "ctmi_probe": (i32, [i32, vp, vp, vp]),
    "ctmi_clock_probe": (i32, [i32, vp, vp]),
    "ctmi_probe_dyn_lds": (i32, [i64, vp, vp]),
    "ctmi_profile_begin": (i32, []),
    "ctmi_profile_end": (i32, [C.POINTER(f32), C.POINTER(i32)]),
}

_lib = None


def lib_path() -> str:
    return LIB_PATH


def load():
    """Load (once) and type the shared library.  Raises CtmiError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CtmiError(
            f"{LIB_PATH} is missing: the MI355X kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc). There is no CPU fallback for the product path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:                                       # pragma: no cover
        raise CtmiError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CtmiError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ctmi_abi_version() != ABI_VERSION:
        raise CtmiError(f"ABI mismatch: library {lib.ctmi_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ctmi_last_error()
        raise CtmiError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
