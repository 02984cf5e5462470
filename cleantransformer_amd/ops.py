"""Tensor-level wrappers over the C ABI (include/ctmi355.h).

PyTorch is used here only as plumbing: it owns device memory (caching allocator) and the
current HIP stream.  Every arithmetic operation on the hot path is a ctmi_* kernel; a missing
library or a failing launch raises (there is no eager/CPU fallback).
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
import math
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import AttnDesc, BF16, F16, F32, check

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise _lib.CtmiError(f"unsupported compute dtype {dtype}: the ctmi355 kernels run fp32, bf16 or fp16")


# amp.autocast(dtype=...) override of a model's config.compute_dtype for the forwards run inside the context (and their backwards: the saved
# activations carry the dtype).  Process-wide like torch's autocast state is thread-wide; the training loop is single-threaded on the forward side.
_AUTOCAST = threading.local()


def effective_compute_dtype(model_dtype: torch.dtype) -> torch.dtype:
    """The dtype a model forward computes in: amp.autocast(dtype=torch.float16 / torch.bfloat16)'s while such a context is active (torch.autocast
    semantics: the context, not the module, picks the matmul dtype — ft_bloom_DDP.py:122), else the model's own config.compute_dtype."""
    ac = get_autocast_dtype()
    return ac if ac is not None else model_dtype


def get_autocast_dtype():
    return getattr(_AUTOCAST, "dtype", None)


def set_autocast_dtype(dtype) -> None:
    """Per THREAD, like torch's autocast state: a forward on another thread (a data-loader worker that runs a model, a second trainer) does not pick
    up this thread's context (round-5 advisor)."""
    _AUTOCAST.dtype = dtype


def __getattr__(name):                                    # `ops._AUTOCAST_DTYPE` (read-only view of this thread's override; rounds 3-5 had a module global)
    if name == "_AUTOCAST_DTYPE":
        return get_autocast_dtype()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CtmiError("ctmi355 ops need tensors on an MI355X device (cuda:N); there is no CPU path in the product "
                                 "package — the CPU oracle lives under oracle/ and is test infrastructure only")


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x2d: Tensor, w: Tensor, b: Tensor, eps: float):
    """x2d [rows, cols] (fp32|bf16, contiguous); w,b fp32 [cols] -> y, mean, rstd."""
    _need_cuda(x2d, w, b)
    rows, cols = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    check(_lib.load().ctmi_layernorm_fwd(_p(x2d), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, cols, float(eps),
                                         dt_code(x2d.dtype), _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dres: Optional[Tensor] = None):
    """-> dx (same dtype as x; + dres fused), dw fp32, db fp32."""
    rows, cols = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    dw = torch.empty(cols, dtype=torch.float32, device=x.device)
    db = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.ctmi_layernorm_bwd_ws(rows, cols), dtype=torch.float32, device=x.device)
    check(lib.ctmi_layernorm_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dw), _p(db), 0, _p(ws),
                                 rows, cols, dt_code(x.dtype), _stream()), "layernorm_bwd")
    return dx, dw, db


# ------------------------------------------------------------------------------------------------ GEMM
class KernelTimer:
    """Opt-in HIP-event bracket around tagged launches (bench.py's live per-kernel roofline measurement).  Events are
    recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self, tags):
        self.tags = set(tags)
        self.events = {t: [] for t in self.tags}

    def ms(self, tag):
        ev = self.events[tag]
        return [a.elapsed_time(b) for a, b in ev]


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _timer
    _timer = t


_SPLITK_WS = {}


def _splitk_ws(device) -> Tensor:
    """Scratch for split-K GEMMs (fp32 slabs; 8 x [8192,1024] fits), one per (device, stream): kernels on one stream
    run in order, but the weight-gradient side stream must not share slabs with the main stream."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty(16 * 4096 * 1024, dtype=torch.float32, device=device)
        _SPLITK_WS[key] = ws
    return ws


_SIDE_STREAMS = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The per-device side HIP stream on which weight-gradient GEMMs and bias-gradient reductions run concurrently with
    the dgrad / attention-backward chain (they feed nothing downstream in backward)."""
    st = _SIDE_STREAMS.get(device)
    if st is None:
        prio = os.environ.get("CTMI_SIDE_STREAM_PRIORITY")               # "low" / "high" / an integer: hipStreamCreateWithPriority (experiments)
        if prio is None:
            st = torch.cuda.Stream(device=device)
        else:
            hip = C.CDLL("libamdhip64.so")
            lo, hi = C.c_int(0), C.c_int(0)
            hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))   # least, greatest (numerically: greatest priority is the smaller number)
            val = lo.value if prio == "low" else (hi.value if prio == "high" else int(prio))
            h = C.c_void_p()
            with torch.cuda.device(device):
                if hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(val)) != 0:
                    raise _lib.CtmiError("hipStreamCreateWithPriority failed")
            st = torch.cuda.ExternalStream(h.value, device=device)
            st._ctmi_priority = (val, lo.value, hi.value)
        _SIDE_STREAMS[device] = st
    return st


def gemm(A: Tensor, lda: int, a_kmajor: bool, B: Tensor, ldb: int, b_kmajor: bool, M: int, N: int, K: int, *,
         out: Optional[Tensor] = None, out_f32: bool = False, bias: Optional[Tensor] = None,
         residual: Optional[Tensor] = None, epilogue: int = _lib.EPI_NONE, aux_in: Optional[Tensor] = None,
         aux_out: Optional[Tensor] = None, alpha: float = 1.0, beta: int = 0, tag: Optional[str] = None) -> Tensor:
    dtype = A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else dtype, device=A.device)
    timed = _timer is not None and tag in _timer.tags
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = _splitk_ws(A.device) if (epilogue == _lib.EPI_NONE and bias is None and residual is None) else None
    ldc = out.stride(0) if out.dim() == 2 else N                       # a column-padded output buffer keeps its own row pitch
    check(_lib.load().ctmi_gemm(_p(A), lda, int(a_kmajor), _p(B), ldb, int(b_kmajor), _p(out), ldc, M, N, K, float(alpha), int(beta),
                                _p(bias), _p(residual), int(epilogue), _p(aux_in), _p(aux_out), int(out_f32), dt_code(dtype),
                                _p(ws), 0 if ws is None else ws.numel() * 4, _stream()), "gemm")
    if timed:
        e1.record()
        _timer.events[tag].append((e0, e1))
    return out


def linear_fwd(x2d: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
               epilogue: int = _lib.EPI_NONE, aux_out: Optional[Tensor] = None, tag: Optional[str] = None,
               out: Optional[Tensor] = None) -> Tensor:
    """y[T,out] = epilogue(x[T,in] @ w[out,in]^T + bias) (+ residual).  w in the compute dtype."""
    T, K = x2d.shape
    N = w.shape[0]
    return gemm(x2d, K, False, w, K, False, T, N, K, bias=bias, residual=residual, epilogue=epilogue, aux_out=aux_out, tag=tag, out=out)


def linear_dgrad(dy: Tensor, w: Tensor, epilogue: int = _lib.EPI_NONE, aux_in: Optional[Tensor] = None,
                 residual: Optional[Tensor] = None) -> Tensor:
    """dx[T,in] = dy[T,out] @ w[out,in]  (w read K-major: no transposed copy)."""
    T, Nout = dy.shape
    Kin = w.shape[1]
    return gemm(dy, Nout, False, w, Kin, True, T, Kin, Nout, epilogue=epilogue, aux_in=aux_in, residual=residual)


def linear_wgrad(dy: Tensor, x2d: Tensor, out: Optional[Tensor] = None, accumulate: bool = False, alpha: float = 1.0) -> Tensor:
    """dW[out,in] (fp32) = alpha * dy[T,out]^T @ x[T,in]."""
    T, Nout = dy.shape
    Kin = x2d.shape[1]
    return gemm(dy, Nout, True, x2d, Kin, True, Nout, Kin, T, out=out, out_f32=True, beta=int(accumulate), alpha=alpha)


def wgrad_grouped(problems, in_out: bool = False):
    """The weight (and bias) gradients of several Linears in ONE launch (include/ctmi355.h ctmi_wgrad_grouped; autograd of modeling_bloom.py:79,121,
    256,267): problems = [(dy [T,n_out], x [T,n_in], want_bias_grad)], bf16 -> [(dw fp32 [n_out,n_in] (or [n_in,n_out] if in_out), db fp32 [n_out] or None)].
    Raises CtmiError (unsupported) for shapes outside the kernel's tiling; callers fall back to linear_wgrad / colsum per product."""
    lib = _lib.load()
    arr = (_lib.WgradProblem * len(problems))()
    outs = []
    T = problems[0][0].shape[0]
    for i, (dy, x, want_db) in enumerate(problems):
        _need_cuda(dy, x)
        assert dy.is_contiguous() and x.is_contiguous() and dy.shape[0] == T and x.shape[0] == T
        n_out, n_in = dy.shape[1], x.shape[1]
        dw = torch.empty((n_in, n_out) if in_out else (n_out, n_in), dtype=torch.float32, device=dy.device)
        db = torch.empty(n_out, dtype=torch.float32, device=dy.device) if want_db else None
        arr[i].dy, arr[i].x, arr[i].dw, arr[i].db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (None if db is None else db.data_ptr())
        arr[i].n_out, arr[i].n_in, arr[i].in_out = n_out, n_in, int(in_out)
        outs.append((dw, db))
    ws = _splitk_ws(problems[0][0].device)
    check(lib.ctmi_wgrad_grouped(arr, len(problems), T, dt_code(problems[0][0].dtype), ws.data_ptr(), ws.numel() * 4, _stream()), "wgrad_grouped")
    return outs


def colsum(x2d: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    M, N = x2d.shape
    lib = _lib.load()
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(lib.ctmi_colsum_ws(M, N), dtype=torch.float32, device=x2d.device)
    check(lib.ctmi_colsum(_p(x2d), N, _p(out), int(accumulate), _p(ws), M, N, dt_code(x2d.dtype), _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------ attention
class MaskInfo:
    """Device-side digest of attention_mask [B,S]: ALiBi key positions, key validity, first valid key."""
    __slots__ = ("kpos", "kvalid", "first_valid", "B", "S", "seen_params", "shared_params")

    def __init__(self, attention_mask: Tensor):
        _need_cuda(attention_mask)
        self.seen_params, self.shared_params = set(), False             # per-forward record of the block parameters (see note_block_params)
        am = _c(attention_mask.to(torch.int64))
        B, S = am.shape
        self.B, self.S = B, S
        self.kpos = torch.empty((B, S), dtype=torch.float32, device=am.device)
        self.kvalid = torch.empty((B, S), dtype=torch.int32, device=am.device)
        self.first_valid = torch.empty((B,), dtype=torch.int32, device=am.device)
        check(_lib.load().ctmi_mask_prep(_p(am), _p(self.kpos), _p(self.kvalid), _p(self.first_valid), B, S, _stream()),
              "mask_prep")


def _strided_desc(B, nh, Sq, Sk, hd, q_str, k_str, v_str, o_str, scale, causal, am_str=(0, 0, 0, 0), future_fill: float = 0.0,
                  dropout_p: float = 0.0, dropout_seed: int = 0) -> AttnDesc:
    d = AttnDesc()
    d.B, d.nh, d.Sq, d.Sk, d.hd = B, nh, Sq, Sk, hd
    d.q_bs, d.q_hs, d.q_rs = q_str
    d.k_bs, d.k_hs, d.k_rs = k_str
    d.v_bs, d.v_hs, d.v_rs = v_str
    d.o_bs, d.o_hs, d.o_rs = o_str
    d.am_b, d.am_h, d.am_q, d.am_k = am_str
    d.scale = float(scale)
    d.causal = int(causal)
    d.future_fill = float(future_fill)                                # 0 = finfo.min (Bloom); GPT-2 passes -1e4 (modeling_gpt.py:88-89)
    d.dropout_p, d.dropout_seed = float(dropout_p), int(dropout_seed) & 0xFFFFFFFF     # attention-probability dropout (0 = off)
    return d


def _view_ptr(t: Tensor):
    return C.c_void_p(t.data_ptr())


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, out: Tensor, desc: AttnDesc, slopes: Optional[Tensor],
             mask: Optional[MaskInfo], add_mask: Optional[Tensor] = None):
    """q,k,v,out: tensors (possibly views) whose data_ptr is element (b=0,h=0,row=0,d=0); strides in `desc`."""
    dev = q.device
    stat_m = torch.empty((desc.B, desc.nh, desc.Sq), dtype=torch.float32, device=dev)
    stat_l = torch.empty((desc.B, desc.nh, desc.Sq), dtype=torch.float32, device=dev)
    check(_lib.load().ctmi_attn_fwd(_view_ptr(q), _view_ptr(k), _view_ptr(v), _view_ptr(out), _p(stat_m), _p(stat_l),
                                    _p(slopes), _p(mask.kpos) if (mask is not None and slopes is not None) else None,
                                    _p(mask.kvalid) if mask is not None else None,
                                    _p(mask.first_valid) if mask is not None else None,
                                    _p(add_mask), C.byref(desc), dt_code(q.dtype), _stream()), "attn_fwd")
    return stat_m, stat_l


def attn_bwd(q, k, v, o, d_o, stat_m, stat_l, dq, dk, dv, desc: AttnDesc, slopes, mask: Optional[MaskInfo],
             add_mask: Optional[Tensor] = None):
    delta = torch.empty((desc.B, desc.nh, desc.Sq), dtype=torch.float32, device=q.device)
    check(_lib.load().ctmi_attn_bwd(_view_ptr(q), _view_ptr(k), _view_ptr(v), _view_ptr(o), _view_ptr(d_o), _p(stat_m), _p(stat_l),
                                    _view_ptr(dq), _view_ptr(dk), _view_ptr(dv), _p(delta), _p(slopes),
                                    _p(mask.kpos) if (mask is not None and slopes is not None) else None,
                                    _p(mask.kvalid) if mask is not None else None,
                                    _p(mask.first_valid) if mask is not None else None,
                                    _p(add_mask), C.byref(desc), dt_code(q.dtype), _stream()), "attn_bwd")


def fused_qkv_desc(B: int, S: int, nh: int, hd: int, causal: bool, dropout_p: float = 0.0, dropout_seed: int = 0) -> AttnDesc:
    """Strides of the head-interleaved fused QKV activation [B,S,nh,3,hd] (modeling_bloom.py:81-82) and of the
    merged-head context [B,S,nh*hd]."""
    H = nh * hd
    qkv = (S * 3 * H, 3 * hd, 3 * H)
    return _strided_desc(B, nh, S, S, hd, qkv, qkv, qkv, (S * H, hd, H), 1.0 / math.sqrt(hd), causal, dropout_p=dropout_p,
                         dropout_seed=dropout_seed)


# ------------------------------------------------------------------------------------------------ embedding / CE
def embed_fwd(table: Tensor, ids: Tensor, err_flag: Optional[Tensor] = None) -> Tensor:
    _need_cuda(table, ids)
    V, H = table.shape
    ids_c = _c(ids)
    out = torch.empty((*ids.shape, H), dtype=table.dtype, device=table.device)
    check(_lib.load().ctmi_embed_fwd(_p(table), _p(ids_c), _p(out), ids_c.numel(), H, V, dt_code(table.dtype), _p(err_flag),
                                     _stream()), "embed_fwd")
    return out


def embed_bwd(dout: Tensor, ids: Tensor, dtable: Tensor, scale: float = 1.0) -> None:
    """dtable (fp32 [V,H]) += scale * scatter(dout rows by ids)."""
    V, H = dtable.shape
    ids_c = _c(ids)
    check(_lib.load().ctmi_embed_bwd(_p(dout), _p(ids_c), _p(dtable), ids_c.numel(), H, V, dt_code(dout.dtype), float(scale),
                                     _stream()), "embed_bwd")


def ce_fwd(logits2d: Tensor, labels: Tensor, seq: int, shift: int, ignore_index: int = -100, denom_mode: int = 0,
           denom_rows: int = 0):
    """-> loss_out fp32[2] (= [loss, 1/denom]), row_lse fp32[N]."""
    N, Cn = logits2d.shape
    dev = logits2d.device
    row_lse = torch.empty(N, dtype=torch.float32, device=dev)
    row_loss = torch.empty(N, dtype=torch.float32, device=dev)
    loss_out = torch.empty(2, dtype=torch.float32, device=dev)
    check(_lib.load().ctmi_ce_fwd(_p(logits2d), logits2d.stride(0), _p(labels), _p(row_lse), _p(row_loss), _p(loss_out), N, Cn,
                                  seq, shift, ignore_index, denom_mode, denom_rows, dt_code(logits2d.dtype), _stream()), "ce_fwd")
    return loss_out, row_lse


def ce_bwd(logits2d: Tensor, labels: Tensor, row_lse: Tensor, loss_out: Tensor, gout: Optional[Tensor], seq: int, shift: int,
           ignore_index: int = -100, out: Optional[Tensor] = None) -> Tensor:
    N, Cn = logits2d.shape
    if out is None:
        out = torch.empty((N, Cn), dtype=logits2d.dtype, device=logits2d.device)
    check(_lib.load().ctmi_ce_bwd(_p(logits2d), logits2d.stride(0), _p(labels), _p(row_lse), _p(loss_out), _p(gout), _p(out),
                                  out.stride(0), N, Cn, seq, shift, ignore_index, dt_code(logits2d.dtype), _stream()), "ce_bwd")
    return out


def ce_soft_fwd(logits2d: Tensor, target: Tensor, denom_mode: int, denom_rows: int):
    """probability targets (loss.py:43-46) -> loss_out fp32[2], row_lse fp32[N], row_tsum fp32[N]."""
    _need_cuda(logits2d, target)
    N, Cn = logits2d.shape
    assert target.shape == (N, Cn) and target.dtype == torch.float32 and target.stride(1) == 1 and logits2d.stride(1) == 1
    dev = logits2d.device
    row_lse = torch.empty(N, dtype=torch.float32, device=dev)
    row_tsum = torch.empty(N, dtype=torch.float32, device=dev)
    row_loss = torch.empty(N, dtype=torch.float32, device=dev)
    loss_out = torch.empty(2, dtype=torch.float32, device=dev)
    check(_lib.load().ctmi_ce_soft_fwd(_p(logits2d), logits2d.stride(0), _p(target), target.stride(0), _p(row_lse), _p(row_tsum),
                                       _p(row_loss), _p(loss_out), N, Cn, denom_mode, denom_rows, dt_code(logits2d.dtype), _stream()),
          "ce_soft_fwd")
    return loss_out, row_lse, row_tsum


def ce_soft_bwd(logits2d: Tensor, target: Tensor, row_lse: Tensor, row_tsum: Tensor, loss_out: Tensor, gout: Optional[Tensor]) -> Tensor:
    N, Cn = logits2d.shape
    out = torch.empty((N, Cn), dtype=logits2d.dtype, device=logits2d.device)
    check(_lib.load().ctmi_ce_soft_bwd(_p(logits2d), logits2d.stride(0), _p(target), target.stride(0), _p(row_lse), _p(row_tsum),
                                       _p(loss_out), _p(gout), _p(out), out.stride(0), N, Cn, dt_code(logits2d.dtype), _stream()),
          "ce_soft_bwd")
    return out


# The upstream gradient the training loop is about to send into the loss node, when it is known BEFORE the forward: 1 / accumulation steps
# (trainer.py:468-504 backpropagates `loss / ga`), the loss scale of a GradScaler (ft_bloom_DDP.py:123-127: a device scalar), or both.
# The fused loss folds it into dlogits in its one pass (fp32, one rounding); its backward then rescales only if the gradient that actually
# arrives is a different number.  Process-wide (the training loop is single-threaded on the forward side).
_EXPECTED_LOSS_GRAD = {"factor": 1.0, "dev": None, "owner": None}


def set_expected_loss_grad(factor: Optional[float] = None, scale: Optional[Tensor] = None, owner=None) -> None:
    """Persistent form (amp.GradScaler registers its device scale here); `None` leaves a field as it is, `scale=False` clears it.
    `owner` (the scaler) is held by weak reference: the registration ends with the scaler — deleted or disabled — instead of folding a
    stale scale into every later loss of the process (round-4 advisor).  One registration at a time: a second scaler replaces the first;
    that costs the first its saved rescale pass, never correctness (the loss node compares what it folded with what arrives)."""
    if factor is not None:
        if not factor > 0.0:
            raise ValueError("expected loss gradient factor must be > 0")
        _EXPECTED_LOSS_GRAD["factor"] = float(factor)
    if scale is False:
        _EXPECTED_LOSS_GRAD["dev"] = _EXPECTED_LOSS_GRAD["owner"] = None
    elif scale is not None:
        if scale.numel() != 1 or scale.dtype != torch.float32:
            raise ValueError("expected loss gradient scale: one fp32 element on the device")
        import weakref
        _EXPECTED_LOSS_GRAD["dev"] = scale
        _EXPECTED_LOSS_GRAD["owner"] = weakref.ref(owner) if owner is not None else None


@contextlib.contextmanager
def expected_loss_grad(factor: float = 1.0):
    """`with ops.expected_loss_grad(1 / ga): loss = model(...); (loss / ga).backward()` — the Trainer's accumulation micro-steps."""
    old = _EXPECTED_LOSS_GRAD["factor"]
    set_expected_loss_grad(factor=factor)
    try:
        yield
    finally:
        _EXPECTED_LOSS_GRAD["factor"] = old


def current_expected_loss_grad():
    own = _EXPECTED_LOSS_GRAD["owner"]
    if _EXPECTED_LOSS_GRAD["dev"] is not None and own is not None:
        o = own()
        if o is None or not o.is_enabled():                             # the scaler that registered this scale is gone
            _EXPECTED_LOSS_GRAD["dev"] = _EXPECTED_LOSS_GRAD["owner"] = None
    return _EXPECTED_LOSS_GRAD["factor"], _EXPECTED_LOSS_GRAD["dev"]


def ce_fwd_bwd(logits2d: Tensor, labels: Tensor, seq: int, shift: int, ignore_index: int = -100, denom_mode: int = 0,
               denom_rows: int = 0, grad_factor: float = 1.0, grad_factor_dev: Optional[Tensor] = None):
    """Loss and dlogits (for the upstream gradient grad_factor * grad_factor_dev[0]; default 1) in one pass -> loss_out fp32[2],
    row_lse fp32[N], dlogits [N,C]."""
    N, Cn = logits2d.shape
    dev = logits2d.device
    row_lse = torch.empty(N, dtype=torch.float32, device=dev)
    row_loss = torch.empty(N, dtype=torch.float32, device=dev)
    loss_out = torch.empty(2, dtype=torch.float32, device=dev)
    if logits2d.stride(0) != Cn:
        # a padded row pitch (odd vocabulary — GPT-2's 50257 in rows of 50272): dlogits gets the same pitch with zeroed pad columns (dl._base is the
        # padded buffer).  The loss node's BACKWARD registers it in ZERO_PADDED — that table is per thread, and the backward pass runs on autograd's
        # worker thread, not on the one that ran this forward
        Vp = logits2d.stride(0)
        buf = torch.empty((N, Vp), dtype=logits2d.dtype, device=dev)
        buf[:, Cn:].zero_()
        dl = buf[:, :Cn]
    else:
        dl = torch.empty((N, Cn), dtype=logits2d.dtype, device=dev)
    check(_lib.load().ctmi_ce_fwd_bwd(_p(logits2d), logits2d.stride(0), _p(labels), _p(row_lse), _p(row_loss), _p(loss_out), _p(dl),
                                      dl.stride(0), N, Cn, seq, shift, ignore_index, denom_mode, denom_rows, float(grad_factor),
                                      _p(grad_factor_dev), dt_code(logits2d.dtype), _stream()), "ce_fwd_bwd")
    return loss_out, row_lse, dl


def ce_fused_ok(logits2d: Tensor) -> bool:
    """Rows 16-byte aligned — densely packed, or (round 6) at the padded pitch ops.pad_rows gives an odd vocabulary: the single-pass
    loss+gradient kernel applies (it takes the row pitch and finishes a row's last C % 8 classes one by one)."""
    vec = 16 // logits2d.element_size()
    C = logits2d.shape[1]
    dense = logits2d.stride(0) == C and C % vec == 0
    padded = logits2d.stride(0) != C and logits2d.stride(0) == pad_rows(C)
    return logits2d.is_cuda and logits2d.stride(1) == 1 and (dense or padded) and logits2d.stride(0) % vec == 0 and logits2d.data_ptr() % 16 == 0


def scale_if_(x2d: Tensor, s_dev: Tensor, applied: float = 1.0, applied_dev: Optional[Tensor] = None) -> Tensor:
    """x2d *= s_dev[0] / (applied * applied_dev[0]), skipped on the device when the two are exactly equal."""
    rows, cols = x2d.shape
    check(_lib.load().ctmi_scale_if(_p(x2d), x2d.stride(0), rows, cols, _p(s_dev), float(applied), _p(applied_dev), dt_code(x2d.dtype),
                                    _stream()), "scale_if")
    return x2d


def scale_if_passes() -> int:
    """How many scale_if_ calls of this process really rescaled their tensor (include/ctmi355.h ctmi_scale_if_passes; synchronises)."""
    return int(_lib.load().ctmi_scale_if_passes())


# ------------------------------------------------------------------------------------------------ one Bloom block per call
class _BlockLayout:
    __slots__ = ("off", "bytes", "bwd_ws_bytes")


_BLOCK_LAYOUTS = {}
_BLOCK_WS = {}


def block_layout(B: int, S: int, H: int, nh: int, dtype: torch.dtype) -> _BlockLayout:
    key = (B, S, H, nh, dtype)
    lay = _BLOCK_LAYOUTS.get(key)
    if lay is None:
        lib = _lib.load()
        offs = (C.c_int64 * len(_lib.BLK_SLOTS))()
        lay = _BlockLayout()
        lay.bytes = int(lib.ctmi_bloom_block_layout(B, S, H, nh, dt_code(dtype), offs))
        lay.off = dict(zip(_lib.BLK_SLOTS, [int(o) for o in offs]))
        lay.bwd_ws_bytes = int(lib.ctmi_bloom_block_bwd_ws(B, S, H, nh, dt_code(dtype)))
        _BLOCK_LAYOUTS[key] = lay
    return lay


def _block_ws(device, nbytes: int, slot: int = 0) -> Tensor:
    """Scratch of the block backward (intermediate gradients, partial rows): grown on demand, one buffer per device and slot.  With the
    side stream joined inside every call consecutive blocks share slot 0; with the join deferred to the end of the backward pass
    (bloom_block_bwd(defer_join=True)) they alternate between two slots, each guarded by the event of the call that used it last."""
    ws = _BLOCK_WS.get((device, slot))
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _BLOCK_WS[(device, slot)] = ws
    return ws


# deferred join of the weight-gradient side stream: per device [next slot, {slot: event of the last call that used it}]
_DEFER = {}
_LAST_DEFERRED_GRAD_PTRS = []
_DEFER_JOIN = os.environ.get("CTMI_WGRAD_DEFER_JOIN", "1") != "0"


_DEFER_HOLDS = 0


def hold_deferred_wgrad_join() -> None:
    """THE CONTRACT for anything that touches parameter gradients DURING a backward pass — gradient hooks of any kind (tensor hooks,
    post-accumulate hooks, hooks on the AccumulateGrad node as torch DDP / FSDP / apex register them), bucket copies, in-backward optimizers:
    call this once (and release_deferred_wgrad_join() when detached).  While any hold is active the block backward joins its weight-gradient
    side stream before it returns, so a gradient handed to autograd is complete on the compute stream.  cleantransformer_amd's
    DistributedDataParallel holds it for its lifetime; CTMI_WGRAD_DEFER_JOIN=0 in the environment is the process-wide form."""
    global _DEFER_HOLDS
    _DEFER_HOLDS += 1


def release_deferred_wgrad_join() -> None:
    global _DEFER_HOLDS
    _DEFER_HOLDS = max(0, _DEFER_HOLDS - 1)


def note_block_params(mask: Optional["MaskInfo"], params) -> None:
    """Called by a block node's forward with its parameters and the per-forward MaskInfo: a parameter that feeds TWO block nodes of one forward
    (weights shared across layers) gets its two gradients summed by the autograd engine on the compute stream — the deferred join must stay off
    for that forward."""
    if mask is None:
        return
    seen = getattr(mask, "seen_params", None)
    if seen is None:                                                    # (a per-forward context object other than ops.MaskInfo)
        seen = set()
        mask.seen_params, mask.shared_params = seen, False
    for q in params:
        k = id(q)
        if k in seen:
            mask.shared_params = True
        seen.add(k)


def _multi_rank_job() -> bool:
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:                                                   # noqa: BLE001
        return False


def params_allow_deferred_grads(params, mask: Optional["MaskInfo"] = None) -> bool:
    """The parameter gradients of a block may complete on the side stream AFTER the autograd node has returned only if nothing touches them
    before the end of the backward pass.  Refused when: a hold is active (hold_deferred_wgrad_join: the explicit contract of wrappers that
    install gradient hooks — the data-parallel wrapper's bucket copy runs in one), the environment says so, grad mode is on inside this
    backward (backward(create_graph=True): AccumulateGrad then clones the gradient on the compute stream), a parameter of this forward feeds
    two block nodes (the engine sums the two gradients on the compute stream), a gradient is already accumulated in .grad (AccumulateGrad adds
    on the compute stream) — and, as a second line behind the contract, when torch's own hook lists on the parameter are non-empty."""
    if not _DEFER_JOIN or _DEFER_HOLDS > 0 or torch.is_grad_enabled():
        return False
    if mask is not None and getattr(mask, "shared_params", False):
        return False
    # A process group with more than one rank means SOME wrapper reduces these gradients during backward.  This package's own wrapper holds the
    # contract above; torch's DistributedDataParallel / FSDP / apex hook the AccumulateGrad node, which neither the contract nor the hook lists
    # below can see — so in a multi-rank job the join is never deferred (round-5 advisor: default to the safe side)
    if _multi_rank_job():
        return False
    for p in params:
        if p.grad is not None or getattr(p, "_post_accumulate_grad_hooks", None) or getattr(p, "_backward_hooks", None):
            return False
    return True


class BlockActs:
    """Activations one block keeps for its backward: ONE slab (layout: ctmi_bloom_block_layout) plus the geometry."""
    __slots__ = ("slab", "lay", "B", "S", "H", "nh", "dtype", "flags", "attn_scale", "future_fill")

    def view(self, slot: str, rows: int, cols: int, dtype=None) -> Tensor:
        dtype = dtype or self.dtype
        n = rows * cols * (torch.finfo(dtype).bits // 8)
        o = self.lay.off[slot]
        return self.slab[o:o + n].view(dtype).view(rows, cols)

    @property
    def out(self) -> Tensor:
        return self.view("out", self.B * self.S, self.H)

    @property
    def qkv(self) -> Tensor:
        return self.view("qkv", self.B * self.S, 3 * self.H)

    def geometry(self):
        """Everything but the slab tensor (small Python values): what an autograd node keeps on itself — the slab goes through
        save_for_backward so that it is released with the node's other saved tensors right after the backward."""
        return (self.lay, self.B, self.S, self.H, self.nh, self.dtype, self.flags, self.attn_scale, self.future_fill)

    @staticmethod
    def rebuild(slab: Tensor, geo) -> "BlockActs":
        a = BlockActs()
        a.slab = slab
        a.lay, a.B, a.S, a.H, a.nh, a.dtype, a.flags, a.attn_scale, a.future_fill = geo
        return a


class KVOut(list):
    """The out-parameter through which a block node hands back its presents; it also carries the caller's grad mode into the node's
    forward (inside autograd.Function.forward grad mode is always off, and needs_input_grad ignores torch.no_grad())."""

    def __init__(self):
        super().__init__()
        self.grad = torch.is_grad_enabled()


class LazyKV:
    """``(present_k, present_v)`` [B,nh,S,hd] of one block (modeling_bloom.py:88-92 / modeling_gpt.py:73-75 return them on every call).
    * forward that will be differentiated: nothing is copied — the pair is built on demand as VIEWS of the activation slab (a training
      step never looks at them).  The object holds the slab only until the node's backward calls release(): the list of presents a
      training loop leaves lying around (``outputs, _ = model(...)``) then pins nothing through the next forward.
    * forward without a graph (prefill of a generation): contiguous copies made at once, so the block's slab (5x the K/V bytes) is
      freed as soon as the next block has consumed its output."""
    __slots__ = ("_slab", "_geo", "_kv", "_blocked")

    def __init__(self, acts: BlockActs, blocked: bool, eager: bool):
        self._slab, self._geo, self._blocked, self._kv = (None if eager else acts.slab), acts.geometry(), blocked, None
        if eager:
            k, v = self._views(acts)
            self._kv = (k.contiguous(), v.contiguous())

    def _views(self, acts: BlockActs):
        B, S, nh = acts.B, acts.S, acts.nh
        if self._blocked:                                               # q | k | v  [B,S,3,nh,hd]
            qv = acts.qkv.view(B, S, 3, nh, -1)
            return qv[:, :, 1].transpose(1, 2), qv[:, :, 2].transpose(1, 2)
        qv = acts.qkv.view(B, S, nh, 3, -1)                             # head-interleaved [B,S,nh,3,hd]
        return qv[:, :, :, 1, :].transpose(1, 2), qv[:, :, :, 2, :].transpose(1, 2)

    def release(self) -> None:
        """Called by the block's backward: the activations are about to be freed (views already handed out keep their storage)."""
        self._slab = None

    def _make(self):
        if self._kv is None:
            if self._slab is None:
                raise RuntimeError("the K/V presents of a differentiated forward are views of activations that its backward has already "
                                   "released; read them before backward(), or run the forward under torch.no_grad()")
            self._kv = self._views(BlockActs.rebuild(self._slab, self._geo))
            self._slab = None
        return self._kv

    def __getitem__(self, i):
        return self._make()[i]

    def __iter__(self):
        return iter(self._make())

    def __len__(self):
        return 2


def _fill_block_desc(d: "_lib.BloomBlock", x2: Tensor, params, mask: Optional[MaskInfo], slopes: Optional[Tensor], eps: float,
                     post_ln_res: bool, B: int, S: int, H: int, nh: int, slab: Tensor, flags: int = 0, attn_scale: float = 0.0,
                     future_fill: float = 0.0) -> None:
    d.B, d.S, d.H, d.nh = B, S, H, nh
    d.eps, d.post_ln_res, d.dtype = float(eps), int(post_ln_res), dt_code(x2.dtype)
    d.flags, d.attn_scale, d.future_fill = int(flags), float(attn_scale), float(future_fill)
    for name, t in zip(_lib.BLK_PARAMS, params):
        setattr(d, name, t.data_ptr())
    d.slopes = None if slopes is None else slopes.data_ptr()
    d.kpos = mask.kpos.data_ptr() if (mask is not None and slopes is not None) else None
    d.kvalid = None if mask is None else mask.kvalid.data_ptr()
    d.first_valid = None if mask is None else mask.first_valid.data_ptr()
    d.x = x2.data_ptr()
    d.slab = slab.data_ptr()


def bloom_block_fwd(x2: Tensor, params, mask: Optional[MaskInfo], slopes: Optional[Tensor], eps: float, post_ln_res: bool,
                    B: int, S: int, nh: int, flags: int = 0, attn_scale: float = 0.0, future_fill: float = 0.0) -> BlockActs:
    """modeling_bloom.py:142-159 for x2 [B*S, H]; `params` = the 12 block parameters in _lib.BLK_PARAMS order, weight
    matrices already in the compute dtype ([out,in]).  One library call; returns the saved activations (whose `.out` is the result).
    flags / attn_scale / future_fill spell the same pre-LN block the way GPT-2 lays it out (modeling_gpt.py:144-149)."""
    _need_cuda(x2, *params)
    H = x2.shape[1]
    acts = BlockActs()
    acts.B, acts.S, acts.H, acts.nh, acts.dtype = B, S, H, nh, x2.dtype
    acts.flags, acts.attn_scale, acts.future_fill = flags, attn_scale, future_fill
    acts.lay = block_layout(B, S, H, nh, x2.dtype)
    acts.slab = torch.empty(acts.lay.bytes, dtype=torch.uint8, device=x2.device)
    d = _lib.BloomBlock()
    _fill_block_desc(d, x2, params, mask, slopes, eps, post_ln_res, B, S, H, nh, acts.slab, flags, attn_scale, future_fill)
    check(_lib.load().ctmi_bloom_block_fwd(C.byref(d), _stream()), "bloom_block_fwd")
    return acts


_GROUPED = {}
_WGRAD_STREAM_ENV = os.environ.get("CTMI_WGRAD_STREAM", "auto")           # "1" / "0" force the weight-gradient side stream on / off


def block_wgrad_grouped(B: int, S: int, H: int, dtype: torch.dtype, flags: int = 0) -> bool:
    """Does the block backward take the grouped weight-gradient launch at this geometry (include/ctmi355.h ctmi_bloom_block_wgrad_grouped)?"""
    key = (B, S, H, dtype, flags)
    v = _GROUPED.get(key)
    if v is None:
        v = bool(_lib.load().ctmi_bloom_block_wgrad_grouped(B, S, H, dt_code(dtype), int(flags)))
        _GROUPED[key] = v
    return v


def bloom_block_bwd(acts: BlockActs, x2: Tensor, params, mask: Optional[MaskInfo], slopes: Optional[Tensor], eps: float,
                    post_ln_res: bool, dout2: Tensor, use_side_stream: Optional[bool] = None, defer_join: bool = False):
    """Backward of bloom_block_fwd -> (dx [T,H] in the compute dtype, the 12 fp32 parameter gradients in BLK_PARAMS order).
    use_side_stream None (default): the library's four weight-gradient products run on a side stream — unless the geometry takes the GROUPED
    weight-gradient launch (round 5: bf16, aligned shapes), which fills the GPU by itself: the whole backward then runs on the compute stream
    (same-box A/B, profiles/r05_wgrad_grouped.txt: 36.67 vs 36.98 ms per step; CTMI_WGRAD_STREAM=1 / 0 forces either).
    defer_join (only from inside a backward pass, with the side stream): the parameter gradients complete on the side stream; the compute
    stream waits for it ONCE, in a callback at the end of the backward pass (see params_allow_deferred_grads for when that is sound)."""
    _need_cuda(x2, dout2)
    B, S, H, nh = acts.B, acts.S, acts.H, acts.nh
    dev = x2.device
    if use_side_stream is None:
        use_side_stream = (_WGRAD_STREAM_ENV != "0") if _WGRAD_STREAM_ENV in ("0", "1") else not block_wgrad_grouped(B, S, H, x2.dtype, acts.flags)
    if torch.cuda.is_current_stream_capturing():
        # a captured step (graph.py) is one stream: events recorded by earlier EAGER steps on the side stream must not become dependencies of the
        # capture, and a deferred join is a callback of this backward pass, not of the replays.  Same launches, issued in order.
        use_side_stream = defer_join = False
    d = _lib.BloomBlock()
    _fill_block_desc(d, x2, params, mask, slopes, eps, post_ln_res, B, S, H, nh, acts.slab, acts.flags, acts.attn_scale, acts.future_fill)
    g = _lib.BloomBlockGrads()
    dx = torch.empty_like(x2)
    # Conv1D weights: gradients in the parameter's own [in,out] layout — the transpose of an [out,in] compute copy's shape, or simply
    # the shape of the weight when that is passed [in,out] itself (BLK_W_IN_OUT)
    flip = bool(acts.flags & _lib.BLK_WGRAD_IN_OUT) != bool(acts.flags & _lib.BLK_W_IN_OUT)
    grads = [torch.empty(tuple(reversed(p.shape)) if (flip and p.dim() == 2) else p.shape, dtype=torch.float32, device=dev) for p in params]
    g.dout, g.dx = dout2.data_ptr(), dx.data_ptr()
    for name, t in zip(_lib.BLK_PARAMS, grads):
        setattr(g, "d" + name, t.data_ptr())
    defer_join = bool(defer_join and use_side_stream)
    main = torch.cuda.current_stream(dev)
    st = None
    slot = 0
    if defer_join:
        st = _DEFER.setdefault(dev, [0, {}])
        slot = st[0]
        st[0] ^= 1
        prev = st[1].get(slot)
        if prev is not None:
            main.wait_event(prev)                                   # the side-stream work of the call that last used this scratch (two blocks ago)
    elif dev in _DEFER and _DEFER[dev][1] and not torch.cuda.is_current_stream_capturing():        # (GraphedStep synchronises before it captures)
        for ev in _DEFER[dev][1].values():                          # a joined call after deferred ones shares slot 0: let their side work finish first
            main.wait_event(ev)
        _DEFER[dev][1].clear()
    ws = _block_ws(dev, acts.lay.bwd_ws_bytes, slot)
    g.ws, g.ws_bytes = ws.data_ptr(), ws.numel()
    sk = _splitk_ws(dev)
    g.splitk_ws, g.splitk_ws_bytes = sk.data_ptr(), sk.numel() * 4
    if use_side_stream:
        side = side_stream(dev)
        with torch.cuda.stream(side):
            sk2 = _splitk_ws(dev)
        g.side_stream = side.cuda_stream
        g.side_splitk_ws, g.side_splitk_ws_bytes = sk2.data_ptr(), sk2.numel() * 4
    g.defer_join = int(defer_join)
    check(_lib.load().ctmi_bloom_block_bwd(C.byref(d), C.byref(g), _stream()), "bloom_block_bwd")
    if defer_join:
        ev = torch.cuda.Event()
        ev.record(side)
        st[1][slot] = ev
        # memory the side stream still reads or writes was allocated on the compute stream: keep the caching allocator from handing it out
        # again before the side stream has passed this point
        acts.slab.record_stream(side)
        dout2.record_stream(side)
        for t in grads:
            t.record_stream(side)
        _LAST_DEFERRED_GRAD_PTRS[:] = [t.data_ptr() for t in grads]   # (tests: autograd must ADOPT these tensors as .grad, not copy them on the compute stream)
        # THE join, at the end of this backward pass: optimizer, clipping, scaler of the caller see finished gradients.  One callback per deferred
        # call (a few microseconds of host time each, the waits after the first are no-ops): no "already queued" flag that a backward pass
        # which raised half-way could leave set.
        torch.autograd.Variable._execution_engine.queue_callback(lambda main=main, side=side: main.wait_stream(side))
    return dx, grads


def dropout(x: Tensor, p: float, seed: int, residual: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """(keep ? x/(1-p) : 0) (+ residual), keep(i) = keep_hash(i, seed) >= p*2^32 over the flat element index — and, applied to a gradient
    with the same seed, its own backward."""
    _need_cuda(x, residual)
    x = _c(x)
    if residual is not None:
        residual = _c(residual)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().ctmi_dropout(_p(x), _p(residual), _p(out), x.numel(), float(p), int(seed) & 0xFFFFFFFF, dt_code(x.dtype), _stream()),
          "dropout")
    return out


class DropoutFn(torch.autograd.Function):
    """torch.nn.functional.dropout(x, p, training=True) (+ residual) on the counter-based mask; nothing but the seed is saved."""

    @staticmethod
    def forward(ctx, x, p, seed, residual=None):
        ctx.p, ctx.seed, ctx.has_res = p, seed, residual is not None
        return dropout(x, p, seed, residual)

    @staticmethod
    def backward(ctx, dy):
        return dropout(dy, ctx.p, ctx.seed), None, None, (dy if ctx.has_res else None)


# ------------------------------------------------------------------------------------------------ utilities
def cast(src: Tensor, dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    _need_cuda(src)
    src = _c(src)
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.load().ctmi_cast(_p(src), dt_code(src.dtype), _p(out), dt_code(dtype), src.numel(), _stream()), "cast")
    return out


def transpose_cast(src: Tensor, dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    """src fp32 [R,C] -> out [C,R] in `dtype`."""
    _need_cuda(src)
    src = _c(src)
    R, Cc = src.shape
    if out is None:
        out = torch.empty((Cc, R), dtype=dtype, device=src.device)
    check(_lib.load().ctmi_transpose_cast(_p(src), _p(out), dt_code(dtype), R, Cc, _stream()), "transpose_cast")
    return out


def compute_weight_t(p: Tensor, dtype: torch.dtype) -> Tensor:
    """The [out,in] compute-dtype copy of a parameter stored as [in,out] (GPT-2 Conv1D).  Cached on the parameter and
    refreshed when its version counter or storage moved, or when the fused optimizer marked it stale (the fused AdamW
    updates parameters without moving the counter; see optimizer.py)."""
    t = getattr(p, "_ct_wt", None)
    if t is None or t.dtype != dtype or t.device != p.device or getattr(p, "_ct_wt_ver", -1) != p._version \
            or getattr(p, "_ct_wt_ptr", 0) != p.data_ptr() or getattr(p, "_ct_wt_stale", False):
        t = transpose_cast(p.detach(), dtype, out=t if (t is not None and t.dtype == dtype and t.device == p.device) else None)
        p._ct_wt, p._ct_wt_ver, p._ct_wt_ptr, p._ct_wt_stale = t, p._version, p.data_ptr(), False
    return t


def sumsq(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=x.device)
        accumulate = True
    check(_lib.load().ctmi_sumsq(_p(x), x.numel(), _p(out), int(accumulate), _stream()), "sumsq")
    return out


def scale_(x: Tensor, s: float, s_dev: Optional[Tensor] = None) -> Tensor:
    check(_lib.load().ctmi_scale(_p(x), x.numel(), float(s), _p(s_dev), _stream()), "scale")
    return x


def scale_copy(src: Tensor, dst: Tensor, s: float) -> Tensor:
    """dst = s * src (fp32, flat; dst may alias src)."""
    _need_cuda(src, dst)
    check(_lib.load().ctmi_scale_copy(_p(src), _p(dst), src.numel(), float(s), _stream()), "scale_copy")
    return dst


def set_launch_policy(shared: bool, reserve_cus: Optional[int] = None) -> None:
    """GEMM launch policy (include/ctmi355.h ctmi_set_launch_policy): shared=True while RCCL kernels hold CUs under backward."""
    lib = _lib.load()
    if reserve_cus is None:
        r = C.c_int(0)
        lib.ctmi_get_launch_policy(None, C.byref(r))
        reserve_cus = r.value
    check(lib.ctmi_set_launch_policy(int(shared), int(reserve_cus)), "set_launch_policy")                    # (True == 1: the round-2 "shared")


def profile_begin() -> None:
    """Start bracketing every library launch with HIP events (include/ctmi355.h ctmi_profile_begin)."""
    check(_lib.load().ctmi_profile_begin(), "profile_begin")


def profile_end():
    """-> {class name: (milliseconds, brackets)} since profile_begin()."""
    n = len(_lib.PROF_CLASSES)
    ms, cnt = (C.c_float * n)(), (C.c_int * n)()
    check(_lib.load().ctmi_profile_end(ms, cnt), "profile_end")
    return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.PROF_CLASSES)}


def clock_probe(device=None, mfma_iters: int = 20000) -> float:
    """Shader clock in MHz under a short all-CU bf16 MFMA load (include/ctmi355.h ctmi_clock_probe: s_memtime over s_memrealtime in one
    wave while 2048 workgroups issue matrix instructions).  Synchronises the device: call it outside timed regions."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    check(_lib.load().ctmi_clock_probe(int(mfma_iters), _p(out), _stream()), "clock_probe")
    torch.cuda.synchronize(dev)
    cyc, ref = out.tolist()
    return 100.0 * cyc / max(ref, 1)


def set_attn_path(mask: int) -> int:
    """Attention kernel family for the bf16 training shapes (include/ctmi355.h ctmi_attn_set_path): bit 0 forward, bit 1 backward on the
    256-row kernels; returns the previous value."""
    return int(_lib.load().ctmi_attn_set_path(int(mask)))


class DirectComm:
    """An RCCL communicator owned by the library (include/ctmi355.h ctmi_ddp_*): the data-parallel collectives without the
    torch.distributed layers, with a per-communicator channel cap (= the CUs the collectives may hold).  Every call orders the
    collective behind the CURRENT torch stream and returns at once; ``wait()`` makes the current stream wait for everything issued so
    far.  Buffers handed to a collective must stay referenced until a ``wait()`` that follows it."""

    def __init__(self, unique_id: bytes, rank: int, world: int, max_channels: int = 0):
        if len(unique_id) != 128:
            raise ValueError("DirectComm: the unique id is 128 bytes (DirectComm.unique_id() on rank 0)")
        h = C.c_void_p()
        idb = C.create_string_buffer(bytes(unique_id), 128)
        check(_lib.load().ctmi_ddp_create(idb, int(rank), int(world), int(max_channels), C.byref(h)), "ddp_create")
        self._h, self.rank, self.world, self.max_channels = h, int(rank), int(world), int(max_channels)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().ctmi_ddp_unique_id(buf), "ddp_unique_id")
        return buf.raw

    def all_reduce(self, t: torch.Tensor) -> None:
        _need_cuda(t)
        if not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("DirectComm.all_reduce: contiguous fp32 / bf16 tensors only")
        check(_lib.load().ctmi_ddp_all_reduce(self._h, _p(t), t.numel(), dt_code(t.dtype), _stream()), "ddp_all_reduce")

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        _need_cuda(out, inp)
        nb = inp.numel() * inp.element_size()
        if not (out.is_contiguous() and inp.is_contiguous()) or out.numel() * out.element_size() != nb * self.world:
            raise ValueError("DirectComm.all_gather: contiguous tensors, out = world x inp")
        check(_lib.load().ctmi_ddp_all_gather(self._h, _p(inp), _p(out), nb, _stream()), "ddp_all_gather")

    def broadcast(self, t: torch.Tensor, root: int = 0) -> None:
        _need_cuda(t)
        if not t.is_contiguous():
            raise ValueError("DirectComm.broadcast: contiguous tensors only")
        check(_lib.load().ctmi_ddp_broadcast(self._h, _p(t), t.numel() * t.element_size(), int(root), _stream()), "ddp_broadcast")

    def wait(self) -> None:
        check(_lib.load().ctmi_ddp_wait(self._h, _stream()), "ddp_wait")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().ctmi_ddp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_launch_policy() -> Tuple[int, int]:
    """(shared level 0 | 1 | 2, reserved CUs) — 0 persistent, 1 shared (co-resident tiles), 2 flow (one workgroup per tile, single-GPU tiles)"""
    sh, r = C.c_int(0), C.c_int(0)
    _lib.load().ctmi_get_launch_policy(C.byref(sh), C.byref(r))
    return int(sh.value), r.value


def invalidate_compute_copies(p: Tensor) -> None:
    """Forget the cached compute-dtype copies of a parameter whose storage was rewritten behind the version counter."""
    for a in ("_ct_shadow", "_ct_shadow_pad", "_ct_shadow_ver", "_ct_shadow_ptr", "_ct_wt", "_ct_wt_ver", "_ct_wt_ptr"):
        if hasattr(p, a):
            delattr(p, a)


def argmax_lastdim(x2d: Tensor) -> Tensor:
    rows, cols = x2d.shape
    out = torch.empty(rows, dtype=torch.int64, device=x2d.device)
    check(_lib.load().ctmi_argmax(_p(x2d), x2d.stride(0), _p(out), rows, cols, dt_code(x2d.dtype), _stream()), "argmax")
    return out


def row_lse(x2d: Tensor) -> Tensor:
    """[rows, 2] fp32 = (max, log sum exp(x - max)) per row: the two terms torch.log_softmax subtracts (generation_util.py:200)."""
    _need_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cols = x2d.shape
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x2d.device)
    check(_lib.load().ctmi_row_lse(_p(x2d), x2d.stride(0), _p(stats), rows, cols, dt_code(x2d.dtype), _stream()), "row_lse")
    return stats


def group_topk(x2d: Tensor, group: int, k: int, stats: Optional[Tensor] = None, add: Optional[Tensor] = None,
               add_mul: float = 1.0) -> Tuple[Tensor, Tensor]:
    """Best k of the group*cols candidates of every `group` consecutive rows; score = ((x - max) - logsum) + add*add_mul when
    stats/add are given.  Returns (values [G,k] fp32 descending, flat indices [G,k] int64; ties by ascending index)."""
    _need_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.shape[0] % group == 0
    rows, cols = x2d.shape
    G = rows // group
    val = torch.empty(G, k, dtype=torch.float32, device=x2d.device)
    idx = torch.empty(G, k, dtype=torch.int64, device=x2d.device)
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape == (rows, 2)
    if add is not None:
        add = add.to(torch.float32).contiguous().view(-1)
        assert add.numel() == rows
    check(_lib.load().ctmi_group_topk(_p(x2d), x2d.stride(0), _p(stats), _p(add), float(add_mul), _p(val), _p(idx), G, group, cols, k,
                                      dt_code(x2d.dtype), _stream()), "group_topk")
    return val, idx


def scores_filter(x2d: Tensor, divisor: float = 1.0, thr: Optional[Tensor] = None, fill: float = float("-inf")) -> Tensor:
    """x / divisor, entries below the per-row threshold replaced by `fill` (fp32 scores; logits_processor.py:35-56)."""
    _need_cuda(x2d)
    assert x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.stride(1) == 1
    rows, cols = x2d.shape
    out = torch.empty(rows, cols, dtype=torch.float32, device=x2d.device)
    ts = 0
    if thr is not None:
        assert thr.dtype == torch.float32 and thr.dim() == 1 and thr.numel() == rows
        ts = thr.stride(0)
    check(_lib.load().ctmi_scores_filter(_p(x2d), x2d.stride(0), float(divisor), _p(thr), ts, float(fill), _p(out), cols, rows, cols,
                                         _stream()), "scores_filter")
    return out


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _shadow_flag(shadows) -> int:
    """the operand copies one fused optimizer launch writes are all bf16 or all IEEE half (the optimizers group by dtype)"""
    dts = {s_.dtype for s_ in (shadows or []) if s_ is not None}
    if len(dts) > 1 or (dts and next(iter(dts)) not in (torch.bfloat16, torch.float16)):
        raise _lib.CtmiError(f"fused optimizer: the operand copies of one launch must share one 16-bit dtype, got {dts}")
    return _lib.OPT_SHADOW_F16 if dts == {torch.float16} else 0


def adamw_step(params, grads, exp_avg, exp_avg_sq, shadows, *, lr, beta1, beta2, eps, weight_decay, step, decoupled,
               mutate_grad=False, grad_scale=1.0, legacy_grid=False) -> None:
    n = len(params)
    if n == 0:
        return
    sizes = (C.c_int64 * n)(*[p.numel() for p in params])
    sh = _ptr_array(shadows) if shadows is not None else None
    check(_lib.load().ctmi_adamw_step(_ptr_array(params), _ptr_array(grads), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), sh,
                                      sizes, n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                      int(decoupled), int(bool(mutate_grad)) | _shadow_flag(shadows) | (_lib.OPT_LEGACY_GRID if legacy_grid else 0),
                                      float(grad_scale), _stream()), "adamw_step")


def adamw_set_hyper(hyper_dev: Tensor, *, lr, beta1, beta2, eps, weight_decay, step, decoupled, mutate_grad=False, grad_scale=1.0,
                    shadow_f16=False) -> None:
    """Write the hyper-parameter record of AdamW step `step` into `hyper_dev` (12 x 4 bytes on the device; include/ctmi355.h ctmi_adamw_set_hyper)."""
    if hyper_dev.numel() * hyper_dev.element_size() < 48 or not hyper_dev.is_cuda:
        raise _lib.CtmiError("adamw_set_hyper: a device buffer of >= 48 bytes is needed")
    check(_lib.load().ctmi_adamw_set_hyper(_p(hyper_dev), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                           int(decoupled), int(bool(mutate_grad)) | (_lib.OPT_SHADOW_F16 if shadow_f16 else 0),
                                           float(grad_scale), _stream()), "adamw_set_hyper")


def adamw_step_dev(params, grads, exp_avg, exp_avg_sq, shadows, hyper_dev: Tensor) -> None:
    """ctmi_adamw_step with the hyper-parameters read from `hyper_dev` (capturable in a hipGraph)."""
    n = len(params)
    if n == 0:
        return
    sizes = (C.c_int64 * n)(*[p.numel() for p in params])
    sh = _ptr_array(shadows) if shadows is not None else None
    check(_lib.load().ctmi_adamw_step_dev(_ptr_array(params), _ptr_array(grads), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), sh, sizes, n,
                                          _p(hyper_dev), _stream()), "adamw_step_dev")


def amp_unscale(grads, state: Tensor) -> None:
    """grads[i] *= 1/state[0] in place; state[2] = 1 if anything is not finite (GradScaler.unscale_)."""
    n = len(grads)
    if n == 0:
        return
    _need_cuda(state, *grads)
    assert state.dtype == torch.float32 and state.numel() >= 3 and all(g.dtype == torch.float32 and g.is_contiguous() for g in grads)
    sizes = (C.c_int64 * n)(*[g.numel() for g in grads])
    check(_lib.load().ctmi_amp_unscale(_ptr_array(grads), sizes, n, _p(state), _stream()), "amp_unscale")


def amp_update(state: Tensor, growth: float, backoff: float, interval: int) -> None:
    _need_cuda(state)
    check(_lib.load().ctmi_amp_update(_p(state), float(growth), float(backoff), int(interval), _stream()), "amp_update")


def sgd_step(params, grads, bufs, shadows, *, lr, momentum, dampening, weight_decay, first_step) -> None:
    n = len(params)
    if n == 0:
        return
    sizes = (C.c_int64 * n)(*[p.numel() for p in params])
    check(_lib.load().ctmi_sgd_step(_ptr_array(params), _ptr_array(grads), _ptr_array(bufs) if bufs is not None else None,
                                    _ptr_array(shadows) if shadows is not None else None, sizes, n, float(lr), float(momentum or 0.0),
                                    float(dampening or 0.0), float(weight_decay or 0.0), int(bool(first_step)) | _shadow_flag(shadows), _stream()), "sgd_step")


# ------------------------------------------------------------------------------------------------ bf16 shadows
PAD_ROWS = 32


def pad_rows(n: int) -> int:
    return (n + PAD_ROWS - 1) // PAD_ROWS * PAD_ROWS


class _ZeroPadded(threading.local):
    """dlogits buffers whose columns [V, pad) are known to be zero: data_ptr -> (padded column count, the buffer) — set by the loss
    backward and consumed by the LM-head backward of the same step.  Both run back to back on ONE autograd worker thread, so the
    hand-off is thread-local (no cross-thread aliasing of a recycled data_ptr), and holds at most one entry (it pins its buffer)."""

    def __init__(self):
        self.d = {}

    def clear(self):
        self.d.clear()

    def __setitem__(self, k, v):
        self.d[k] = v

    def pop(self, k, default=None):
        return self.d.pop(k, default)

    def get(self, k, default=None):
        return self.d.get(k, default)


ZERO_PADDED = _ZeroPadded()

def compute_weight(p: Tensor, dtype: torch.dtype) -> Tensor:
    """The matrix `p` (an fp32 master parameter) in the compute dtype.  fp32 -> p itself.  bf16 / fp16 -> a cached shadow,
    refreshed when p's version counter moved (torch optimizers / load_state_dict), when the compute dtype changed (an autocast
    context around a model of another dtype), or when the fused optimizer — which does not move the counter — invalidated it; a
    bf16 shadow is written directly by the fused optimizer instead."""
    if dtype == torch.float32 or p.dtype == dtype:
        return p.detach()
    sh = getattr(p, "_ct_shadow", None)
    if sh is not None and sh.dtype != dtype:
        sh = None                                                           # one shadow per parameter: the compute dtype changed
    if sh is None or sh.device != p.device or sh.shape != p.shape or getattr(p, "_ct_shadow_ver", -1) != p._version \
            or getattr(p, "_ct_shadow_ptr", 0) != p.data_ptr():
        if (sh is None or sh.shape != p.shape or sh.device != p.device) and p.dim() == 2 and p.shape[0] % PAD_ROWS != 0:
            # a table whose row count is not a multiple of 32 (GPT-2's V = 50257): the shadow is the head of a zero-padded
            # buffer, so the LM-head dgrad can run with an aligned, 32-divisible K (see models.modeling_bloom.LMHeadFn)
            buf = torch.zeros((pad_rows(p.shape[0]), p.shape[1]), dtype=dtype, device=p.device)
            sh = buf[:p.shape[0]]
            p._ct_shadow_pad = buf
        sh = cast(p.detach(), dtype, out=sh if (sh is not None and sh.shape == p.shape and sh.device == p.device) else None)
        p._ct_shadow = sh
        p._ct_shadow_ver = p._version
        p._ct_shadow_ptr = p.data_ptr()
    return sh
