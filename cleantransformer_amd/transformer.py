"""MI355X-native counterparts of CleanTransformer/transformer.py (same class names, constructor signatures,
parameter names and forward semantics) — LayerNorm (transformer.py:61-89), AttentionLayer (12-58),
TransformerBlock (92-121), ExampleConfig (124-131).  All arithmetic runs in the ctmi355 HIP kernels.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib, ops

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------ autograd glue
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, eps: float):
        cols = weight.numel()
        x2 = x.reshape(-1, cols)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w, b = weight.detach().reshape(-1), bias.detach().reshape(-1)
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.wshape = weight.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, w, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(x2.shape)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx, dw, db = ops.layernorm_bwd(dy2, x2, w, mean, rstd)
        return dx.view(dy.shape), dw.view(ctx.wshape), db.view(ctx.wshape), None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b.  W, b are fp32 master parameters; the GEMM runs in x.dtype (bf16 uses the cached shadow)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor]):
        K = weight.shape[1]
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        wc = ops.compute_weight(weight, x.dtype)
        y = ops.linear_fwd(x2, wc, None if bias is None else bias.detach())
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, weight = ctx.saved_tensors
        wc = ops.compute_weight(weight, x2.dtype)
        dy2 = dy.reshape(-1, weight.shape[0])
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = ops.linear_dgrad(dy2, wc)
        dw = ops.linear_wgrad(dy2, x2)
        db = ops.colsum(dy2) if ctx.has_bias else None
        return dx.view(*dy.shape[:-1], weight.shape[1]), dw, db


class FFNFn(torch.autograd.Function):
    """transformer.py:98-102 FFN: W2 relu(W1 x + b1) + b2, with the ReLU and its mask fused into the GEMM epilogues."""

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
        H = w1.shape[1]
        x2 = x.reshape(-1, H)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w1c, w2c = ops.compute_weight(w1, x.dtype), ops.compute_weight(w2, x.dtype)
        a = ops.linear_fwd(x2, w1c, b1.detach(), epilogue=_lib.EPI_RELU)
        y = ops.linear_fwd(a, w2c, b2.detach())
        ctx.save_for_backward(x2, a, w1, w2)
        return y.view(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, a, w1, w2 = ctx.saved_tensors
        w1c, w2c = ops.compute_weight(w1, x2.dtype), ops.compute_weight(w2, x2.dtype)
        dy2 = dy.reshape(-1, w2.shape[0])
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dw2 = ops.linear_wgrad(dy2, a)
        db2 = ops.colsum(dy2)
        da = ops.linear_dgrad(dy2, w2c, epilogue=_lib.EPI_DRELU, aux_in=a)
        dw1 = ops.linear_wgrad(da, x2)
        db1 = ops.colsum(da)
        dx = ops.linear_dgrad(da, w1c)
        return dx.view(dy.shape[:-1] + (w1.shape[1],)), dw1, db1, dw2, db2


class MHAFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd) + additive_mask) v on [B,S,H] tensors split into heads (transformer.py:25-57)."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, add_mask: Optional[Tensor], nh: int, scale: float, drop_p: float = 0.0,
                drop_seed: int = 0):
        B, S, H = q.shape
        hd = H // nh
        q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
        st = (S * H, hd, H)
        am, am_str = None, (0, 0, 0, 0)
        if add_mask is not None:
            am = add_mask.to(torch.float32)
            am = am.expand(B, nh, S, S) if am.dim() == 4 else am.reshape((1,) * (4 - am.dim()) + tuple(am.shape)).expand(B, nh, S, S)
            am_str = tuple(am.stride())
        desc = ops._strided_desc(B, nh, S, S, hd, st, st, st, st, scale, False, am_str, dropout_p=drop_p, dropout_seed=drop_seed)
        out = torch.empty_like(q)
        stat_m, stat_l = ops.attn_fwd(q, k, v, out, desc, None, None, am)
        ctx.save_for_backward(q, k, v, out, stat_m, stat_l, am)
        ctx.desc = desc
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        q, k, v, out, stat_m, stat_l, am = ctx.saved_tensors
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attn_bwd(q, k, v, out, dout, stat_m, stat_l, dq, dk, dv, ctx.desc, None, None, am)
        return dq, dk, dv, None, None, None, None, None


def _dropout(x: Tensor, p: float, training: bool, residual: Optional[Tensor] = None) -> Tensor:
    """torch.nn.Dropout(p)(x) (+ residual) in training mode on the kernels' counter-based mask (ctmi_dropout): a fresh seed per call
    from torch's CPU generator (rng.next_seed), the backward regenerates the mask.  p = 0 or eval(): identity (+ residual)."""
    if p > 0.0 and training:
        from . import rng
        return ops.DropoutFn.apply(x, float(p), rng.next_seed(), residual)
    return x if residual is None else _AddFn.apply(residual, x)


# ------------------------------------------------------------------------------------------------ modules
class LayerNorm(torch.nn.Module):
    """transformer.py:61-89.  `normalized_shape` may be an int or a tuple of trailing dims; eps sits inside the
    mean, i.e. y = w * (x-mean)/sqrt(var_biased + eps) + b."""

    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape, self.eps = tuple(normalized_shape), eps
        self.weight = torch.nn.Parameter(torch.ones(self.normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(self.normalized_shape))

    def forward(self, x: Tensor) -> Tensor:
        nd = len(self.normalized_shape)
        if tuple(x.shape[-nd:]) != self.normalized_shape:
            raise ValueError(f"LayerNorm: trailing dims {tuple(x.shape[-nd:])} != normalized_shape {self.normalized_shape}")
        return LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class AttentionLayer(torch.nn.Module):
    """transformer.py:12-58: separate q/k/v Linear(H,H), softmax(QK^T/sqrt(hd) + mask), P V, merge heads; no out-proj."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        assert config.hidden_size % config.num_attention_heads == 0
        self.dim, self.m_head = config.hidden_size, config.num_attention_heads
        self.q_linear = torch.nn.Linear(config.hidden_size, config.hidden_size)
        self.k_linear = torch.nn.Linear(config.hidden_size, config.hidden_size)
        self.v_linear = torch.nn.Linear(config.hidden_size, config.hidden_size)
        self.dropout = torch.nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask=None, head_mask=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is not supported (the reference's `if head_mask:` is ill-defined for tensors; "
                                      "every caller passes None) — SURVEY Q11")
        q = LinearFn.apply(hidden_states, self.q_linear.weight, self.q_linear.bias)
        k = LinearFn.apply(hidden_states, self.k_linear.weight, self.k_linear.bias)
        v = LinearFn.apply(hidden_states, self.v_linear.weight, self.v_linear.bias)
        drop_p = float(self.dropout.p) if self.training else 0.0                  # transformer.py:46-47: dropout on the softmax output
        seed = 0
        if drop_p > 0.0:
            from . import rng
            seed = rng.next_seed()
        return MHAFn.apply(q, k, v, attention_mask, self.m_head, 1.0 / math.sqrt(self.dim / self.m_head), drop_p, seed)


class TransformerBlock(torch.nn.Module):
    """transformer.py:92-121: post-LN block, FFN = Linear(H,4H)-ReLU-Linear(4H,H); config attr `layer_norm_epsilong` [sic]."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.attention = AttentionLayer(config)
        self.ffw = torch.nn.Sequential(
            torch.nn.Linear(config.hidden_size, config.hidden_size * 4),
            torch.nn.ReLU(),
            torch.nn.Linear(config.hidden_size * 4, config.hidden_size),
        )
        self.norm1 = LayerNorm(config.hidden_size, config.layer_norm_epsilong)
        self.norm2 = LayerNorm(config.hidden_size, config.layer_norm_epsilong)
        self.dropout = torch.nn.Dropout(config.hidden_dropout_prob)

    def forward(self, x):
        y = self.norm1(_dropout(self.attention(x), self.dropout.p, self.training, residual=x))
        f = FFNFn.apply(y, self.ffw[0].weight, self.ffw[0].bias, self.ffw[2].weight, self.ffw[2].bias)
        return self.norm2(_dropout(f, self.dropout.p, self.training, residual=y))


class _AddFn(torch.autograd.Function):
    """Residual add of the generic post-LN block (its attention has no output projection to fuse the add into).
    Bloom — the measured path — fuses every residual add into a GEMM epilogue instead."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        return torch.add(a, b)

    @staticmethod
    def backward(ctx, g: Tensor):
        return g, g


class ExampleConfig():
    """transformer.py:124-131."""

    def __init__(self):
        self.num_attention_heads = 3
        self.layer_norm_epsilong = 1e-5
        self.resid_pdrop = 0.1
        self.attention_probs_dropout_prob = 0.1
        self.hidden_size = 12
        self.hidden_dropout_prob = 0.1
