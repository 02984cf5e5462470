"""MI355X-native counterpart of CleanTransformer/loss.py: CrossEntropyLoss (loss.py:29-49), computed by the ctmi355
row-wise log-sum-exp kernels (one 256-thread workgroup per row, fp32 statistics, 16-byte loads).

Numerics: the reference's CE is un-stabilised (exp / sum / log) and returns inf once a logit exceeds ~88.7; this kernel
subtracts the row max, so it equals the reference wherever the reference is finite and equals
torch.nn.CrossEntropyLoss everywhere.  `mean` divides by ``input.shape[0]`` exactly as loss.py:47-48 does
(no ignore_index handling there); pass ``ignore_index`` to get torch semantics instead.
"""
from __future__ import annotations

import torch

from . import ops


class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, reduction, ignore_index):
        n, c = logits.shape
        lg = logits if logits.is_contiguous() else logits.contiguous()
        tg = target.to(torch.int64)
        tg = tg if tg.is_contiguous() else tg.contiguous()
        if ignore_index is None:
            mode, ign = (1 if reduction == 'mean' else 2), -(1 << 62)            # loss.py: divide by N / plain sum
        else:
            mode, ign = (0 if reduction == 'mean' else 2), int(ignore_index)     # torch semantics
        loss_out, row_lse = ops.ce_fwd(lg, tg, seq=n, shift=0, ignore_index=ign, denom_mode=mode, denom_rows=n)
        ctx.save_for_backward(lg, tg, row_lse, loss_out)
        ctx.ign = ign
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        lg, tg, row_lse, loss_out = ctx.saved_tensors
        g = gout.to(torch.float32).reshape(1).contiguous()
        d = ops.ce_bwd(lg, tg, row_lse, loss_out, g, seq=lg.shape[0], shift=0, ignore_index=ctx.ign)
        return d, None, None, None


class SoftTargetCrossEntropyFn(torch.autograd.Function):
    """loss.py:43-46: probability targets [N, C]:  -sum t * log_softmax(x)  (÷ N for 'mean').  The gradient flows to the logits
    only (targets are data in every caller; the reference's autograd would also produce d/dt = -log_softmax)."""

    @staticmethod
    def forward(ctx, logits, target, reduction):
        n, c = logits.shape
        lg = logits if logits.is_contiguous() else logits.contiguous()
        tg = target.to(torch.float32)
        tg = tg if tg.is_contiguous() else tg.contiguous()
        loss_out, row_lse, row_tsum = ops.ce_soft_fwd(lg, tg, denom_mode=1 if reduction == 'mean' else 2, denom_rows=n)
        ctx.save_for_backward(lg, tg, row_lse, row_tsum, loss_out)
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        lg, tg, row_lse, row_tsum, loss_out = ctx.saved_tensors
        g = gout.to(torch.float32).reshape(1).contiguous()
        return ops.ce_soft_bwd(lg, tg, row_lse, row_tsum, loss_out, g), None, None


class CrossEntropyLoss(torch.nn.Module):
    """loss.py:29-49.  input [N, C]; target [N] class indices, or [N, C] probabilities (the reference's second branch)."""

    def __init__(self, reduction='mean', ignore_index=None):
        super().__init__()
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, input, target):
        if input.dim() != 2:
            raise ValueError("CrossEntropyLoss expects input [N, C] (loss.py:41 gathers along dim 1)")
        if target.dim() != input.dim() - 1:                  # loss.py:39: anything that is not [N] is taken as probabilities
            if target.shape != input.shape:
                raise ValueError("probability targets must have the shape of the input (loss.py:45 multiplies them element-wise)")
            if self.ignore_index is not None:
                raise ValueError("ignore_index applies to class-index targets only")
            return SoftTargetCrossEntropyFn.apply(input, target, self.reduction)
        return CrossEntropyFn.apply(input, target, self.reduction, self.ignore_index)
