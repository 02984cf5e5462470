"""MI355X-native counterpart of CleanTransformer/models/modeling_gpt.py (BASELINE configs[3]): same class
names, constructor signatures, parameter names / state_dict keys and forward semantics, every matmul / LayerNorm /
attention / embedding / loss running in the ctmi355 HIP kernels through the C ABI.

What differs from the Bloom path, and how it maps onto the same kernels:
  * ``Conv1D`` keeps its weight as ``[in, out]`` (modeling_gpt.py:32-46).  The GEMM kernels consume ``[out, in]``, so a
    transposed compute-dtype copy is cached on the parameter (``ops.compute_weight_t``, ``ctmi_transpose_cast``; rebuilt
    after every optimizer step); the weight gradient is produced directly in the parameter's own layout
    (``dW[in,out] = x^T dy`` is the same TN kernel with the operands swapped).
  * the fused ``c_attn`` activation is ``[B,S,3H]`` laid out q | k | v (modeling_gpt.py:71-73), not head-interleaved: the
    attention kernels take it through (batch, head, row) strides, no split / permute copies.
  * causal mask: the reference REPLACES future scores by -1e4 (``w*b - 1e4*(1-b)``, :88-89) and ADDS ``(1-mask)*finfo.min``
    per key (:91-92, :172-175).  exp(-1e4 - rowmax) is exactly 0 in fp32 whenever a row has one visible unpadded key; rows whose
    whole causal window is padding (LEFT padding) are the one place the reference attends to the future — the -1e4 of the
    future keys sits above the finfo.min of the padded visible ones.  The attention kernels take that fill value
    (``ctmi_attn_desc.future_fill = -1e4``), so left-padded batches match the reference too (tests/golden/tiny_gpt_leftpad.npz).
  * ``gelu_new`` (:113-119) is the same tanh GELU as Bloom's: the GELU / dGELU GEMM epilogues serve it.
  * ``version='gpt'`` is the post-LN GPT-1 block (:138-143), anything else the pre-LN GPT-2 block with ``ln_f`` (:144-149).
  * every Dropout must be inactive (p = 0 or eval): note the reference's MLP ends in ``torch.nn.Dropout()`` with p = 0.5.
The reference's forward has no loss; ``GPTLMHeadModel.forward(..., labels=...)`` additionally returns the shifted
cross-entropy (same kernel as Bloom's), which is what a training step needs.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import _lib, ops, rng
from ..generation.generation_util import GenerationMixin
from ..transformer import LayerNorm
from .modeling_bloom import EmbedFn, LMHeadFn, ShiftedCrossEntropyFn, _TieCtx, _torch_dtype

Tensor = torch.Tensor


class GPTConfig():
    def __init__(self, vocab_size=100, n_embd=100, n_positions=100, n_layer=3, n_head=2, n_ctx=2000,
                 embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1, layer_norm_epsilon=1e-5,
                 afn='gelu_new', compute_dtype="fp32",
                 **kwargs):
        self.vocab_size = vocab_size
        self.n_embd = n_embd
        self.n_positions = n_positions
        self.n_layer = n_layer
        self.n_head = n_head
        self.n_ctx = n_ctx
        self.embd_pdrop, self.attn_pdrop, self.resid_pdrop = embd_pdrop, attn_pdrop, resid_pdrop
        self.layer_norm_epsilon = layer_norm_epsilon
        self.afn = afn
        self.compute_dtype = compute_dtype
        for k, v in kwargs.items():
            setattr(self, k, v)


# ------------------------------------------------------------------------------------------------ autograd glue
class Conv1DFn(torch.autograd.Function):
    """y = x W + b (+ residual), W stored [in,out] (modeling_gpt.py:45-46).  The residual add rides in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, residual: Optional[Tensor]):
        K, N = weight.shape
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, N)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        y = ops.linear_fwd(x2, ops.compute_weight_t(weight, x.dtype), bias.detach(), residual=r2)
        ctx.save_for_backward(x2, weight)
        ctx.has_res = residual is not None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, weight = ctx.saved_tensors
        K, N = weight.shape
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = ops.linear_dgrad(dy2, ops.compute_weight_t(weight, x2.dtype))
        dw = ops.linear_wgrad(x2, dy2)                                      # [in,out] = x^T dy: the parameter's own layout
        db = ops.colsum(dy2)
        return dx.view(*dy.shape[:-1], K), dw, db, (dy if ctx.has_res else None)


class GPTMLPFn(torch.autograd.Function):
    """residual + Conv1D(gelu_new(Conv1D(x)))  (modeling_gpt.py:128-133 with the block's residual add, :142/:148)."""

    @staticmethod
    def forward(ctx, x: Tensor, w_fc: Tensor, b_fc: Tensor, w_proj: Tensor, b_proj: Tensor, residual: Tensor):
        H = w_fc.shape[0]
        x2 = x.reshape(-1, H)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = None
        if residual is not None:                                            # None: the caller adds it after dropout (mlp[3].p > 0)
            r2 = residual.reshape(-1, H)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        ctx.has_res = residual is not None
        u = torch.empty((x2.shape[0], w_fc.shape[1]), dtype=x.dtype, device=x.device)
        g = ops.linear_fwd(x2, ops.compute_weight_t(w_fc, x.dtype), b_fc.detach(), epilogue=_lib.EPI_GELU, aux_out=u)
        y = ops.linear_fwd(g, ops.compute_weight_t(w_proj, x.dtype), b_proj.detach(), residual=r2)
        ctx.save_for_backward(x2, u, g, w_fc, w_proj)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, u, g, w_fc, w_proj = ctx.saved_tensors
        H = w_fc.shape[0]
        dy2 = dy.reshape(-1, H)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dw_proj = ops.linear_wgrad(g, dy2)
        db_proj = ops.colsum(dy2)
        du = ops.linear_dgrad(dy2, ops.compute_weight_t(w_proj, x2.dtype), epilogue=_lib.EPI_DGELU, aux_in=u)
        dw_fc = ops.linear_wgrad(x2, du)
        db_fc = ops.colsum(du)
        dx = ops.linear_dgrad(du, ops.compute_weight_t(w_fc, x2.dtype))
        return dx.view(dy.shape), dw_fc, db_fc, dw_proj, db_proj, (dy if ctx.has_res else None)


class GPTAttnFn(torch.autograd.Function):
    """softmax(causal(q k^T / sqrt(hd)) + key mask) v on the fused [B,S,3H] = q | k | v activation (modeling_gpt.py:69-101)."""

    @staticmethod
    def forward(ctx, qkv: Tensor, mask: ops.MaskInfo, nh: int, scale: float, drop_p: float = 0.0, drop_seed: int = 0):
        B, S, H3 = qkv.shape
        H = H3 // 3
        hd = H // nh
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        q2 = qkv.view(B * S, H3)
        st = (S * H3, hd, H3)
        desc = ops._strided_desc(B, nh, S, S, hd, st, st, st, (S * H, hd, H), scale, S > 1, future_fill=-1e4, dropout_p=drop_p,
                                 dropout_seed=drop_seed)
        out = torch.empty((B, S, H), dtype=qkv.dtype, device=qkv.device)
        stat_m, stat_l = ops.attn_fwd(q2, q2[:, H:], q2[:, 2 * H:], out, desc, None, mask)
        ctx.save_for_backward(q2, out, stat_m, stat_l)
        ctx.desc, ctx.mask, ctx.H = desc, mask, H
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        q2, out, stat_m, stat_l = ctx.saved_tensors
        H = ctx.H
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dqkv = torch.empty_like(q2)
        ops.attn_bwd(q2, q2[:, H:], q2[:, 2 * H:], out, dout, stat_m, stat_l, dqkv, dqkv[:, H:], dqkv[:, 2 * H:],
                     ctx.desc, None, ctx.mask)
        return dqkv.view(out.shape[0], out.shape[1], 3 * H), None, None, None, None, None


class GPT2BlockFn(torch.autograd.Function):
    """One pre-LN GPT-2/3 block (modeling_gpt.py:144-149 with AttentionLayer :52-101 and the MLP :122-133) as ONE autograd node and ONE
    library call per direction — the block-level entry point of the C ABI (ctmi_bloom_block_fwd/bwd) with the GPT-2 flags: q | k | v
    blocked QKV activation, -1e4 future fill, optional score scaling, weights read AND weight gradients written in Conv1D's own
    [in,out] layout (the bf16 shadows the optimizer maintains are used as they are: no transposed compute copies on this path).  Used
    for training steps without dropout; everything else (GPT-1's post-LN order, dropout, KV-cache decode) keeps the per-op path."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, wa, ba, wp, bp, n2w, n2b, wf, bf, wo, bo, mask: ops.MaskInfo, nh: int, eps: float, scale: float, kv_out: list):
        B, S, H = x.shape
        cd = x.dtype
        x2 = x.reshape(B * S, H)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        params = (n1w.detach(), n1b.detach(), ops.compute_weight(wa, cd), ba.detach(), ops.compute_weight(wp, cd), bp.detach(),
                  n2w.detach(), n2b.detach(), ops.compute_weight(wf, cd), bf.detach(), ops.compute_weight(wo, cd), bo.detach())
        acts = ops.bloom_block_fwd(x2, params, mask, None, eps, False, B, S, nh, flags=_lib.BLK_QKV_BLOCKED | _lib.BLK_WGRAD_IN_OUT | _lib.BLK_W_IN_OUT,
                                   attn_scale=scale, future_fill=-1e4)
        ops.note_block_params(mask, (n1w, n1b, wa, ba, wp, bp, n2w, n2b, wf, bf, wo, bo))
        grad = getattr(kv_out, "grad", True) and any(ctx.needs_input_grad)                                # see BloomBlockFn: slab through save_for_backward, presents lazily
        if grad:
            ctx.save_for_backward(x2, n1w, n1b, wa, ba, wp, bp, n2w, n2b, wf, bf, wo, bo, acts.slab)
            ctx.geo, ctx.mask, ctx.eps, ctx.shape = acts.geometry(), mask, eps, (B, S, H)
        kv = ops.LazyKV(acts, blocked=True, eager=not grad)
        if grad:
            ctx.kv = kv
        kv_out.append(kv)
        return acts.out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 18
        x2, n1w, n1b, wa, ba, wp, bp, n2w, n2b, wf, bf, wo, bo, slab = ctx.saved_tensors
        B, S, H = ctx.shape
        cd = x2.dtype
        dout2 = dout.reshape(B * S, H)
        dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
        params = (n1w.detach(), n1b.detach(), ops.compute_weight(wa, cd), ba.detach(), ops.compute_weight(wp, cd), bp.detach(),
                  n2w.detach(), n2b.detach(), ops.compute_weight(wf, cd), bf.detach(), ops.compute_weight(wo, cd), bo.detach())
        ctx.kv.release()
        defer = x2.is_cuda and ops.params_allow_deferred_grads((n1w, n1b, wa, ba, wp, bp, n2w, n2b, wf, bf, wo, bo), ctx.mask)
        dx, g = ops.bloom_block_bwd(ops.BlockActs.rebuild(slab, ctx.geo), x2, params, ctx.mask, None, ctx.eps, False, dout2, defer_join=defer)
        return (dx.view(B, S, H), *g, None, None, None, None, None)


def _attend_cached(qkv: Tensor, past, mask: ops.MaskInfo, nh: int, scale: float, drop_p: float = 0.0, drop_seed: int = 0):
    """Inference with a KV cache (modeling_gpt.py:75-80): returns the context and the concatenated (k, v) [B,nh,Sk,hd]."""
    B, S, H3 = qkv.shape
    H = H3 // 3
    hd = H // nh
    qv = qkv.view(B, S, 3, nh, hd)
    k_new, v_new = qv[:, :, 1].transpose(1, 2), qv[:, :, 2].transpose(1, 2)
    k = torch.cat((past[0], k_new), dim=-2).contiguous() if past is not None else k_new.contiguous()
    v = torch.cat((past[1], v_new), dim=-2).contiguous() if past is not None else v_new.contiguous()
    Sk = k.shape[-2]
    q2 = qkv.reshape(B * S, H3)
    cs = (nh * Sk * hd, Sk * hd, hd)
    desc = ops._strided_desc(B, nh, S, Sk, hd, (S * H3, hd, H3), cs, cs, (S * H, hd, H), scale, S > 1, future_fill=-1e4,
                             dropout_p=drop_p, dropout_seed=drop_seed)
    out = torch.empty((B, S, H), dtype=qkv.dtype, device=qkv.device)
    ops.attn_fwd(q2, k, v, out, desc, None, mask)
    return out, (k, v)


# ------------------------------------------------------------------------------------------------ modules
class Conv1D(torch.nn.Module):
    """modeling_gpt.py:32-46: a Linear whose weight is stored [in, out]."""

    def __init__(self, out_dim, input_dim):
        super(Conv1D, self).__init__()
        w = torch.empty(input_dim, out_dim)
        torch.nn.init.normal_(w, std=0.02)
        self.weight = torch.nn.Parameter(w)
        self.bias = torch.nn.Parameter(torch.zeros(out_dim))

    def forward(self, x, residual=None):
        return Conv1DFn.apply(x, self.weight, self.bias, residual)


class AttentionLayer(torch.nn.Module):
    def __init__(self, config, scale=False):
        super().__init__()
        self.config, self.scale = config, scale
        self.n_state, self.n_head, self.n_ctx = config.n_embd, config.n_head, config.n_ctx
        assert self.n_state % self.n_head == 0
        # kept for state_dict parity with the reference (modeling_gpt.py:56); the kernels never read it
        self.register_buffer("bias", torch.tril(torch.ones(self.n_ctx, self.n_ctx)).view(1, 1, self.n_ctx, self.n_ctx))
        self.c_attn = Conv1D(self.n_state * 3, self.n_state)
        self.c_proj = Conv1D(self.n_state, self.n_state)
        self.attn_dropout = torch.nn.Dropout(config.attn_pdrop)
        self.resid_dropout = torch.nn.Dropout(config.resid_pdrop)

    def forward(self, hidden_states, k_v_past=None, attention_mask=None, head_mask=None, residual=None):
        """`attention_mask` is the per-forward ops.MaskInfo built by GPTModel.  Returns c_proj(context) (+ residual)."""
        if head_mask is not None:
            raise NotImplementedError("head_mask is not supported (the reference's `if head_mask:` is only safe for None)")
        # dropout (modeling_gpt.py:96 on the softmax output, :100 on the projected context): the kernels' counter-based masks
        pa = float(self.attn_dropout.p) if self.training else 0.0
        pr = float(self.resid_dropout.p) if self.training else 0.0
        sa = rng.next_seed() if pa > 0.0 else 0
        hd = self.n_state // self.n_head
        scale = 1.0 / math.sqrt(hd) if self.scale else 1.0
        qkv = self.c_attn(hidden_states)
        B, S, _ = qkv.shape
        if k_v_past is not None or not (torch.is_grad_enabled() and qkv.requires_grad):
            ctxv, kv = _attend_cached(qkv, k_v_past, attention_mask, self.n_head, scale, pa, sa)
        else:
            ctxv = GPTAttnFn.apply(qkv, attention_mask, self.n_head, scale, pa, sa)
            qv = qkv.view(B, S, 3, self.n_head, hd)
            kv = (qv[:, :, 1].transpose(1, 2), qv[:, :, 2].transpose(1, 2))           # views, like the reference's k_v_past
        if pr > 0.0:                                                    # residual + dropout(c_proj(a)): the add moves out of the GEMM epilogue
            return ops.DropoutFn.apply(self.c_proj(ctxv), pr, rng.next_seed(), residual), kv
        return self.c_proj(ctxv, residual=residual), kv


class NewGELUActivation(torch.nn.Module):
    """modeling_gpt.py:113-119; kept for module-tree parity (the GELU runs inside the c_fc GEMM epilogue)."""

    def forward(self, input):
        raise RuntimeError("NewGELUActivation is fused into the MLP GEMM epilogue and is never called on its own")


ACT2FN = {'gelu_new': NewGELUActivation}


class TransformerBlock(torch.nn.Module):
    def __init__(self, config, scale=False, version='gpt'):
        super(TransformerBlock, self).__init__()
        n_embd = config.n_embd
        self.version = version
        if config.afn != 'gelu_new':
            raise NotImplementedError("only afn='gelu_new' (the GPT / GPT-2 activation) is built")
        self.attn = AttentionLayer(config, scale)
        self.norm1 = LayerNorm(n_embd, eps=config.layer_norm_epsilon)
        self.mlp = torch.nn.Sequential(
            Conv1D(4 * n_embd, n_embd),
            ACT2FN[config.afn](),
            Conv1D(n_embd, 4 * n_embd),
            torch.nn.Dropout()
        )
        self.norm2 = LayerNorm(n_embd, eps=config.layer_norm_epsilon)

    def _mlp(self, x, residual):
        fc, proj = self.mlp[0], self.mlp[2]
        pm = float(self.mlp[3].p) if self.training else 0.0             # the trailing torch.nn.Dropout() — p = 0.5 in the reference
        if pm > 0.0:
            return ops.DropoutFn.apply(GPTMLPFn.apply(x, fc.weight, fc.bias, proj.weight, proj.bias, None), pm, rng.next_seed(), residual)
        return GPTMLPFn.apply(x, fc.weight, fc.bias, proj.weight, proj.bias, residual)

    def forward(self, x, attn_output=None, attention_mask=None, head_mask=None, k_v_past=None):
        if attn_output is not None:
            raise NotImplementedError("precomputed attn_output is not supported")
        if self.version == 'gpt':                                                    # GPT-1: post-LN (:138-143)
            s1, k_v_past = self.attn(x, attention_mask=attention_mask, head_mask=head_mask, k_v_past=k_v_past, residual=x)
            n1 = self.norm1(s1)
            output = self.norm2(self._mlp(n1, residual=n1))
        elif (k_v_past is None and torch.is_grad_enabled() and x.requires_grad
              and not (self.training and (self.attn.attn_dropout.p > 0.0 or self.attn.resid_dropout.p > 0.0 or self.mlp[3].p > 0.0))):
            # GPT-2/3 pre-LN training step without dropout: one autograd node, one library call per direction
            a, fc, proj = self.attn, self.mlp[0], self.mlp[2]
            hd = a.n_state // a.n_head
            kv = ops.KVOut()
            output = GPT2BlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.c_attn.weight, a.c_attn.bias, a.c_proj.weight, a.c_proj.bias,
                                       self.norm2.weight, self.norm2.bias, fc.weight, fc.bias, proj.weight, proj.bias,
                                       attention_mask, a.n_head, self.norm1.eps, (1.0 / math.sqrt(hd)) if a.scale else 1.0, kv)
            k_v_past = kv[0]
        else:                                                                        # GPT-2/3: pre-LN (:144-149), per-op
            x, k_v_past = self.attn(self.norm1(x), attention_mask=attention_mask, head_mask=head_mask, k_v_past=k_v_past, residual=x)
            output = self._mlp(self.norm2(x), residual=x)
        return output, k_v_past


class GPTModel(torch.nn.Module):
    def __init__(self, config, version='gpt'):
        super(GPTModel, self).__init__()
        self.version = version
        self.config = config
        self.tokens_embed = torch.nn.Embedding(config.vocab_size, config.n_embd)
        self.position_embed = torch.nn.Embedding(config.n_positions, config.n_embd)
        self.drop = torch.nn.Dropout(config.embd_pdrop)
        self.blocks = torch.nn.ModuleList([TransformerBlock(config, scale=True, version=version) for _ in range(config.n_layer)])
        if version != 'gpt':
            self.ln_f = LayerNorm(config.n_embd, eps=config.layer_norm_epsilon)
        self._tie: Optional[_TieCtx] = None

    def forward(self, input_ids, attention_mask=None, position_ids=None, segment_ids=None, k_v_pasts=None):
        if k_v_pasts is None:
            k_v_pasts = [None] * len(self.blocks)
        else:
            k_v_pasts = list(k_v_pasts)
        S = input_ids.shape[1]
        if attention_mask is None:
            if position_ids is None:
                raise TypeError("attention_mask or position_ids is required (the reference derives positions from the mask)")
            past_len = 0 if k_v_pasts[0] is None else k_v_pasts[0][0].shape[2]
            attention_mask = torch.ones((input_ids.shape[0], S + past_len), dtype=torch.long, device=input_ids.device)
        if position_ids is None:                                                     # modeling_gpt.py:166-169
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            position_ids = position_ids[:, -S:]
        cd = ops.effective_compute_dtype(_torch_dtype(getattr(self.config, "compute_dtype", "fp32")))
        minfo = ops.MaskInfo(attention_mask)                                         # the additive (1-mask)*finfo.min of :171-175
        tok = EmbedFn.apply(input_ids, self.tokens_embed.weight, cd, self._tie)
        pos = EmbedFn.apply(position_ids.contiguous(), self.position_embed.weight, cd, None)
        hidden_states = tok + pos
        if segment_ids is not None:
            hidden_states = hidden_states + EmbedFn.apply(segment_ids.view(-1, segment_ids.size(-1)), self.tokens_embed.weight, cd, None)
        if self.training and self.drop.p > 0.0:                          # embd_pdrop (modeling_gpt.py:190)
            hidden_states = ops.DropoutFn.apply(hidden_states, float(self.drop.p), rng.next_seed(), None)
        for i, block in enumerate(self.blocks):
            hidden_states, k_v_pasts[i] = block(hidden_states, attention_mask=minfo, k_v_past=k_v_pasts[i])
        if self.version == 'gpt':
            return hidden_states, k_v_pasts
        return self.ln_f(hidden_states), k_v_pasts


class GPTLMHeadModel(torch.nn.Module, GenerationMixin):
    def __init__(self, config, version='gpt'):
        super(GPTLMHeadModel, self).__init__()
        self.config = config
        self.version = version
        self.gpt = GPTModel(config, version=version)
        self.lm_head = torch.nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self._tie_weights()

    def _tie_weights(self):
        self.lm_head.weight = self.gpt.tokens_embed.weight

    def ct_tied_weight(self):
        w = self.gpt.tokens_embed.weight
        return w if self.lm_head.weight is w else None

    def set_compute_dtype(self, dtype):
        self.config.compute_dtype = dtype
        return self

    def forward(self, input_ids, attention_mask=None, segment_ids=None, position_ids=None, k_v_pasts=None, labels=None):
        tied = self.lm_head.weight is self.gpt.tokens_embed.weight
        tie = _TieCtx() if (tied and torch.is_grad_enabled() and self.lm_head.weight.requires_grad and segment_ids is None) else None
        self.gpt._tie = tie
        try:
            hidden_states, k_v_pasts = self.gpt(input_ids, attention_mask, position_ids, segment_ids, k_v_pasts)
        finally:
            self.gpt._tie = None
        lm_logits = LMHeadFn.apply(hidden_states, self.lm_head.weight, tie)
        outputs = (lm_logits, hidden_states)
        if labels is not None:                                                       # extension: the loss a training step needs
            outputs = (ShiftedCrossEntropyFn.apply(lm_logits, labels),) + outputs
        return outputs, k_v_pasts
