"""MI355X-native counterpart of CleanTransformer/models/modeling_bloom.py.

Same public surface as the reference (class names, constructor signatures, attribute and state_dict key names,
forward signatures and the nested ``((loss, logits, hidden), k_v_pasts)`` return shape — modeling_bloom.py:17-232),
so callers shaped like examples/ft_bloom.py / ft_bloom_DDP.py run unchanged.  Underneath, every block is ONE autograd
node whose forward/backward is a fixed sequence of ctmi355 HIP kernels:

    LayerNorm -> fused-QKV GEMM(+bias) -> flash attention (ALiBi + causal + key mask folded in, no [S,S] tensor)
    -> dense GEMM(+bias +residual) -> LayerNorm -> h->4h GEMM(+bias, tanh-GELU, keeps pre-activation)
    -> 4h->h GEMM(+bias +residual)

Parameters stay fp32 (master weights).  ``config.compute_dtype`` ("fp32" | "bf16") selects the storage type of
activations and of the weight matrices fed to the matrix cores (bf16 shadows are refreshed by the fused optimizer or
lazily when a parameter's version counter moves).  Parameter gradients are always produced in fp32.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import _lib, ops
from ..generation.generation_util import GenerationMixin
from ..transformer import LayerNorm

Tensor = torch.Tensor


class BloomConfig():
    """modeling_bloom.py:17-54 (same arguments, same ``n_embed`` synonym) plus ``compute_dtype``."""

    def __init__(
            self,
            vocab_size=250880,
            hidden_size=64,
            n_layer=2,
            num_attention_heads=8,
            layer_norm_epsilon=1e-5,
            initializer_range=0.02,
            use_cache=True,
            bos_token_id=1,
            eos_token_id=2,
            apply_residual_connection_post_layernorm=False,
            hidden_dropout=0.0,
            attention_dropout=0.0,
            pretraining_tp=1,
            slow_but_exact=False,
            compute_dtype="fp32",
            **kwargs,
    ):
        self.vocab_size = vocab_size
        n_embed = kwargs.pop("n_embed", None)
        self.hidden_size = hidden_size if n_embed is None else n_embed
        self.n_layer = n_layer
        self.n_head = self.num_attention_heads = num_attention_heads
        self.layer_norm_epsilon = layer_norm_epsilon
        self.initializer_range = initializer_range
        self.use_cache = use_cache
        self.pretraining_tp = pretraining_tp
        self.apply_residual_connection_post_layernorm = apply_residual_connection_post_layernorm
        self.hidden_dropout = hidden_dropout
        self.attention_dropout = attention_dropout
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.slow_but_exact = slow_but_exact
        self.num_hidden_layers = self.n_layer
        self.compute_dtype = compute_dtype


def _torch_dtype(name) -> torch.dtype:
    if isinstance(name, torch.dtype):
        return name
    return {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
            "fp16": torch.float16, "float16": torch.float16, "half": torch.float16}[str(name)]


def alibi_slopes(num_heads: int) -> Tensor:
    """Per-head ALiBi slopes, computed in fp32 exactly as modeling_bloom.py:312-325 does (host side, nh values)."""
    p2 = 2 ** math.floor(math.log2(num_heads))
    base = torch.tensor(2 ** (-(2 ** -(math.log2(p2) - 3))), dtype=torch.float32)
    slopes = torch.pow(base, torch.arange(1, 1 + p2, dtype=torch.int32))
    if p2 != num_heads:
        extra_base = torch.tensor(2 ** (-(2 ** -(math.log2(2 * p2) - 3))), dtype=torch.float32)
        n_extra = min(p2, num_heads - p2)
        slopes = torch.cat([slopes, torch.pow(extra_base, torch.arange(1, 1 + 2 * n_extra, 2, dtype=torch.int32))], dim=0)
    return slopes


import os as _os

_WGRAD_SIDE_STREAM = None       # None = ops.bloom_block_bwd decides (side stream unless the grouped weight-gradient launch applies); True / False force it (bench.py's breakdown pass)
_CHECK_IDS = _os.environ.get("CTMI_CHECK_IDS", "0") == "1"


class _AttnCtx:
    """Per-forward digest shared by all blocks: mask info on device, slopes, geometry."""

    def __init__(self, attention_mask: Tensor, nh: int, slopes: Tensor):
        self.mask = ops.MaskInfo(attention_mask)
        self.slopes = slopes
        self.nh = nh


# ------------------------------------------------------------------------------------------------ one block = one node
class BloomBlockFn(torch.autograd.Function):
    """modeling_bloom.py:142-159 (+ 76-124, 255-271) forward and its full backward: ONE autograd node and ONE library call
    per direction (ctmi_bloom_block_fwd / ctmi_bloom_block_bwd run the fixed kernel sequences; include/ctmi355.h)."""

    @staticmethod
    def forward(ctx, x, ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2, actx: _AttnCtx, eps: float,
                post_ln_res: bool, kv_out: list):
        B, S, H = x.shape
        cd = x.dtype
        x2 = x.reshape(B * S, H)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        params = (ln1_w.detach(), ln1_b.detach(), ops.compute_weight(wqkv, cd), bqkv.detach(), ops.compute_weight(wd, cd), bd.detach(),
                  ln2_w.detach(), ln2_b.detach(), ops.compute_weight(w1, cd), b1.detach(), ops.compute_weight(w2, cd), b2.detach())
        acts = ops.bloom_block_fwd(x2, params, actx.mask, actx.slopes, eps, post_ln_res, B, S, actx.nh)
        ops.note_block_params(actx.mask, (ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2))
        # The slab is a tensor: it goes through save_for_backward (released right after this node's backward — as a Python attribute of
        # ctx it lived as long as anything referenced the graph, i.e. through the NEXT step's forward in the reference loop); only
        # geometry stays on ctx.  Without a graph nothing is saved and the K/V presents are copied out (ops.LazyKV).
        grad = getattr(kv_out, "grad", True) and any(ctx.needs_input_grad)
        if grad:
            ctx.save_for_backward(x2, ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2, acts.slab)
            ctx.geo, ctx.actx, ctx.eps, ctx.post_ln_res, ctx.shape = acts.geometry(), actx, eps, post_ln_res, (B, S, H)
        kv = ops.LazyKV(acts, blocked=False, eager=not grad)
        if grad:
            ctx.kv = kv
        kv_out.append(kv)
        return acts.out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 17
        x2, ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2, slab = ctx.saved_tensors
        B, S, H = ctx.shape
        cd = x2.dtype
        dout2 = dout.reshape(B * S, H)
        dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
        params = (ln1_w.detach(), ln1_b.detach(), ops.compute_weight(wqkv, cd), bqkv.detach(), ops.compute_weight(wd, cd), bd.detach(),
                  ln2_w.detach(), ln2_b.detach(), ops.compute_weight(w1, cd), b1.detach(), ops.compute_weight(w2, cd), b2.detach())
        # Parameter gradients (wgrad GEMMs + bias column sums) feed nothing further down the backward chain: inside the
        # library call they run on a side HIP stream, concurrently with the dgrad GEMMs / attention backward, and are joined
        # back into the current stream before the call returns, so everything downstream (autograd accumulation, DDP hooks,
        # optimizer) is ordered.
        ctx.kv.release()
        side = _WGRAD_SIDE_STREAM
        # single process, no accumulation pending, no gradient hooks: a side stream (if one is used at all) is joined once, at the end of the backward pass
        defer = side is not False and x2.is_cuda and ops.params_allow_deferred_grads((ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2), ctx.actx.mask)
        dx, g = ops.bloom_block_bwd(ops.BlockActs.rebuild(slab, ctx.geo), x2, params, ctx.actx.mask, ctx.actx.slopes, ctx.eps, ctx.post_ln_res, dout2,
                                    use_side_stream=side, defer_join=defer)
        return (dx.view(B, S, H), *g, None, None, None, None)


class BloomBlockDropoutFn(torch.autograd.Function):
    """The same block with dropout (modeling_bloom.py:111 attention_dropout on the softmax output, :122 and :270 hidden_dropout on
    the two projections before their residual adds), training mode, p > 0.  Per-op launches from Python: no Bloom checkpoint of
    the SFT path uses dropout (bloom-560m / 7b1: 0.0), so this is the complete-but-unmeasured variant; the measured path is the
    one-call BloomBlockFn above.  The masks are the kernels' counter-based ones (ops.dropout, ctmi_attn_desc.dropout_*): three
    seeds per block and forward, nothing but the seeds is saved for the backward."""

    @staticmethod
    def forward(ctx, x, ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2, actx: _AttnCtx, eps: float,
                post_ln_res: bool, p_hidden: float, p_attn: float, seeds, kv_out: list):
        B, S, H = x.shape
        T = B * S
        nh = actx.nh
        hd = H // nh
        cd = x.dtype
        x2 = x.reshape(T, H)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        wqkv_c, wd_c = ops.compute_weight(wqkv, cd), ops.compute_weight(wd, cd)
        w1_c, w2_c = ops.compute_weight(w1, cd), ops.compute_weight(w2, cd)
        s_attn, s_h1, s_h2 = seeds
        ln1, mean1, rstd1 = ops.layernorm_fwd(x2, ln1_w.detach(), ln1_b.detach(), eps)
        qkv = ops.linear_fwd(ln1, wqkv_c, bqkv.detach())
        desc = ops.fused_qkv_desc(B, S, nh, hd, causal=S > 1, dropout_p=p_attn, dropout_seed=s_attn)
        att = torch.empty((T, H), dtype=cd, device=x.device)
        stat_m, stat_l = ops.attn_fwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], att, desc, actx.slopes, actx.mask)
        res1 = ln1 if post_ln_res else x2
        if p_hidden > 0.0:
            h1 = ops.dropout(ops.linear_fwd(att, wd_c, bd.detach()), p_hidden, s_h1, residual=res1)
        else:
            h1 = ops.linear_fwd(att, wd_c, bd.detach(), residual=res1)
        ln2, mean2, rstd2 = ops.layernorm_fwd(h1, ln2_w.detach(), ln2_b.detach(), eps)
        u = torch.empty((T, 4 * H), dtype=cd, device=x.device)
        g = ops.linear_fwd(ln2, w1_c, b1.detach(), epilogue=_lib.EPI_GELU, aux_out=u)
        res2 = ln2 if post_ln_res else h1
        if p_hidden > 0.0:
            out = ops.dropout(ops.linear_fwd(g, w2_c, b2.detach()), p_hidden, s_h2, residual=res2)
        else:
            out = ops.linear_fwd(g, w2_c, b2.detach(), residual=res2)
        ctx.save_for_backward(x2, ln1_w, wqkv, wd, ln2_w, w1, w2, mean1, rstd1, ln1, qkv, att, stat_m, stat_l, h1, mean2, rstd2, ln2, u, g)
        ctx.actx, ctx.desc, ctx.post_ln_res, ctx.shape = actx, desc, post_ln_res, (B, S, H)
        ctx.p_hidden, ctx.seeds = p_hidden, seeds
        qv = qkv.view(B, S, nh, 3, hd)
        kv_out.append((qv[:, :, :, 1, :].transpose(1, 2), qv[:, :, :, 2, :].transpose(1, 2)))
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        if dout is None:
            return (None,) * 20
        (x2, ln1_w, wqkv, wd, ln2_w, w1, w2, mean1, rstd1, ln1, qkv, att, stat_m, stat_l, h1, mean2, rstd2, ln2, u, g) = ctx.saved_tensors
        B, S, H = ctx.shape
        T = B * S
        hd = H // ctx.actx.nh
        cd = x2.dtype
        post, ph = ctx.post_ln_res, ctx.p_hidden
        _, s_h1, s_h2 = ctx.seeds
        wqkv_c, wd_c = ops.compute_weight(wqkv, cd), ops.compute_weight(wd, cd)
        w1_c, w2_c = ops.compute_weight(w1, cd), ops.compute_weight(w2, cd)
        dout2 = dout.reshape(T, H)
        dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
        # MLP: out = res2 + drop(W2 gelu(W1 ln2 + b1) + b2)
        dm = ops.dropout(dout2, ph, s_h2) if ph > 0.0 else dout2              # gradient of the projection output: same mask as the forward
        dw2, db2 = ops.linear_wgrad(dm, g), ops.colsum(dm)
        du = ops.linear_dgrad(dm, w2_c, epilogue=_lib.EPI_DGELU, aux_in=u)
        dw1, db1 = ops.linear_wgrad(du, ln2), ops.colsum(du)
        dln2 = ops.linear_dgrad(du, w1_c, residual=dout2 if post else None)
        dh1, dln2_w, dln2_b = ops.layernorm_bwd(dln2, h1, ln2_w.detach(), mean2, rstd2, dres=None if post else dout2)
        # attention: h1 = res1 + drop(Wd att + bd)
        dd = ops.dropout(dh1, ph, s_h1) if ph > 0.0 else dh1
        dwd, dbd = ops.linear_wgrad(dd, att), ops.colsum(dd)
        datt = ops.linear_dgrad(dd, wd_c)
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv, qkv[:, hd:], qkv[:, 2 * hd:], att, datt, stat_m, stat_l, dqkv, dqkv[:, hd:], dqkv[:, 2 * hd:], ctx.desc,
                     ctx.actx.slopes, ctx.actx.mask)
        dwqkv, dbqkv = ops.linear_wgrad(dqkv, ln1), ops.colsum(dqkv)
        dln1 = ops.linear_dgrad(dqkv, wqkv_c, residual=dh1 if post else None)
        dx, dln1_w, dln1_b = ops.layernorm_bwd(dln1, x2, ln1_w.detach(), mean1, rstd1, dres=None if post else dh1)
        return (dx.view(B, S, H), dln1_w, dln1_b, dwqkv, dbqkv, dwd, dbd, dln2_w, dln2_b, dw1, db1, dw2, db2,
                None, None, None, None, None, None, None)


def _decode_block(blk: "BloomBlock", x: Tensor, actx: _AttnCtx, past, eps: float, post_ln_res: bool):
    """Inference-only block forward with a KV cache (modeling_bloom.py:88-92): same kernels, nothing saved."""
    B, S, H = x.shape
    T = B * S
    nh = actx.nh
    hd = H // nh
    cd = x.dtype
    sa, mlp = blk.self_attention, blk.mlp
    x2 = x.reshape(T, H)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    ln1, _, _ = ops.layernorm_fwd(x2, blk.input_layernorm.weight.detach(), blk.input_layernorm.bias.detach(), eps)
    qkv = ops.linear_fwd(ln1, ops.compute_weight(sa.query_key_value.weight, cd), sa.query_key_value.bias.detach())
    qv = qkv.view(B, S, nh, 3, hd)
    k_new, v_new = qv[:, :, :, 1, :].transpose(1, 2), qv[:, :, :, 2, :].transpose(1, 2)
    k = torch.cat((past[0], k_new), dim=-2).contiguous()                                   # [B,nh,Sk,hd]
    v = torch.cat((past[1], v_new), dim=-2).contiguous()
    Sk = k.shape[-2]
    qs = (S * 3 * H, 3 * hd, 3 * H)
    cs = (nh * Sk * hd, Sk * hd, hd)
    desc = ops._strided_desc(B, nh, S, Sk, hd, qs, cs, cs, (S * H, hd, H), 1.0 / math.sqrt(hd), causal=S > 1)
    att = torch.empty((T, H), dtype=cd, device=x.device)
    ops.attn_fwd(qkv, k, v, att, desc, actx.slopes, actx.mask)
    h1 = ops.linear_fwd(att, ops.compute_weight(sa.dense.weight, cd), sa.dense.bias.detach(), residual=ln1 if post_ln_res else x2)
    ln2, _, _ = ops.layernorm_fwd(h1, blk.post_attention_layernorm.weight.detach(), blk.post_attention_layernorm.bias.detach(), eps)
    u = torch.empty((T, 4 * H), dtype=cd, device=x.device)
    g = ops.linear_fwd(ln2, ops.compute_weight(mlp.dense_h_to_4h.weight, cd), mlp.dense_h_to_4h.bias.detach(),
                       epilogue=_lib.EPI_GELU, aux_out=u)
    out = ops.linear_fwd(g, ops.compute_weight(mlp.dense_4h_to_h.weight, cd), mlp.dense_4h_to_h.bias.detach(),
                         residual=ln2 if post_ln_res else h1)
    return out.view(B, S, H), (k, v)


class _TieCtx:
    """Carries the LM-head dW to the embedding backward when the two share one [V,H] parameter, so the tied
    gradient is produced once (GEMM writes it, the embedding scatter-adds into it) instead of two dense tensors."""

    def __init__(self):
        self.pending: Optional[Tensor] = None
        self.n_tokens = 0
        self.embed_wants = False
        self.early = None              # the data-parallel wrapper's tied-gradient reducer, once it took the LM-head part


class EmbedFn(torch.autograd.Function):
    """modeling_bloom.py:190 (word_embeddings): row gather; backward = scatter-add into the fp32 [V,H] grad."""

    @staticmethod
    def forward(ctx, ids: Tensor, weight: Tensor, cd: torch.dtype, tie: Optional[_TieCtx]):
        table = ops.compute_weight(weight, cd)
        if _CHECK_IDS:
            # torch.nn.Embedding faults on ids outside [0, V); the gather kernel clamps them to row 0 (and the scatter skips them)
            # rather than fault asynchronously.  CTMI_CHECK_IDS=1 validates on the host (one device synchronisation per forward).
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= weight.shape[0]:
                raise IndexError(f"token id out of range: min {lo}, max {hi}, vocabulary {weight.shape[0]}")
        out = ops.embed_fwd(table, ids, None)
        ctx.save_for_backward(ids)
        ctx.vh, ctx.tie = tuple(weight.shape), tie
        if tie is not None:
            tie.embed_wants = weight.requires_grad
            tie.n_tokens = ids.numel()        # the data-parallel row exchange agrees on max over ranks of this — at LM-head BACKWARD time
                                              # (trainer/ddp.py: only a backward that takes the early path issues the collective)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        (ids,) = ctx.saved_tensors
        tie = ctx.tie
        dout = dout if dout.is_contiguous() else dout.contiguous()
        if tie is not None and tie.pending is not None:
            dw, tie.pending = tie.pending, None
        else:
            dw = torch.zeros(ctx.vh, dtype=torch.float32, device=dout.device)
        if tie is not None and tie.early is not None:
            # data parallel: dw already holds the AVERAGED LM-head part (its all-reduce ran under the whole backward);
            # the embedding part is T rows per rank, exchanged as rows instead of as a second dense [V,H] reduction
            early, tie.early = tie.early, None
            early.finish(dw, dout.view(-1, ctx.vh[1]), ids)
        else:
            ops.embed_bwd(dout.view(-1, ctx.vh[1]), ids, dw)
        return None, dw, None, None


class LMHeadFn(torch.autograd.Function):
    """modeling_bloom.py:220 lm_head (no bias, weight usually tied to the embedding table)."""

    @staticmethod
    def forward(ctx, hidden: Tensor, weight: Tensor, tie: Optional[_TieCtx]):
        B, S, H = hidden.shape
        h2 = hidden.reshape(B * S, H)
        h2 = h2 if h2.is_contiguous() else h2.contiguous()
        V = weight.shape[0]
        wc = ops.compute_weight(weight, hidden.dtype)
        out = None
        if V % 8 != 0 and hidden.dtype != torch.float32:
            # odd vocabulary (GPT-2: 50257): logits live in a buffer whose row pitch is padded to a multiple of 32, so rows
            # stay 16-byte aligned for the loss kernels and the two LM-head backward GEMMs; the caller sees the [.., :V] view
            out = torch.empty((B * S, ops.pad_rows(V)), dtype=hidden.dtype, device=hidden.device)[:, :V]
        logits = ops.linear_fwd(h2, wc, tag="lm_head_fwd", out=out)
        ctx.save_for_backward(h2, weight)
        ctx.tie, ctx.shape = tie, (B, S, H)
        return logits.view(B, S, V)

    @staticmethod
    def backward(ctx, dlogits: Tensor):
        h2, weight = ctx.saved_tensors
        B, S, H = ctx.shape
        V = weight.shape[0]
        d2 = dlogits.reshape(B * S, V)
        wc = ops.compute_weight(weight, h2.dtype)
        tie = ctx.tie
        tied = tie is not None and tie.embed_wants
        sync = getattr(weight, "_ct_tied_sync", None) if tied else None       # set by trainer/ddp.py on the shared [V,H] parameter
        pre = sync.prescale(weight) if sync is not None else None             # 1/world when this step reduces the dense part early
        Vp = ops.ZERO_PADDED.pop(d2.data_ptr(), (None, None))[0] if d2.is_cuda else None
        wpad = getattr(weight, "_ct_shadow_pad", None)
        if Vp is not None and d2.stride() == (Vp, 1) and wpad is not None and wpad.shape[0] == Vp and wpad.dtype == h2.dtype:
            # zero-padded dlogits [T, Vp] against the zero-padded table copy [Vp, H]: both GEMMs see aligned, 32-divisible
            # extents (K = Vp for the dgrad, M = Vp for the wgrad) and the padding contributes exact zeros
            dp = d2.as_strided((B * S, Vp), (Vp, 1))
            dh = ops.linear_dgrad(dp, wpad)
            dw = ops.linear_wgrad(dp, h2, alpha=1.0 if pre is None else pre)[:V]
        else:
            d2 = d2 if d2.is_contiguous() else d2.contiguous()
            rows = sync.chunk_rows(V, H) if pre is not None else V
            if rows < V:
                # data parallel: the [V,H] weight gradient in row pieces, each handed to the all-reduce as soon as its GEMM is enqueued
                # (dW[c0:c1] = dlogits[:, c0:c1]^T h: the K-major A operand is a column window of dlogits, no copy)
                dw = torch.empty((V, H), dtype=torch.float32, device=d2.device)
                sync.announce(tie.n_tokens, d2.device)
                for c0 in range(0, V, rows):
                    c1 = min(V, c0 + rows)
                    ops.gemm(d2[:, c0:], V, True, h2, H, True, c1 - c0, H, B * S, out=dw[c0:c1], out_f32=True, alpha=pre)
                    sync.begin(dw[c0:c1])
                tie.pending, tie.early = dw, sync
                dh = ops.linear_dgrad(d2, wc)
                return dh.view(B, S, H), None, None
            dh = ops.linear_dgrad(d2, wc)
            dw = ops.linear_wgrad(d2, h2, alpha=1.0 if pre is None else pre)
        if tied:
            tie.pending = dw                                   # the embedding backward (always later) finishes and returns it
            if pre is not None:
                sync.announce(tie.n_tokens, dw.device)
                sync.begin(dw)
                tie.early = sync
            dw = None
        return dh.view(B, S, H), dw, None


_FUSED_CE = _os.environ.get("CTMI_FUSED_CE", "1") != "0"


class ShiftedCrossEntropyFn(torch.autograd.Function):
    """modeling_bloom.py:224-230: logits[..., :-1, :] vs labels[..., 1:], torch CE 'mean'; the shift is index
    arithmetic inside the kernel (no .contiguous() copy of the logits).  When a gradient will be wanted, the loss and
    dlogits come out of ONE pass over the logits (ctmi_ce_fwd_bwd): the backward is then only a conditional rescale."""

    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor):
        B, S, V = logits.shape
        l2 = logits.reshape(B * S, V)
        l2 = l2 if l2.stride(1) == 1 else l2.contiguous()                  # a padded row pitch is fine: the kernels take ld
        lab = labels.to(torch.int64)
        lab = lab if lab.is_contiguous() else lab.contiguous()
        ctx.shape = (B, S, V)
        ctx.fused = bool(_FUSED_CE and ctx.needs_input_grad[0] and ops.ce_fused_ok(l2))
        fac, fdev = ops.current_expected_loss_grad()
        if fdev is not None and fdev.device != l2.device:
            fdev = None
        if ctx.fused and l2.dtype == torch.float16 and fdev is None:
            # IEEE half cannot hold an UNSCALED dlogits: (p - y) / N at V = 250 880, T = 8192 is ~5e-10, below the smallest half subnormal,
            # and a scale that arrives only in backward would multiply zeros.  Without a registered device scale (a torch.amp.GradScaler,
            # or no scaler at all) the two-pass form is used: its backward receives the real upstream gradient in fp32 and applies it
            # BEFORE the half cast, as the reference's fp32 CE under autocast does (round-5 advisor).
            ctx.fused = False
        if ctx.fused:
            # the upstream gradient the loop announced (1 / accumulation steps, a GradScaler's device scale; 1 otherwise) rides in this pass
            if fdev is not None:
                # a private snapshot: the node compares the gradient that arrives in backward with what THIS forward folded, also when the
                # scaler's live scale has moved in between (update() / load_state_dict() between forward and backward; round-4 advisor)
                fdev = fdev.clone()
            loss_out, _, dl = ops.ce_fwd_bwd(l2, lab, seq=S, shift=1, ignore_index=-100, denom_mode=0, grad_factor=fac, grad_factor_dev=fdev)
            ctx.dl, ctx.used, ctx.applied = dl, False, (fac, fdev)
            return loss_out[0].clone()
        loss_out, row_lse = ops.ce_fwd(l2, lab, seq=S, shift=1, ignore_index=-100, denom_mode=0)
        ctx.save_for_backward(l2, lab, row_lse, loss_out)
        return loss_out[0].clone()

    @staticmethod
    def backward(ctx, gout: Tensor):
        B, S, V = ctx.shape
        g = gout.to(torch.float32).reshape(1)
        g = g if g.is_contiguous() else g.contiguous()
        if ctx.fused:
            if ctx.used:
                raise RuntimeError("the fused loss node keeps ONE gradient buffer and rescales it in place: it cannot be "
                                   "backpropagated twice (set CTMI_FUSED_CE=0 for retain_graph workflows)")
            ctx.used = True
            d = ops.scale_if_(ctx.dl, g, *ctx.applied)                    # no-op on the device when gout is what the forward was told to expect
            ctx.dl = None
            if d.stride(0) != V and d._base is not None:
                # padded row pitch (odd vocabulary): the pad columns were zeroed with the buffer — tell the LM-head backward, as the two-pass form does below
                ops.ZERO_PADDED.clear()                                   # at most one live entry (it pins its buffer)
                ops.ZERO_PADDED[d.data_ptr()] = (d.stride(0), d._base)
            return d.view(B, S, V), None
        l2, lab, row_lse, loss_out = ctx.saved_tensors
        out = None
        if l2.is_cuda and l2.stride(0) != V and l2.stride(0) == ops.pad_rows(V):
            # logits came with a padded row pitch (odd vocabulary): give dlogits the same pitch and zero its pad columns, and
            # tell the LM-head backward so (it then runs both of its GEMMs over the padded extent)
            Vp = l2.stride(0)
            buf = torch.empty((B * S, Vp), dtype=l2.dtype, device=l2.device)
            buf[:, V:].zero_()
            out = buf[:, :V]
            ops.ZERO_PADDED.clear()                                       # at most one live entry (it pins its buffer)
            ops.ZERO_PADDED[out.data_ptr()] = (Vp, buf)
        d = ops.ce_bwd(l2, lab, row_lse, loss_out, g, seq=S, shift=1, ignore_index=-100, out=out)
        return d.view(B, S, V), None


# ------------------------------------------------------------------------------------------------ modules
class BloomAttentionLayer(torch.nn.Module):
    """Parameter container with the reference's names (modeling_bloom.py:57-74); the math lives in BloomBlockFn."""

    def __init__(self, config):
        super().__init__()
        self.pretraining_tp = config.pretraining_tp
        self.slow_but_exact = config.slow_but_exact
        self.hidden_size = config.hidden_size
        self.num_heads = config.n_head
        self.head_dim = self.hidden_size // self.num_heads
        self.hidden_dropout = config.hidden_dropout
        self.inv_norm_factor = 1.0 / math.sqrt(self.head_dim)
        self.beta = 1.0
        self.query_key_value = torch.nn.Linear(self.hidden_size, 3 * self.hidden_size, bias=True)
        self.dense = torch.nn.Linear(self.hidden_size, self.hidden_size)
        self.attention_dropout = torch.nn.Dropout(config.attention_dropout)


class BloomGelu(torch.nn.Module):
    """modeling_bloom.py:289-305; kept for attribute parity (the GELU runs inside the h->4h GEMM epilogue)."""

    def __init__(self):
        super().__init__()


class BloomMLP(torch.nn.Module):
    """modeling_bloom.py:243-253 parameter container."""

    def __init__(self, config):
        super().__init__()
        hidden_size = config.hidden_size
        self.pretraining_tp = config.pretraining_tp
        self.slow_but_exact = config.slow_but_exact
        self.dense_h_to_4h = torch.nn.Linear(hidden_size, 4 * hidden_size)
        self.gelu_impl = BloomGelu()
        self.dense_4h_to_h = torch.nn.Linear(4 * hidden_size, hidden_size)
        self.hidden_dropout = config.hidden_dropout


class BloomBlock(torch.nn.Module):
    """modeling_bloom.py:127-159."""

    def __init__(self, config):
        super().__init__()
        hidden_size = config.hidden_size
        self.input_layernorm = LayerNorm(hidden_size, eps=config.layer_norm_epsilon)
        self.num_heads = config.n_head
        self.self_attention = BloomAttentionLayer(config)
        self.post_attention_layernorm = LayerNorm(hidden_size, eps=config.layer_norm_epsilon)
        self.mlp = BloomMLP(config)
        self.apply_residual_connection_post_layernorm = config.apply_residual_connection_post_layernorm
        self.hidden_dropout = config.hidden_dropout
        self.eps = config.layer_norm_epsilon

    def forward(self, hidden_states, attention_mask, alibi, head_mask, k_v_past=None):
        """`attention_mask` here is the per-forward _AttnCtx built by BloomModel (the reference passes the bool mask and
        the alibi tensor; both are folded into the attention kernel, so `alibi` is unused)."""
        if head_mask is not None:
            raise NotImplementedError("head_mask is not supported (SURVEY Q11: callers always pass None)")
        if self.self_attention.pretraining_tp > 1 and self.self_attention.slow_but_exact:
            raise Exception("pretraining_tp and slow_but_exact not supported yet")        # modeling_bloom.py:118-119
        actx: _AttnCtx = attention_mask
        sa, mlp = self.self_attention, self.mlp
        if k_v_past is not None:
            if torch.is_grad_enabled() and hidden_states.requires_grad:
                raise NotImplementedError("training through a KV cache is not supported")
            return _decode_block(self, hidden_states, actx, k_v_past, self.eps, self.apply_residual_connection_post_layernorm)
        p_hidden = float(self.hidden_dropout) if self.training else 0.0
        p_attn = float(self.self_attention.attention_dropout.p) if self.training else 0.0
        if p_hidden > 0.0 or p_attn > 0.0:
            from .. import rng
            kv = []
            out = BloomBlockDropoutFn.apply(
                hidden_states, self.input_layernorm.weight, self.input_layernorm.bias,
                sa.query_key_value.weight, sa.query_key_value.bias, sa.dense.weight, sa.dense.bias,
                self.post_attention_layernorm.weight, self.post_attention_layernorm.bias,
                mlp.dense_h_to_4h.weight, mlp.dense_h_to_4h.bias, mlp.dense_4h_to_h.weight, mlp.dense_4h_to_h.bias,
                actx, self.eps, self.apply_residual_connection_post_layernorm, p_hidden, p_attn,
                (rng.next_seed(), rng.next_seed(), rng.next_seed()), kv)
            return out, kv[0]
        kv = ops.KVOut()
        out = BloomBlockFn.apply(
            hidden_states, self.input_layernorm.weight, self.input_layernorm.bias,
            sa.query_key_value.weight, sa.query_key_value.bias, sa.dense.weight, sa.dense.bias,
            self.post_attention_layernorm.weight, self.post_attention_layernorm.bias,
            mlp.dense_h_to_4h.weight, mlp.dense_h_to_4h.bias, mlp.dense_4h_to_h.weight, mlp.dense_4h_to_h.bias,
            actx, self.eps, self.apply_residual_connection_post_layernorm, kv)
        return out, kv[0]


class BloomModel(torch.nn.Module):
    """modeling_bloom.py:162-205."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_heads = config.n_head
        self.embed_dim = config.hidden_size
        self.word_embeddings = torch.nn.Embedding(config.vocab_size, self.embed_dim)
        self.word_embeddings_layernorm = LayerNorm(self.embed_dim, eps=config.layer_norm_epsilon)
        self.blocks = torch.nn.ModuleList([BloomBlock(config) for _ in range(config.num_hidden_layers)])
        self.ln_f = LayerNorm(self.embed_dim, eps=config.layer_norm_epsilon)
        self._slopes = None
        self._tie: Optional[_TieCtx] = None

    def _alibi_slopes(self, device) -> Tensor:
        if self._slopes is None or self._slopes.device != device:
            self._slopes = alibi_slopes(self.num_heads).to(device)
        return self._slopes

    def forward(self, input_ids, attention_mask, head_mask, k_v_pasts=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is not supported (SURVEY Q11)")
        if k_v_pasts is None:
            k_v_pasts = [None] * self.config.n_layer
        if attention_mask is None:
            past_len = 0 if k_v_pasts[0] is None else k_v_pasts[0][0].shape[2]
            attention_mask = torch.ones((input_ids.shape[0], input_ids.shape[1] + past_len), dtype=torch.long, device=input_ids.device)
        cd = ops.effective_compute_dtype(_torch_dtype(getattr(self.config, "compute_dtype", "fp32")))
        emb = EmbedFn.apply(input_ids, self.word_embeddings.weight, cd, self._tie)
        hidden_states = self.word_embeddings_layernorm(emb)
        actx = _AttnCtx(attention_mask, self.num_heads, self._alibi_slopes(input_ids.device))
        for i, block in enumerate(self.blocks):
            hidden_states, k_v_pasts[i] = block(hidden_states, attention_mask=actx, alibi=None, head_mask=None,
                                                k_v_past=k_v_pasts[i])
        return self.ln_f(hidden_states), k_v_pasts


class BloomForCausalLM(torch.nn.Module, GenerationMixin):
    """modeling_bloom.py:208-232."""

    def __init__(self, config: BloomConfig):
        super().__init__()
        self.config = config
        self.bloom = BloomModel(config)
        self.lm_head = torch.nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def _tie_weight(self):
        self.lm_head.weight = self.bloom.word_embeddings.weight

    def ct_tied_weight(self):
        """The [V,H] parameter shared by the embedding and the LM head (None when untied): its gradient is the sum of a
        dense part produced FIRST in backward and T sparse rows produced LAST — trainer/ddp.py reduces them separately."""
        w = self.bloom.word_embeddings.weight
        return w if self.lm_head.weight is w else None

    def set_compute_dtype(self, dtype):
        """'fp32' (parity mode), 'bf16' (the measured MFMA path) or 'fp16' (functional path: the reference's autocast dtype); fp32 master weights /
        grads / optimizer state in every mode."""
        self.config.compute_dtype = dtype
        return self

    def forward(self, input_ids, attention_mask=None, head_mask=None, k_v_pasts=None, labels=None, **kwargs):
        tied = self.lm_head.weight is self.bloom.word_embeddings.weight
        tie = _TieCtx() if (tied and torch.is_grad_enabled() and self.lm_head.weight.requires_grad) else None
        self.bloom._tie = tie
        try:
            hidden_states, k_v_pasts = self.bloom(input_ids, attention_mask, head_mask, k_v_pasts)
        finally:
            self.bloom._tie = None
        lm_logits = LMHeadFn.apply(hidden_states, self.lm_head.weight, tie)
        outputs = (lm_logits, hidden_states)
        if labels is not None:
            loss = ShiftedCrossEntropyFn.apply(lm_logits, labels)
            outputs = (loss,) + outputs
        return outputs, k_v_pasts
