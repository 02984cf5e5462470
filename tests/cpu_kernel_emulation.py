"""TEST INFRASTRUCTURE: CPU emulation of the ctmi355 kernel *contracts* (include/ctmi355.h), written with plain torch
ops.  ``install()`` monkeypatches ``cleantransformer_amd.ops`` so that the product's host logic — autograd nodes,
the hand-derived block backward, the tied-weight gradient hand-off, optimizers, DDP — can be exercised in the build
container (no GPU) against the golden vectors.  It is never used by the product path or on the GPU box's -m gpu tests,
where the real HIP kernels run and are checked one by one against the oracle.
"""
from __future__ import annotations

import math

import torch

FMIN = torch.finfo(torch.float32).min


def _gelu(x):
    return x * 0.5 * (1.0 + torch.tanh(0.79788456 * x * (1 + 0.044715 * x * x)))


def _dgelu(x):
    t = torch.tanh(0.79788456 * x * (1 + 0.044715 * x * x))
    return 0.5 * x * ((1 - t * t) * (0.79788456 + 0.1070322243 * x * x)) + 0.5 * (1 + t)


def layernorm_fwd(x2d, w, b, eps):
    x = x2d.float()
    mean = x.mean(-1)
    var = ((x - mean[:, None]) ** 2).mean(-1)
    rstd = 1.0 / torch.sqrt(var + eps)
    y = w * ((x - mean[:, None]) * rstd[:, None]) + b
    return y.to(x2d.dtype), mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dres=None):
    xh = (x.float() - mean[:, None]) * rstd[:, None]
    g = dy.float() * w
    dx = rstd[:, None] * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dres is not None:
        dx = dx + dres.float()
    return dx.to(x.dtype), (dy.float() * xh).sum(0), dy.float().sum(0)


def gemm(A, lda, a_kmajor, B, ldb, b_kmajor, M, N, K, *, out=None, out_f32=False, bias=None, residual=None, epilogue=0,
         aux_in=None, aux_out=None, alpha=1.0, beta=0, tag=None):
    # pointer + leading-dimension semantics of the C ABI (A / B may be windows of a larger buffer: element 0 of the view is the origin)
    a = A.as_strided((K, M), (lda, 1)).t() if a_kmajor else A.as_strided((M, K), (lda, 1))
    b = B.as_strided((K, N), (ldb, 1)) if b_kmajor else B.as_strided((N, K), (ldb, 1)).t()
    v = alpha * (a.float() @ b.float())
    if bias is not None:
        v = v + bias
    cd = A.dtype
    if epilogue == 1:
        v = v.to(cd).float()
        aux_out.copy_(v.to(cd))
        v = _gelu(v)
    elif epilogue == 2:
        v = v * _dgelu(aux_in.float())
    elif epilogue == 5:                                    # GELUG: the forward leaves gelu'(x) for the backward
        v = v.to(cd).float()
        aux_out.copy_(_dgelu(v).to(cd))
        v = _gelu(v)
    elif epilogue == 6:                                    # MUL
        v = v * aux_in.float()
    elif epilogue == 3:
        v = torch.relu(v)
    elif epilogue == 4:
        v = torch.where(aux_in.float() > 0, v, torch.zeros(()))
    if residual is not None:
        v = v + residual.float()
    odt = torch.float32 if out_f32 else cd
    if out is None:
        out = torch.empty((M, N), dtype=odt)
        beta = 0
    if beta:
        v = v + out.float()
    out.copy_(v.to(out.dtype))
    return out


def colsum(x2d, out=None, accumulate=False):
    s = x2d.float().sum(0)
    if out is None:
        return s
    out.copy_(out + s if accumulate else s)
    return out


class MaskInfo:
    def __init__(self, attention_mask):
        am = attention_mask.to(torch.int64)
        self.B, self.S = am.shape
        self.kpos = ((am.cumsum(-1) - 1) * am).float()
        self.kvalid = (am != 0).to(torch.int32)
        fv = torch.full((self.B,), self.S, dtype=torch.int32)
        for b in range(self.B):
            nz = (am[b] != 0).nonzero()
            if len(nz):
                fv[b] = int(nz[0])
        self.first_valid = fv


def _strided(t, B, nh, S, hd, bs, hs, rs):
    return t.as_strided((B, nh, S, hd), (bs, hs, rs, 1), t.storage_offset())


def _scores(q, k, desc, slopes, mask, add_mask):
    B, nh, Sq, Sk = desc.B, desc.nh, desc.Sq, desc.Sk
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) * desc.scale
    if slopes is not None:
        s = s + slopes.view(1, nh, 1, 1) * mask.kpos.view(B, 1, 1, Sk)
    if add_mask is not None:
        s = s + add_mask.as_strided((B, nh, Sq, Sk), (desc.am_b, desc.am_h, desc.am_q, desc.am_k), add_mask.storage_offset())
    masked = torch.zeros(B, 1, Sq, Sk, dtype=torch.bool)
    ff = float(getattr(desc, "future_fill", 0.0)) or FMIN            # 0 = finfo.min (Bloom); GPT-2: -1e4 on future pairs
    if desc.causal:
        qi = torch.arange(Sq).view(Sq, 1) + (Sk - Sq)
        fut = (torch.arange(Sk).view(1, Sk) > qi).view(1, 1, Sq, Sk)
        masked = masked | fut
        s = torch.where(fut.expand(B, nh, Sq, Sk), torch.full((), ff), s)
    if mask is not None:
        pad = (mask.kvalid.view(B, 1, 1, Sk) == 0)
        masked = masked | pad
        s = torch.where(pad.expand(B, nh, Sq, Sk), torch.full((), FMIN), s)
    masked = masked.expand(B, nh, Sq, Sk)
    return s, masked


def _attn_keep(desc):
    """[B,nh,Sq,Sk] keep mask and scale of the attention-probability dropout, as the kernels define it (include/ctmi355.h)."""
    from cleantransformer_amd import rng
    p = float(getattr(desc, "dropout_p", 0.0))
    if p == 0.0:
        return None, 1.0
    B, nh, Sq, Sk = desc.B, desc.nh, desc.Sq, desc.Sk
    c = torch.arange(B * nh * Sq * Sk, dtype=torch.int64).view(B, nh, Sq, Sk)
    keep = rng.keep_hash(c, int(desc.dropout_seed)) >= rng.drop_threshold(p)
    return keep, 1.0 / (1.0 - p)


def dropout(x, p, seed, residual=None, out=None):
    from cleantransformer_amd import rng
    keep = rng.keep_hash(torch.arange(x.numel(), dtype=torch.int64), int(seed)) >= rng.drop_threshold(p)
    y = torch.where(keep.view(x.shape), x.float() * torch.tensor(1.0 / (1.0 - p), dtype=torch.float32), torch.zeros(()))
    if residual is not None:
        y = y + residual.float()
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def attn_fwd(q, k, v, out, desc, slopes, mask, add_mask=None):
    B, nh, Sq, Sk, hd = desc.B, desc.nh, desc.Sq, desc.Sk, desc.hd
    qv = _strided(q, B, nh, Sq, hd, desc.q_bs, desc.q_hs, desc.q_rs)
    kv = _strided(k, B, nh, Sk, hd, desc.k_bs, desc.k_hs, desc.k_rs)
    vv = _strided(v, B, nh, Sk, hd, desc.v_bs, desc.v_hs, desc.v_rs)
    s, _ = _scores(qv, kv, desc, slopes, mask, add_mask)
    m = s.max(-1).values
    p = torch.exp(s - m[..., None])
    l = p.sum(-1)
    pn = p / l[..., None]
    keep, ds = _attn_keep(desc)
    if keep is not None:
        pn = torch.where(keep, pn * ds, torch.zeros(()))
    o = torch.einsum("bhqk,bhkd->bhqd", pn, vv.float())
    _strided(out, B, nh, Sq, hd, desc.o_bs, desc.o_hs, desc.o_rs).copy_(o.to(out.dtype))
    return m, l


def attn_bwd(q, k, v, o, d_o, stat_m, stat_l, dq, dk, dv, desc, slopes, mask, add_mask=None):
    B, nh, Sq, Sk, hd = desc.B, desc.nh, desc.Sq, desc.Sk, desc.hd
    qv = _strided(q, B, nh, Sq, hd, desc.q_bs, desc.q_hs, desc.q_rs)
    kv = _strided(k, B, nh, Sk, hd, desc.k_bs, desc.k_hs, desc.k_rs)
    vv = _strided(v, B, nh, Sk, hd, desc.v_bs, desc.v_hs, desc.v_rs)
    ov = _strided(o, B, nh, Sq, hd, desc.o_bs, desc.o_hs, desc.o_rs).float()
    gv = _strided(d_o, B, nh, Sq, hd, desc.o_bs, desc.o_hs, desc.o_rs).float()
    s, masked = _scores(qv, kv, desc, slopes, mask, add_mask)
    p = torch.exp(s - stat_m[..., None]) / stat_l[..., None]
    delta = (ov * gv).sum(-1)
    dp = torch.einsum("bhqd,bhkd->bhqk", gv, vv.float())
    keep, dsc = _attn_keep(desc)
    pd = p
    if keep is not None:
        dp = torch.where(keep, dp * dsc, torch.zeros(()))
        pd = torch.where(keep, p * dsc, torch.zeros(()))
    ds = torch.where(masked, torch.zeros(()), p * (dp - delta[..., None]))
    _strided(dv, B, nh, Sk, hd, desc.v_bs, desc.v_hs, desc.v_rs).copy_(torch.einsum("bhqk,bhqd->bhkd", pd, gv).to(dv.dtype))
    _strided(dk, B, nh, Sk, hd, desc.k_bs, desc.k_hs, desc.k_rs).copy_((desc.scale * torch.einsum("bhqk,bhqd->bhkd", ds, qv.float())).to(dk.dtype))
    _strided(dq, B, nh, Sq, hd, desc.q_bs, desc.q_hs, desc.q_rs).copy_((desc.scale * torch.einsum("bhqk,bhkd->bhqd", ds, kv.float())).to(dq.dtype))


def embed_fwd(table, ids, err_flag=None):
    return table[ids]


def embed_bwd(dout, ids, dtable, scale=1.0):
    flat = ids.reshape(-1)
    ok = (flat >= 0) & (flat < dtable.shape[0])                      # the kernel skips ids outside [0, V) (padding rows of the DDP exchange)
    rows = dout.float().reshape(-1, dtable.shape[1])
    dtable.index_add_(0, flat[ok], rows[ok] * torch.tensor(scale, dtype=torch.float32))


def _targets(labels, N, seq, shift, ignore):
    lab = labels.reshape(-1)
    tgt = torch.full((N,), -1, dtype=torch.int64)
    for r in range(N):
        s = r % seq
        if s + shift < seq:
            t = int(lab[(r // seq) * seq + s + shift])
            tgt[r] = -1 if t == ignore else t
    return tgt


def ce_fwd(logits2d, labels, seq, shift, ignore_index=-100, denom_mode=0, denom_rows=0):
    N, C = logits2d.shape
    x = logits2d.float()
    lse = torch.logsumexp(x, -1)
    tgt = _targets(labels, N, seq, shift, ignore_index)
    live = tgt >= 0
    row = torch.where(live, lse - x.gather(1, tgt.clamp(min=0)[:, None])[:, 0], torch.zeros(()))
    denom = float(live.sum()) if denom_mode == 0 else (float(denom_rows) if denom_mode == 1 else 1.0)
    return torch.stack([row.double().sum().float() / denom, torch.tensor(1.0 / denom)]), lse


def ce_bwd(logits2d, labels, row_lse, loss_out, gout, seq, shift, ignore_index=-100, out=None):
    N, C = logits2d.shape
    tgt = _targets(labels, N, seq, shift, ignore_index)
    live = tgt >= 0
    p = torch.exp(logits2d.float() - row_lse[:, None])
    p[torch.arange(N)[live], tgt[live]] -= 1.0
    coef = (gout[0] if gout is not None else 1.0) * loss_out[1]
    d = torch.where(live[:, None], p * coef, torch.zeros(()))
    return d.to(logits2d.dtype)


def cast(src, dtype, out=None):
    r = src.to(dtype)
    if out is not None:
        out.copy_(r)
        return out
    return r


def transpose_cast(src, dtype, out=None):
    t = src.detach().t().contiguous().to(dtype)
    if out is None:
        return t
    out.copy_(t)
    return out


def sumsq(x, out=None, accumulate=False):
    s = x.double().pow(2).sum().reshape(1)
    if out is None:
        return s
    out.copy_(out + s if accumulate else s)
    return out


def scale_(x, s, s_dev=None):
    x.mul_(s * (float(s_dev[0]) if s_dev is not None else 1.0))
    return x


def scale_copy(src, dst, s):
    torch.mul(src, s, out=dst)
    return dst


def argmax_lastdim(x2d):
    return x2d.float().argmax(-1)


def ce_soft_fwd(logits2d, target, denom_mode, denom_rows):
    x = logits2d.float()
    lse = torch.logsumexp(x, dim=-1)
    tsum = target.sum(-1)
    row = lse * tsum - (target * x).sum(-1)
    denom = float(denom_rows) if denom_mode == 1 else 1.0
    return torch.stack([row.double().sum() / denom, torch.tensor(1.0 / denom, dtype=torch.float64)]).float(), lse, tsum


def ce_soft_bwd(logits2d, target, row_lse, row_tsum, loss_out, gout):
    coef = (gout[0] if gout is not None else 1.0) * loss_out[1]
    return ((torch.exp(logits2d.float() - row_lse[:, None]) * row_tsum[:, None] - target) * coef).to(logits2d.dtype)


def row_lse(x2d):
    x = x2d.float()
    m = x.max(dim=-1).values
    return torch.stack([m, (x - m[:, None]).exp().sum(-1).log()], dim=1)


def group_topk(x2d, group, k, stats=None, add=None, add_mul=1.0):
    """(value desc, flat index asc) selection, as ctmi_group_topk documents its tie order."""
    s = x2d.float()
    if stats is not None:
        s = (s - stats[:, 0:1]) - stats[:, 1:2]
    if add is not None:
        s = s + add.float().reshape(-1, 1) * add_mul
    s = s.reshape(x2d.shape[0] // group, -1)
    vals, idxs = [], []
    for row in s:
        order = sorted(range(row.numel()), key=lambda j: (-float(row[j]), j)) if row.numel() <= 4096 else None
        if order is None:                                  # large rows: torch.topk, then a stable re-sort of equal values
            v, i = row.topk(k)
            order = sorted(i.tolist(), key=lambda j: (-float(row[j]), j))
        order = order[:k]
        idxs.append(order)
        vals.append([float(row[j]) for j in order])
    return torch.tensor(vals, dtype=torch.float32), torch.tensor(idxs, dtype=torch.int64)


def scores_filter(x2d, divisor=1.0, thr=None, fill=float("-inf")):
    v = x2d / divisor if divisor != 1.0 else x2d.clone()
    if thr is not None:
        v = v.masked_fill(v < thr[:, None], fill)
    return v


def adamw_step(params, grads, exp_avg, exp_avg_sq, shadows, *, lr, beta1, beta2, eps, weight_decay, step, decoupled,
               mutate_grad=False, grad_scale=1.0):
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avg, exp_avg_sq)):
        p = p.data
        gg = g * grad_scale
        if decoupled:
            p.mul_(1.0 - lr * weight_decay)
        else:
            gg = gg + weight_decay * p
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        if decoupled:
            p.sub_((lr / bc1) * (m / (v.sqrt() / math.sqrt(bc2) + eps)))
        else:
            p.sub_(lr * (m / bc1) / ((v / bc2).sqrt() + eps))
        if mutate_grad and not decoupled and weight_decay:
            g.copy_(gg)
        if shadows is not None and shadows[i] is not None:
            shadows[i].copy_(p.to(shadows[i].dtype))


def amp_unscale(grads, state):
    inv = 1.0 / state[0]
    for g in grads:
        g.mul_(inv)
        if not torch.isfinite(g).all():
            state[2] = 1.0


def amp_update(state, growth, backoff, interval):
    if float(state[2]) != 0.0:
        state[0] *= backoff
        state[1] = 0.0
    else:
        t = float(state[1]) + 1.0
        if int(t) == interval:
            ns = state[0] * growth
            if torch.isfinite(ns):
                state[0] = ns
            state[1] = 0.0
        else:
            state[1] = t
    state[2] = 0.0


def sgd_step(params, grads, bufs, shadows, *, lr, momentum, dampening, weight_decay, first_step):
    for i, (p, g) in enumerate(zip(params, grads)):
        p = p.data
        gg = g + weight_decay * p if weight_decay else g.clone()
        if bufs is not None:
            if first_step:
                bufs[i].copy_(gg)
            else:
                bufs[i].mul_(momentum).add_(gg, alpha=1 - dampening)
            gg = bufs[i].clone()
        g.copy_(gg)
        p.sub_(lr * gg)


def ce_fwd_bwd(logits2d, labels, seq, shift, ignore_index=-100, denom_mode=0, denom_rows=0, grad_factor=1.0, grad_factor_dev=None):
    loss_out, lse = ce_fwd(logits2d, labels, seq, shift, ignore_index, denom_mode, denom_rows)
    g = torch.tensor([float(grad_factor) * (float(grad_factor_dev[0]) if grad_factor_dev is not None else 1.0)])
    return loss_out, lse, ce_bwd(logits2d, labels, lse, loss_out, g, seq, shift, ignore_index)


def ce_fused_ok(logits2d):
    return logits2d.stride(1) == 1 and logits2d.stride(0) == logits2d.shape[1]


SCALE_IF_PASSES = [0]


def scale_if_(x2d, s_dev, applied=1.0, applied_dev=None):
    have = float(applied) * (float(applied_dev[0]) if applied_dev is not None else 1.0)
    if float(s_dev[0]) != have:
        SCALE_IF_PASSES[0] += 1
        x2d.copy_((x2d.float() * (float(s_dev[0]) / have)).to(x2d.dtype))
    return x2d


class BlockActs:
    """Emulated counterpart of ops.BlockActs: the saved activations as plain tensors.  The product keeps ONE slab tensor (through
    save_for_backward) plus tensor-free geometry on the autograd node; here the "slab" is an empty tensor that carries the dict of
    activation tensors as a Python attribute and the geometry is the dict of everything else, so the host logic (ops.LazyKV,
    BlockActs.rebuild in the backward) runs unchanged AND the lifetime of the activations follows the slab's, as in the product."""

    def __init__(self, **kw):
        tensors = {k: v for k, v in kw.items() if isinstance(v, torch.Tensor)}
        self._geo = {k: v for k, v in kw.items() if not isinstance(v, torch.Tensor)}
        self.slab = torch.empty(0)
        self.slab._emu = tensors

    def __getattr__(self, name):
        d = self.__dict__
        if name in d.get("_geo", {}):
            return d["_geo"][name]
        slab = d.get("slab")
        if slab is not None and name in slab._emu:
            return slab._emu[name]
        raise AttributeError(name)

    def geometry(self):
        return self._geo

    @staticmethod
    def rebuild(slab, geo):
        return BlockActs(**geo, **slab._emu)


def _fused_desc(B, S, nh, hd):
    from cleantransformer_amd import ops
    return ops.fused_qkv_desc(B, S, nh, hd, causal=S > 1)


def _block_desc(B, S, nh, hd, flags, attn_scale, future_fill):
    from cleantransformer_amd import ops
    H = nh * hd
    if flags & 1:                                          # q | k | v blocked (GPT-2)
        st = (S * 3 * H, hd, 3 * H)
        return ops._strided_desc(B, nh, S, S, hd, st, st, st, (S * H, hd, H), attn_scale or 1.0 / math.sqrt(hd), S > 1, future_fill=future_fill), H
    d = ops.fused_qkv_desc(B, S, nh, hd, causal=S > 1)
    d.scale = attn_scale or d.scale
    d.future_fill = future_fill
    return d, hd


def bloom_block_fwd(x2, params, mask, slopes, eps, post_ln_res, B, S, nh, flags=0, attn_scale=0.0, future_fill=0.0, K=None):
    """The kernel sequence ctmi_bloom_block_fwd issues (csrc/block.hip), on the emulated kernels — or, with K =
    cleantransformer_amd.ops, on the individually verified HIP kernels (the -m gpu tests check the one-call form against it)."""
    K = K or _THIS
    layernorm_fwd, gemm, attn_fwd = K.layernorm_fwd, K.gemm, K.attn_fwd
    ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2 = params
    T, H = x2.shape
    hd = H // nh
    cd = x2.dtype
    ln1, mean1, rstd1 = layernorm_fwd(x2, ln1_w, ln1_b, eps)
    def lin(x, w, N, Kd, **kw):                            # flags & 4: the weight is stored [in, out] (Conv1D) and read K-major
        if flags & 4:
            return gemm(x, Kd, False, w, N, True, T, N, Kd, **kw)
        return gemm(x, Kd, False, w, Kd, False, T, N, Kd, **kw)

    qkv = lin(ln1, wqkv, 3 * H, H, bias=bqkv)
    desc, part = _block_desc(B, S, nh, hd, flags, attn_scale, future_fill)
    att = torch.empty((T, H), dtype=cd, device=x2.device)
    stat_m, stat_l = attn_fwd(qkv, qkv[:, part:], qkv[:, 2 * part:], att, desc, slopes, mask)
    h1 = lin(att, wd, H, H, bias=bd, residual=ln1 if post_ln_res else x2)
    ln2, mean2, rstd2 = layernorm_fwd(h1, ln2_w, ln2_b, eps)
    u = torch.empty((T, 4 * H), dtype=cd, device=x2.device)
    g = lin(ln2, w1, 4 * H, H, bias=b1, epilogue=1, aux_out=u)
    out = lin(g, w2, H, 4 * H, bias=b2, residual=ln2 if post_ln_res else h1)
    return BlockActs(B=B, S=S, H=H, nh=nh, dtype=cd, desc=desc, part=part, flags=flags, ln1=ln1, mean1=mean1, rstd1=rstd1, qkv=qkv, att=att, stat_m=stat_m,
                     stat_l=stat_l, h1=h1, mean2=mean2, rstd2=rstd2, ln2=ln2, u=u, g=g, out=out)


def bloom_block_bwd(a, x2, params, mask, slopes, eps, post_ln_res, dout2, use_side_stream=True, K=None, defer_join=False):
    """The kernel sequence ctmi_bloom_block_bwd issues (csrc/block.hip), on the emulated kernels (or on K = ops, see above)."""
    K = K or _THIS
    layernorm_bwd, gemm, attn_bwd, colsum = K.layernorm_bwd, K.gemm, K.attn_bwd, K.colsum
    ln1_w, ln1_b, wqkv, bqkv, wd, bd, ln2_w, ln2_b, w1, b1, w2, b2 = params
    T, H = x2.shape
    hd = H // a.nh
    post = post_ln_res

    def dgrad(dy, w, **kw):
        if a.flags & 4:                                    # w is [in, out]: the row-major B operand of dx = dy w^T
            return gemm(dy, dy.shape[1], False, w, w.shape[1], False, T, w.shape[0], dy.shape[1], **kw)
        return gemm(dy, dy.shape[1], False, w, w.shape[1], True, T, w.shape[1], dy.shape[1], **kw)

    def wgrad(dy, x):
        if a.flags & 2:                                    # Conv1D weights: gradient in [in, out]
            return gemm(x, x.shape[1], True, dy, dy.shape[1], True, x.shape[1], dy.shape[1], T, out_f32=True)
        return gemm(dy, dy.shape[1], True, x, x.shape[1], True, dy.shape[1], x.shape[1], T, out_f32=True)

    dw2, db2 = wgrad(dout2, a.g), colsum(dout2)
    du = dgrad(dout2, w2, epilogue=2, aux_in=a.u)
    dw1, db1 = wgrad(du, a.ln2), colsum(du)
    dln2 = dgrad(du, w1, residual=dout2 if post else None)
    dh1, dln2_w, dln2_b = layernorm_bwd(dln2, a.h1, ln2_w, a.mean2, a.rstd2, dres=None if post else dout2)
    dwd, dbd = wgrad(dh1, a.att), colsum(dh1)
    datt = dgrad(dh1, wd)
    dqkv = torch.empty_like(a.qkv)
    pt = a.part
    attn_bwd(a.qkv, a.qkv[:, pt:], a.qkv[:, 2 * pt:], a.att, datt, a.stat_m, a.stat_l, dqkv, dqkv[:, pt:], dqkv[:, 2 * pt:], a.desc,
             slopes, mask)
    dwqkv, dbqkv = wgrad(dqkv, a.ln1), colsum(dqkv)
    dln1 = dgrad(dqkv, wqkv, residual=dh1 if post else None)
    dx, dln1_w, dln1_b = layernorm_bwd(dln1, x2, ln1_w, a.mean1, a.rstd1, dres=None if post else dh1)
    return dx, [dln1_w, dln1_b, dwqkv, dbqkv, dwd, dbd, dln2_w, dln2_b, dw1, db1, dw2, db2]


import sys as _sys

_THIS = _sys.modules[__name__]


def install(monkeypatch):
    """Patch cleantransformer_amd.ops in place (pytest's monkeypatch undoes it after the test)."""
    from cleantransformer_amd import ops
    for name in ("layernorm_fwd", "layernorm_bwd", "gemm", "colsum", "MaskInfo", "attn_fwd", "attn_bwd", "embed_fwd", "embed_bwd",
                 "dropout", "ce_fwd", "ce_bwd", "ce_fwd_bwd", "ce_fused_ok", "scale_if_", "BlockActs", "bloom_block_fwd", "bloom_block_bwd", "ce_soft_fwd", "ce_soft_bwd", "cast", "transpose_cast", "sumsq", "scale_", "scale_copy", "argmax_lastdim", "row_lse", "group_topk",
                 "scores_filter", "amp_unscale", "amp_update", "adamw_step", "sgd_step"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
