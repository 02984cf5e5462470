"""CPU restatement (fp32 / fp64, plain torch CPU ops) of the reference SFT hot path.

TEST INFRASTRUCTURE — the oracle.  It is the checker for the HIP path and the timed
``cpu_baseline`` of ``bench.py``; it is never imported by the product package.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors generated in the build container by importing the reference itself
(``tests/golden/make_golden.py``; SURVEY.md Appendix A anchors) and against the reference's
own printed known-answers (loss.py:76-100).

All ``file:line`` citations are relative to the reference checkout
(firechecking/CleanTransformer @ 2024-10-16).

The functions are written functionally over an ordered ``dict`` of parameter tensors whose
keys are the reference's ``state_dict`` names (inference_bloom.py:22-35), so the same
dict can be loaded into the product model.  Gradients come from torch autograd over this
restated forward (the reference has no hand-written backward except GELU, which is
restated explicitly in :func:`gelu_tanh_bwd`).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """transformer.py:71-89.  mean over the trailing ``weight.dim()`` dims;
    ``std = sqrt(mean((x-mean)^2 + eps))`` (eps inside the mean == biased var + eps)."""
    nd = weight.dim()
    lead = list(x.shape[:-nd])
    flat = x.reshape(*lead, -1)
    n = flat.shape[-1]
    mean = (flat.sum(dim=-1) / n).reshape(*lead, *([1] * nd))
    dev = x - mean
    sq = (dev.pow(2) + eps).reshape(*lead, -1)
    std = (sq.sum(dim=-1) / n).reshape(*lead, *([1] * nd)).pow(0.5)
    return weight * (dev / std) + bias


GELU_C0 = 0.79788456          # modeling_bloom.py:344
GELU_C1 = 0.044715            # modeling_bloom.py:344
GELU_C2 = 0.1070322243        # modeling_bloom.py:361-362


def gelu_tanh(x: Tensor) -> Tensor:
    """modeling_bloom.py:335-344 (bloom_gelu_forward)."""
    return x * 0.5 * (1.0 + torch.tanh(GELU_C0 * x * (1 + GELU_C1 * x * x)))


def gelu_tanh_bwd(g: Tensor, x: Tensor) -> Tensor:
    """modeling_bloom.py:348-363 (bloom_gelu_back) — closed-form derivative × upstream grad."""
    t = torch.tanh(GELU_C0 * x * (1 + GELU_C1 * x * x))
    ff = 0.5 * x * ((1 - t * t) * (GELU_C0 + GELU_C2 * x * x)) + 0.5 * (1 + t)
    return ff * g


def alibi_slopes(num_heads: int) -> Tensor:
    """modeling_bloom.py:312-325.  fp32 ``pow`` exactly as the reference computes it."""
    p2 = 2 ** math.floor(math.log2(num_heads))
    base = torch.tensor(2 ** (-(2 ** -(math.log2(p2) - 3))), dtype=torch.float32)
    slopes = torch.pow(base, torch.arange(1, 1 + p2, dtype=torch.int32))
    if p2 != num_heads:
        extra_base = torch.tensor(2 ** (-(2 ** -(math.log2(2 * p2) - 3))), dtype=torch.float32)
        n_extra = min(p2, num_heads - p2)
        extra = torch.pow(extra_base, torch.arange(1, 1 + 2 * n_extra, 2, dtype=torch.int32))
        slopes = torch.cat([slopes, extra], dim=0)
    return slopes


def alibi_positions(attention_mask: Tensor) -> Tensor:
    """modeling_bloom.py:328: ``(cumsum(mask) - 1) * mask`` -> [B, S] (same dtype as mask)."""
    return (attention_mask.cumsum(dim=-1) - 1) * attention_mask


def build_alibi(attention_mask: Tensor, num_heads: int, dtype=torch.float32) -> Tensor:
    """modeling_bloom.py:309-331 -> [B*nh, 1, S]."""
    b, s = attention_mask.shape
    pos = alibi_positions(attention_mask)[:, None, :]
    alibi = alibi_slopes(num_heads)[..., None] * pos
    return alibi.reshape(b * num_heads, 1, s).to(dtype)


def causal_key_mask(attention_mask: Tensor, q_len: int) -> Tensor:
    """modeling_bloom.py:176-185 (_attn_mask).  bool [B,1,q_len,S_k]; True == masked.
    The causal part is only applied when q_len > 1 and is a q_len x q_len tril
    (so it is only shape-compatible with the key axis when there is no KV cache)."""
    b, sk = attention_mask.shape
    keep = attention_mask[:, None, None, :].expand(b, 1, q_len, sk).to(torch.bool)
    masked = ~keep
    if q_len > 1:
        tri = torch.tril(torch.ones(q_len, q_len)).to(torch.bool)[None, None]
        masked = masked | ~tri.expand(b, 1, q_len, q_len)
    return masked


def attention_core(qkv: Tensor, alibi: Tensor, masked: Tensor, num_heads: int,
                   past: Optional[Tuple[Tensor, Tensor]] = None):
    """modeling_bloom.py:80-116: split the head-interleaved fused QKV, scores =
    alibi + q.k/sqrt(hd), masked_fill(finfo.min), softmax, P.V, merge heads.
    Returns (context [B,S,H], (k,v) each [B,nh,S_k,hd])."""
    b, s, three_h = qkv.shape
    h = three_h // 3
    hd = h // num_heads
    x = qkv.view(b, s, num_heads, 3, hd)
    q = x[..., 0, :].transpose(1, 2)
    k = x[..., 1, :].transpose(1, 2)
    v = x[..., 2, :].transpose(1, 2)
    if past is not None:
        k = torch.cat((past[0], k), dim=-2)
        v = torch.cat((past[1], v), dim=-2)
    present = (k, v)
    sk = k.shape[-2]
    q2 = q.reshape(b * num_heads, s, hd)
    kt = k.transpose(2, 3).reshape(b * num_heads, hd, sk)
    v2 = v.reshape(b * num_heads, sk, hd)
    scores = alibi.baddbmm(batch1=q2, batch2=kt, beta=1.0, alpha=1.0 / math.sqrt(hd))
    if scores.dtype == torch.float16:
        scores = scores.float()
    scores = torch.masked_fill(scores.view(b, num_heads, s, sk), masked,
                               torch.finfo(scores.dtype).min)
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs.view(b * num_heads, s, sk), v2)
    ctx = ctx.view(b, num_heads, s, hd).transpose(1, 2).contiguous().view(b, s, h)
    return ctx, present


def cross_entropy(logits: Tensor, target: Tensor, impl: str = "exact") -> Tensor:
    """What Bloom actually calls: torch.nn.CrossEntropyLoss() (modeling_bloom.py:11,228) —
    stable log-softmax, NLL, mean over rows (no label ever equals -100 on this path).

    impl="exact" (default): logsumexp formulation; agrees with an fp64 evaluation to fp32 round-off.
    impl="torch": call torch's own CE kernel, i.e. literally the third-party arithmetic the
    reference runs.  At V=250880 torch's fp32 CPU log_softmax carries a systematic summation
    error (loss 1.7e-6, grads 2.7e-5 relative vs fp64 — measured in tests/test_oracle_golden.py);
    both are inside the 1e-4 parity bar, "exact" is the better checker for a GPU kernel."""
    if impl == "torch":
        return torch.nn.functional.cross_entropy(logits, target)
    x = logits if logits.dtype == torch.float64 else logits.float()
    lse = torch.logsumexp(x, dim=-1)
    picked = x.gather(1, target.view(-1, 1)).squeeze(1)
    return (lse - picked).sum() / logits.shape[0]


def cross_entropy_repo(logits: Tensor, target: Tensor, reduction: str = "mean") -> Tensor:
    """loss.py:34-49 — the repo's own un-stabilised CE (index or probability targets)."""
    e = torch.exp(logits)
    logsm = torch.log(e / e.sum(dim=-1, keepdim=True))
    if target.dim() == logits.dim() - 1:
        loss = -logsm.gather(1, target.unsqueeze(1)).sum()
    else:
        loss = -(target * logsm).sum()
    if reduction == "mean":
        loss = loss / logits.shape[0]
    return loss


def log_softmax_repo(x: Tensor, dim: int) -> Tensor:
    """loss.py:52-60 (note the +1e-9 in the denominator)."""
    e = torch.exp(x)
    return torch.log(e / (e.sum(dim=dim, keepdim=True) + 1e-9))


def nll_repo(logp: Tensor, target: Tensor, reduction: str = "mean") -> Tensor:
    """loss.py:63-73."""
    r = -logp.gather(1, target.unsqueeze(1)).sum()
    return r / logp.shape[0] if reduction == "mean" else r


def mse_repo(a: Tensor, b: Tensor, reduction: str = "mean") -> Tensor:
    """loss.py:17-26."""
    d = (a - b).pow(2)
    return d.mean() if reduction == "mean" else d.sum()


# --------------------------------------------------------------------------------------
# optimizers
# --------------------------------------------------------------------------------------
def adamw_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, t: int, lr: float,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 weight_decay: float = 0.0, decoupled: bool = False) -> None:
    """One in-place Adam step on (p, m, v); ``t`` is the 1-based step index.

    decoupled=False: optimizer.py:75-95 (repo AdamW == Adam + L2: ``g += wd*p`` mutating the
    grad in place, ``p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)``).
    decoupled=True: torch.optim.AdamW as called at ft_bloom.py:70 (``p *= 1-lr*wd`` first;
    ``denom = sqrt(v)/sqrt(1-b2^t) + eps``; ``p -= (lr/(1-b1^t)) * m/denom``).
    """
    if decoupled:
        if weight_decay:
            p.mul_(1.0 - lr * weight_decay)
    elif weight_decay:
        g.add_(p, alpha=weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    if decoupled:
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
    else:
        p.sub_(lr * (m / bc1) / ((v / bc2).sqrt() + eps))


def sgd_update(p: Tensor, g: Tensor, buf: Optional[Tensor], lr: float, momentum: float = 0.0,
               dampening: float = 0.0, weight_decay: float = 0.0) -> Tensor:
    """optimizer.py:30-48.  Returns the momentum buffer (created on first use)."""
    if weight_decay:
        g = g + weight_decay * p
    if momentum:
        if buf is None:
            buf = g.clone()
        else:
            buf.mul_(momentum).add_(g, alpha=1.0 - dampening)
        g = buf
    p.sub_(lr * g)
    return buf


# --------------------------------------------------------------------------------------
# Bloom model, functional over a state-dict-keyed parameter dict
# --------------------------------------------------------------------------------------
class BloomShape:
    def __init__(self, vocab_size: int, hidden_size: int, n_layer: int, num_attention_heads: int,
                 layer_norm_epsilon: float = 1e-5,
                 apply_residual_connection_post_layernorm: bool = False):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.n_layer = n_layer
        self.n_head = num_attention_heads
        self.eps = layer_norm_epsilon
        self.post_ln_residual = apply_residual_connection_post_layernorm


def param_names(shape: BloomShape) -> List[str]:
    """``named_parameters()`` order of the reference model;
    ``lm_head.weight`` is the tied table and is not listed again."""
    names = ["bloom.word_embeddings.weight",
             "bloom.word_embeddings_layernorm.weight", "bloom.word_embeddings_layernorm.bias"]
    for i in range(shape.n_layer):
        for mod in ("input_layernorm", "self_attention.query_key_value", "self_attention.dense",
                    "post_attention_layernorm", "mlp.dense_h_to_4h", "mlp.dense_4h_to_h"):
            names += [f"bloom.blocks.{i}.{mod}.weight", f"bloom.blocks.{i}.{mod}.bias"]
    names += ["bloom.ln_f.weight", "bloom.ln_f.bias"]
    return names


def param_shape(shape: BloomShape, name: str) -> Tuple[int, ...]:
    h, v = shape.hidden_size, shape.vocab_size
    leaf = name.rsplit(".", 2)[-2:]
    mod, kind = leaf[0], leaf[1]
    if name == "bloom.word_embeddings.weight":
        return (v, h)
    table = {"query_key_value": (3 * h, h), "dense": (h, h),
             "dense_h_to_4h": (4 * h, h), "dense_4h_to_h": (h, 4 * h)}
    if mod in table:
        return table[mod] if kind == "weight" else (table[mod][0],)
    return (h,)


def det_init(shape: BloomShape, dtype=torch.float32) -> "OrderedDict[str, Tensor]":
    """SURVEY Appendix A recipe: parameter i <- randn(seed 1000+i) scaled by kind."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    for i, name in enumerate(param_names(shape)):
        shp = param_shape(shape, name)
        r = torch.randn(shp, generator=torch.Generator().manual_seed(1000 + i))
        if len(shp) > 1:
            val = r * 0.02
        elif name.endswith("layernorm.weight") or name.endswith("ln_f.weight"):
            val = 1 + 0.1 * r
        else:
            val = 0.02 * r
        out[name] = val.to(dtype)
    return out


def bloom_block(p: Dict[str, Tensor], i: int, x: Tensor, alibi: Tensor, masked: Tensor,
                shape: BloomShape, past=None):
    """modeling_bloom.py:142-159 + 76-124 + 255-271."""
    pre = f"bloom.blocks.{i}."
    ln1 = layernorm(x, p[pre + "input_layernorm.weight"], p[pre + "input_layernorm.bias"], shape.eps)
    res = ln1 if shape.post_ln_residual else x
    qkv = torch.nn.functional.linear(ln1, p[pre + "self_attention.query_key_value.weight"],
                                     p[pre + "self_attention.query_key_value.bias"])
    ctx, present = attention_core(qkv, alibi, masked, shape.n_head, past)
    attn = res + torch.nn.functional.linear(ctx, p[pre + "self_attention.dense.weight"],
                                            p[pre + "self_attention.dense.bias"])
    ln2 = layernorm(attn, p[pre + "post_attention_layernorm.weight"],
                    p[pre + "post_attention_layernorm.bias"], shape.eps)
    res2 = ln2 if shape.post_ln_residual else attn
    u = torch.nn.functional.linear(ln2, p[pre + "mlp.dense_h_to_4h.weight"], p[pre + "mlp.dense_h_to_4h.bias"])
    out = res2 + torch.nn.functional.linear(gelu_tanh(u), p[pre + "mlp.dense_4h_to_h.weight"],
                                            p[pre + "mlp.dense_4h_to_h.bias"])
    return out, present


def bloom_forward(p: Dict[str, Tensor], shape: BloomShape, input_ids: Tensor,
                  attention_mask: Tensor, labels: Optional[Tensor] = None, pasts=None,
                  ce_impl: str = "exact"):
    """modeling_bloom.py:187-205 + 218-232.
    Returns (loss or None, logits [B,S,V], hidden [B,S,H], presents)."""
    if pasts is None:
        pasts = [None] * shape.n_layer
    emb = p["bloom.word_embeddings.weight"]
    x = torch.nn.functional.embedding(input_ids, emb)
    x = layernorm(x, p["bloom.word_embeddings_layernorm.weight"],
                  p["bloom.word_embeddings_layernorm.bias"], shape.eps)
    alibi = build_alibi(attention_mask, shape.n_head, dtype=x.dtype)
    masked = causal_key_mask(attention_mask, input_ids.shape[1])
    presents = []
    for i in range(shape.n_layer):
        x, pr = bloom_block(p, i, x, alibi, masked, shape, pasts[i])
        presents.append(pr)
    hidden = layernorm(x, p["bloom.ln_f.weight"], p["bloom.ln_f.bias"], shape.eps)
    logits = torch.nn.functional.linear(hidden, emb)          # tied, no bias (:213-216)
    loss = None
    if labels is not None:
        b, s, v = logits.shape
        loss = cross_entropy(logits[..., :-1, :].reshape(b * (s - 1), v),
                             labels[..., 1:].reshape(b * (s - 1)), impl=ce_impl)
    return loss, logits, hidden, presents


def grad_norm(grads) -> float:
    """Global L2 norm in fp64 (what clip_grad_norm_ returns before clipping, trainer.py:491-498)."""
    tot = 0.0
    for g in grads:
        tot += float(g.double().pow(2).sum())
    return math.sqrt(tot)


def loss_and_grads(p: Dict[str, Tensor], shape: BloomShape, input_ids: Tensor,
                   attention_mask: Tensor, ce_impl: str = "exact"):
    """Forward with labels = input_ids.clone() (ft_bloom.py:52) and backward via autograd."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in p.items())
    loss, logits, hidden, _ = bloom_forward(leaves, shape, input_ids, attention_mask,
                                            labels=input_ids.clone(), ce_impl=ce_impl)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), logits.detach(), hidden.detach(), OrderedDict(zip(leaves.keys(), grads))


class AdamState:
    def __init__(self, p: Dict[str, Tensor]):
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in p.items())
        self.t = 0


def train_step(p: Dict[str, Tensor], shape: BloomShape, input_ids: Tensor, attention_mask: Tensor,
               state: AdamState, lr: float = 1e-5, weight_decay: float = 0.01,
               decoupled: bool = True, betas=(0.9, 0.999), eps: float = 1e-8):
    """ft_bloom.py:84-90 step: forward -> zero_grad -> backward -> optimizer.step (in place on p).
    Returns (loss, grad_norm)."""
    loss, _, _, grads = loss_and_grads(p, shape, input_ids, attention_mask)
    gn = grad_norm(grads.values())
    state.t += 1
    with torch.no_grad():
        for k in p:
            adamw_update(p[k], grads[k].clone(), state.m[k], state.v[k], state.t, lr,
                         betas[0], betas[1], eps, weight_decay, decoupled)
    return float(loss), gn


def greedy_decode(p: Dict[str, Tensor], shape: BloomShape, input_ids: Tensor, attention_mask: Tensor,
                  max_gen_len: int, end_ids=None, pad_id: int = 0) -> Tensor:
    """generation_util.py:57-119 with do_sample=False: KV-cached argmax decode.
    Replicates the loop-exit quirk (``step > max_len`` -> max_gen_len+2 tokens)."""
    ids = input_ids.clone()
    mask = attention_mask.clone()
    max_len = max_gen_len + ids.shape[-1]
    pasts = None
    step = 0
    unfinished = torch.ones(ids.shape[0], dtype=torch.long)
    end = None if end_ids is None else torch.tensor(list(end_ids))
    while True:
        with torch.no_grad():
            _, logits, _, pasts = bloom_forward(p, shape, ids[:, step:], mask, None, pasts)
        nxt = torch.argmax(logits[:, -1, :], dim=-1)
        nxt = nxt * unfinished + pad_id * (1 - unfinished)
        if end is not None:
            unfinished = unfinished.mul(nxt.tile(end.shape[0], 1).ne(end.unsqueeze(1)).prod(dim=0))
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        mask = torch.cat([mask, mask[:, -1:]], dim=-1)
        step = ids.shape[1] - 1
        if unfinished.max() == 0 or step > max_len:
            break
    return ids.view(ids.shape[0], 1, -1)


# --------------------------------------------------------------------------------------
# generic MHA / post-LN block of transformer.py (BERT-style), for the AttentionLayer parity
# --------------------------------------------------------------------------------------
def mha_generic(x: Tensor, wq, bq, wk, bk, wv, bv, num_heads: int, add_mask: Optional[Tensor] = None):
    """transformer.py:30-58 — three Linear(H,H), softmax(QK^T/sqrt(hd) + mask) V, merge; no out-proj."""
    b, s, h = x.shape
    hd = h // num_heads

    def split(t):
        return t.view(b, s, num_heads, hd).permute(0, 2, 1, 3)

    q = split(torch.nn.functional.linear(x, wq, bq))
    k = split(torch.nn.functional.linear(x, wk, bk))
    v = split(torch.nn.functional.linear(x, wv, bv))
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(h / num_heads)
    if add_mask is not None:
        w = w + add_mask
    w = torch.softmax(w, dim=-1)
    o = torch.matmul(w, v)
    return o.transpose(1, 2).contiguous().view(b, s, h)


def post_ln_block(x: Tensor, prm: Dict[str, Tensor], num_heads: int, eps: float) -> Tensor:
    """transformer.py:107-121 with dropout p=0: LN1(x+attn(x)); LN2(y + W2 relu(W1 y))."""
    a = mha_generic(x, prm["attention.q_linear.weight"], prm["attention.q_linear.bias"],
                    prm["attention.k_linear.weight"], prm["attention.k_linear.bias"],
                    prm["attention.v_linear.weight"], prm["attention.v_linear.bias"], num_heads)
    y = layernorm(x + a, prm["norm1.weight"], prm["norm1.bias"], eps)
    f = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(y, prm["ffw.0.weight"], prm["ffw.0.bias"])),
                                   prm["ffw.2.weight"], prm["ffw.2.bias"])
    return layernorm(y + f, prm["norm2.weight"], prm["norm2.bias"], eps)
