"""CPU restatement (fp32, plain torch CPU ops) of the reference's GPT / GPT-2 path (CleanTransformer/models/modeling_gpt.py).

TEST INFRASTRUCTURE — the oracle for ``cleantransformer_amd/models/modeling_gpt.py``; never imported by the product package.

Parity status: PINNED by ``tests/test_oracle_golden.py::test_gpt_*`` against ``tests/golden/tiny_gpt.npz``, produced in the
build container by importing the reference's own ``GPTLMHeadModel`` (``tests/golden/make_golden.py gpt``).

``file:line`` citations are relative to the reference checkout (firechecking/CleanTransformer @ 2024-10-16).  Written
functionally over an ordered dict of parameters keyed by the reference's ``state_dict`` names; gradients via autograd.
The reference's model has no loss: the training loss used here is the one ``ft_bloom.py``-style SFT applies to a causal
LM — torch cross entropy of ``logits[:, :-1]`` against ``labels[:, 1:]`` (restated as :func:`oracle.bloom_ref.cross_entropy`).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from .bloom_ref import AdamState, adamw_update, cross_entropy, gelu_tanh, grad_norm, layernorm

Tensor = torch.Tensor


class GPTShape:
    def __init__(self, vocab_size: int, n_embd: int, n_layer: int, n_head: int, n_positions: int, version: str = "gpt2",
                 eps: float = 1e-5):
        self.vocab_size, self.n_embd, self.n_layer, self.n_head = vocab_size, n_embd, n_layer, n_head
        self.n_positions, self.version, self.eps = n_positions, version, eps


def param_names(s: GPTShape) -> List[str]:
    """named_parameters() order of the reference GPTLMHeadModel (lm_head.weight is the tied tokens_embed, visited once)."""
    names = ["gpt.tokens_embed.weight", "gpt.position_embed.weight"]
    for i in range(s.n_layer):
        pre = f"gpt.blocks.{i}."
        names += [pre + "attn.c_attn.weight", pre + "attn.c_attn.bias", pre + "attn.c_proj.weight", pre + "attn.c_proj.bias",
                  pre + "norm1.weight", pre + "norm1.bias", pre + "mlp.0.weight", pre + "mlp.0.bias",
                  pre + "mlp.2.weight", pre + "mlp.2.bias", pre + "norm2.weight", pre + "norm2.bias"]
    if s.version != "gpt":
        names += ["gpt.ln_f.weight", "gpt.ln_f.bias"]
    return names


def param_shape(s: GPTShape, name: str) -> Tuple[int, ...]:
    h = s.n_embd
    if name == "gpt.tokens_embed.weight":
        return (s.vocab_size, h)
    if name == "gpt.position_embed.weight":
        return (s.n_positions, h)
    table = {"attn.c_attn": (h, 3 * h), "attn.c_proj": (h, h), "mlp.0": (h, 4 * h), "mlp.2": (4 * h, h)}   # Conv1D: [in, out]
    for k, shp in table.items():
        if ("." + k + ".") in name:
            return shp if name.endswith("weight") else (shp[1],)
    return (h,)


def det_init(s: GPTShape) -> "OrderedDict[str, Tensor]":
    """Same recipe as the Bloom goldens: parameter i <- randn(seed 1000+i), 0.02 scale for matrices and
    biases, 1 + 0.1 r for LayerNorm weights."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    for i, name in enumerate(param_names(s)):
        shp = param_shape(s, name)
        r = torch.randn(shp, generator=torch.Generator().manual_seed(1000 + i))
        if len(shp) > 1:
            out[name] = r * 0.02
        elif ("norm" in name or "ln_f" in name) and name.endswith("weight"):
            out[name] = 1 + 0.1 * r
        else:
            out[name] = 0.02 * r
    return out


def conv1d(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """modeling_gpt.py:45-46: linear with the weight stored [in, out]."""
    return x @ w + b


def attention(p: Dict[str, Tensor], pre: str, x: Tensor, add_mask: Optional[Tensor], n_head: int, past=None):
    """modeling_gpt.py:67-108 (scale=True as GPTModel builds its blocks, :162)."""
    B, S, H = x.shape
    qkv = conv1d(x, p[pre + "c_attn.weight"], p[pre + "c_attn.bias"])
    q, k, v = qkv.split(H, dim=-1)
    sp = lambda t: t.view(B, S, n_head, -1).permute(0, 2, 1, 3)                          # noqa: E731  (:61-64)
    q, k, v = sp(q), sp(k), sp(v)
    if past is not None:
        k = torch.cat((past[0], k), dim=-2)
        v = torch.cat((past[1], v), dim=-2)
    present = (k, v)
    o = attention_core(q, k, v, add_mask).transpose(1, 2).contiguous().view(B, S, H)
    return conv1d(o, p[pre + "c_proj.weight"], p[pre + "c_proj.bias"]), present


def attention_core(q: Tensor, k: Tensor, v: Tensor, add_mask: Optional[Tensor], score_bias: Optional[Tensor] = None) -> Tensor:
    """modeling_gpt.py:86-95 on split heads q [B,nh,S,hd], k / v [B,nh,Sk,hd]: scores / sqrt(hd), the causal future REPLACED by -1e4
    (`w*b - 1e4*(1-b)`, :88-89), the additive key mask (:91-92), softmax, P.V -> [B,nh,S,hd].  `score_bias` ([B,nh,1|S,Sk], added to
    the scaled scores before the replacement) is not part of the reference: the HIP kernels that implement this fill also serve
    ALiBi models, and the tests check that combination against the same arithmetic."""
    S, Sk = q.size(-2), k.size(-2)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(v.size(-1))
    if score_bias is not None:
        w = w + score_bias
    b = torch.tril(torch.ones(Sk, Sk, dtype=w.dtype))[Sk - S:Sk, :Sk].view(1, 1, S, Sk)     # :88
    w = w * b + -1e4 * (1 - b)                                                              # :89
    if add_mask is not None:
        w = w + add_mask                                                                    # :91-92
    w = torch.softmax(w, dim=-1)
    return torch.matmul(w, v)


def block(p: Dict[str, Tensor], i: int, x: Tensor, add_mask, s: GPTShape, past=None):
    """modeling_gpt.py:136-150 (dropouts inactive)."""
    pre = f"gpt.blocks.{i}."
    mlp = lambda t: conv1d(gelu_tanh(conv1d(t, p[pre + "mlp.0.weight"], p[pre + "mlp.0.bias"])),   # noqa: E731
                           p[pre + "mlp.2.weight"], p[pre + "mlp.2.bias"])
    if s.version == "gpt":
        a, present = attention(p, pre + "attn.", x, add_mask, s.n_head, past)
        n1 = layernorm(x + a, p[pre + "norm1.weight"], p[pre + "norm1.bias"], s.eps)
        out = layernorm(n1 + mlp(n1), p[pre + "norm2.weight"], p[pre + "norm2.bias"], s.eps)
    else:
        a, present = attention(p, pre + "attn.", layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], s.eps),
                               add_mask, s.n_head, past)
        x = x + a
        out = x + mlp(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], s.eps))
    return out, present


def gpt_forward(p: Dict[str, Tensor], s: GPTShape, input_ids: Tensor, attention_mask: Tensor, labels: Optional[Tensor] = None,
                pasts=None, position_ids: Optional[Tensor] = None, segment_ids: Optional[Tensor] = None):
    """modeling_gpt.py:164-193 + 206-214; returns (loss|None, logits, hidden, presents)."""
    S = input_ids.shape[1]
    if position_ids is None:
        pos = attention_mask.long().cumsum(-1) - 1                                          # :166-169
        pos = pos.masked_fill(attention_mask == 0, 1)[:, -S:]
    else:
        pos = position_ids
    am = attention_mask[:, None, None, :].to(torch.float32)
    am = (1.0 - am) * torch.finfo(torch.float32).min                                        # :171-175
    h = p["gpt.tokens_embed.weight"][input_ids] + p["gpt.position_embed.weight"][pos]
    if segment_ids is not None:
        h = h + p["gpt.tokens_embed.weight"][segment_ids.view(-1, segment_ids.size(-1))]    # :184 (segments share the token table)
    presents = []
    for i in range(s.n_layer):
        h, pr = block(p, i, h, am, s, None if pasts is None else pasts[i])
        presents.append(pr)
    if s.version != "gpt":
        h = layernorm(h, p["gpt.ln_f.weight"], p["gpt.ln_f.bias"], s.eps)
    logits = h @ p["gpt.tokens_embed.weight"].t()                                           # tied lm_head (:203-204, :212)
    loss = None
    if labels is not None:
        V = logits.shape[-1]
        loss = cross_entropy(logits[:, :-1, :].reshape(-1, V), labels[:, 1:].reshape(-1))
    return loss, logits, h, presents


def loss_and_grads(p: Dict[str, Tensor], s: GPTShape, input_ids: Tensor, attention_mask: Tensor):
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in p.items())
    loss, logits, hidden, _ = gpt_forward(leaves, s, input_ids, attention_mask, labels=input_ids.clone())
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), logits.detach(), hidden.detach(), OrderedDict(zip(leaves.keys(), grads))


def train_step(p: Dict[str, Tensor], s: GPTShape, input_ids: Tensor, attention_mask: Tensor, state: AdamState,
               lr: float = 1e-5, weight_decay: float = 0.01):
    """forward -> backward -> torch.optim.AdamW(lr) update in place on p; returns (loss, grad_norm)."""
    loss, _, _, grads = loss_and_grads(p, s, input_ids, attention_mask)
    gn = grad_norm(grads.values())
    state.t += 1
    with torch.no_grad():
        for k in p:
            adamw_update(p[k], grads[k].clone(), state.m[k], state.v[k], state.t, lr, 0.9, 0.999, 1e-8, weight_decay, True)
    return float(loss), gn


def greedy_decode(p: Dict[str, Tensor], s: GPTShape, input_ids: Tensor, attention_mask: Tensor, max_gen_len: int,
                  pad_id: int = 0) -> Tensor:
    """generation_util.py:57-119 with do_sample=False over the GPT forward (KV cache :75-80)."""
    ids, mask = input_ids.clone(), attention_mask.clone()
    max_len = max_gen_len + ids.shape[-1]
    pasts, step = None, 0
    while True:
        with torch.no_grad():
            _, logits, _, pasts = gpt_forward(p, s, ids[:, step:], mask, None, pasts)
        nxt = torch.argmax(logits[:, -1, :], dim=-1)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        mask = torch.cat([mask, mask[:, -1:]], dim=-1)
        step = ids.shape[1] - 1
        if step > max_len:
            break
    return ids.view(ids.shape[0], 1, -1)
