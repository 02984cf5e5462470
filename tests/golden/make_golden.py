#!/usr/bin/env python3
"""Golden-vector generator — runs ONLY in the build container, where the reference checkout
is mounted read-only at /root/reference.  It imports the reference's own Python modules,
drives them on deterministic inputs and writes small data files (inputs + expected outputs)
next to this script.  The reference never travels to the GPU box; these fixtures do.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Files written:
    ops.npz          per-op I/O (LayerNorm, GELU fwd/bwd, ALiBi, mask, attention block, CE, optimizers)
    tiny_bloom.npz   tiny Bloom (V=211,H=64,nh=8,L=2,B=4,S=16): logits, every grad, 4-step trajectory,
                     left-padded variant (uniform-row quirk), greedy tokens
    c1_bloom.json    config-1 shape (V=250880,H=1024,nh=16,L=2,B=2,S=128): scalars + slices
    c5_bloom.json    Bloom-7B1 geometry (V=250880,H=4096,nh=32,hd=128), L=1,B=1,S=512: scalars + slices
    tiny_gpt.npz     tiny GPT-2 / GPT-1 (modeling_gpt.py): logits, grads, 3-step trajectory, greedy tokens
    ddp_tiny.npz     2- and 4-rank torch-DDP(gloo) averaged grads of the reference tiny model
    known_answers.json   the reference's own printed self-check values (loss.py:76-100 etc.)
"""
import hashlib
import json
import math
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
REF = "/root/reference"
if REF not in sys.path:
    sys.path.insert(0, REF)

import numpy as np
import torch

from CleanTransformer.transformer import LayerNorm as RefLayerNorm, AttentionLayer as RefAttention, \
    TransformerBlock as RefBlock
from CleanTransformer import loss as ref_loss
from CleanTransformer import optimizer as ref_opt
from CleanTransformer.models import modeling_bloom as rb

HERE = os.path.dirname(os.path.abspath(__file__))


def det_init(model):
    """SURVEY Appendix A."""
    with torch.no_grad():
        for i, (name, p) in enumerate(model.named_parameters()):
            r = torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 + i))
            if p.dim() > 1:
                p.copy_(r * 0.02)
            elif name.endswith("layernorm.weight") or name.endswith("ln_f.weight"):
                p.copy_(1 + 0.1 * r)
            else:
                p.copy_(0.02 * r)


def weights_sha(model):
    h = hashlib.sha256()
    for _, p in model.named_parameters():
        h.update(p.detach().float().contiguous().numpy().tobytes())
    return h.hexdigest()


def build(V, H, L, nh):
    cfg = rb.BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh)
    m = rb.BloomForCausalLM(cfg)
    m._tie_weight()
    det_init(m)
    m.train()
    return cfg, m


def gnorm(m):
    return math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()))


def npy(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------
def gen_ops():
    out = {}
    g = torch.Generator().manual_seed(4242)
    # LayerNorm (transformer.py:61-89), 1-D and 2-D normalized_shape
    x = torch.randn(3, 5, 48, generator=g) * 2 + 0.3
    ln = RefLayerNorm(48, eps=1e-5)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(48, generator=g))
        ln.bias.copy_(0.05 * torch.randn(48, generator=g))
    xr = x.clone().requires_grad_(True)
    y = ln(xr)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update(ln_x=npy(x), ln_w=npy(ln.weight), ln_b=npy(ln.bias), ln_y=npy(y), ln_gy=npy(gy),
               ln_gx=npy(xr.grad), ln_gw=npy(ln.weight.grad), ln_gb=npy(ln.bias.grad))
    x2 = torch.randn(3, 4, 6, generator=g)
    ln2 = RefLayerNorm([4, 6])
    out.update(ln2_x=npy(x2), ln2_y=npy(ln2(x2)))

    # GELU fwd / custom bwd (modeling_bloom.py:335-363)
    gx = torch.randn(4, 33, generator=g) * 3
    gg = torch.randn(4, 33, generator=g)
    out.update(gelu_x=npy(gx), gelu_y=npy(rb.bloom_gelu_forward(gx)), gelu_g=npy(gg),
               gelu_gx=npy(rb.bloom_gelu_back(gg, (gx,))))

    # ALiBi + mask (modeling_bloom.py:309-331, 176-185)
    for nh in (8, 16, 12):
        out[f"alibi_slopes_{nh}"] = npy(rb.build_alibi_tensor(torch.tensor([[0, 1, 1]]), nh, torch.float32))[:, 0, 2]
    am = torch.ones(3, 10, dtype=torch.long)
    am[1, 7:] = 0
    am[2, :4] = 0
    out.update(alibi_mask=npy(am), alibi_8=npy(rb.build_alibi_tensor(am, 8, torch.float32)))
    cfg = rb.BloomConfig(vocab_size=50, hidden_size=64, n_layer=1, num_attention_heads=8)
    bm = rb.BloomModel(cfg)
    out["attn_mask_bool"] = npy(bm._attn_mask(am, (3, 10)))

    # one Bloom attention layer fwd + grads (modeling_bloom.py:76-124), incl. left-pad (uniform row)
    att = rb.BloomAttentionLayer(cfg)
    with torch.no_grad():
        for i, p in enumerate(att.parameters()):
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(77 + i)) * (0.08 if p.dim() > 1 else 0.02))
    hs = torch.randn(3, 10, 64, generator=g)
    res = torch.randn(3, 10, 64, generator=g)
    hs_r, res_r = hs.clone().requires_grad_(True), res.clone().requires_grad_(True)
    alibi = rb.build_alibi_tensor(am, 8, torch.float32)
    mask = bm._attn_mask(am, (3, 10))
    o, (k, v) = att(hs_r, res_r, alibi, attention_mask=mask)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    out.update(att_hs=npy(hs), att_res=npy(res), att_out=npy(o), att_go=npy(go), att_ghs=npy(hs_r.grad),
               att_gres=npy(res_r.grad), att_k=npy(k), att_v=npy(v))
    for n, p in att.named_parameters():
        out["att_p_" + n] = npy(p)
        out["att_g_" + n] = npy(p.grad)

    # generic AttentionLayer + post-LN TransformerBlock (transformer.py:12-58, 92-121), dropout off
    class C:
        num_attention_heads = 4
        layer_norm_epsilong = 1e-5
        attention_probs_dropout_prob = 0.0
        hidden_size = 32
        hidden_dropout_prob = 0.0
    blk = RefBlock(C())
    with torch.no_grad():
        for i, p in enumerate(blk.parameters()):
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(300 + i)) * (0.15 if p.dim() > 1 else 0.05))
        blk.norm1.weight.add_(1.0)
        blk.norm2.weight.add_(1.0)
    bx = torch.randn(2, 7, 32, generator=g)
    bxr = bx.clone().requires_grad_(True)
    by = blk(bxr)
    bgo = torch.randn(by.shape, generator=g)
    by.backward(bgo)
    addmask = torch.zeros(2, 1, 1, 7)
    addmask[1, ..., 5:] = -10000.0
    out.update(blk_x=npy(bx), blk_y=npy(by), blk_go=npy(bgo), blk_gx=npy(bxr.grad),
               mha_y=npy(blk.attention(bx)), mha_addmask=npy(addmask), mha_y_masked=npy(blk.attention(bx, attention_mask=addmask)))
    for n, p in blk.named_parameters():
        out["blk_p_" + n] = npy(p)
        out["blk_g_" + n] = npy(p.grad)

    # CE: torch CE (what Bloom runs) and the repo's CE where finite (loss.py:29-49)
    lg = torch.randn(37, 211, generator=g) * 2
    tg = torch.randint(0, 211, (37,), generator=g)
    lgr = lg.clone().requires_grad_(True)
    l_t = torch.nn.CrossEntropyLoss()(lgr, tg)
    l_t.backward()
    pt = torch.softmax(torch.randn(37, 211, generator=g), dim=-1)
    out.update(ce_logits=npy(lg), ce_target=npy(tg), ce_torch=npy(l_t), ce_dlogits=npy(lgr.grad),
               ce_repo_mean=npy(ref_loss.CrossEntropyLoss('mean')(lg, tg)),
               ce_repo_sum=npy(ref_loss.CrossEntropyLoss('sum')(lg, tg)),
               ce_prob_target=npy(pt), ce_repo_prob=npy(ref_loss.CrossEntropyLoss('mean')(lg, pt)),
               logsm_repo=npy(ref_loss.LogSoftmax(dim=1)(lg)),
               nll_repo=npy(ref_loss.NLLLoss()(ref_loss.LogSoftmax(dim=1)(lg), tg)),
               mse_repo=npy(ref_loss.MSELoss()(lg, pt)))

    # optimizers: 50-step trajectories on a fixed quadratic-ish problem
    def traj(make_opt, steps=50):
        torch.manual_seed(5)
        w = (torch.randn(6, 5, generator=torch.Generator().manual_seed(11))).requires_grad_(True)
        b = (torch.randn(5, generator=torch.Generator().manual_seed(12))).requires_grad_(True)
        opt = make_opt([w, b])
        gen = torch.Generator().manual_seed(13)
        grads = []
        for _ in range(steps):
            xin = torch.randn(4, 6, generator=gen)
            tgt = torch.randn(4, 5, generator=gen)
            l = ((xin @ w + b - tgt) ** 2).sum()
            opt.zero_grad()
            l.backward()
            grads.append(np.concatenate([npy(w.grad).ravel(), npy(b.grad).ravel()]))
            opt.step()
        return npy(w), npy(b), np.stack(grads)

    out["opt_w0"] = npy(torch.randn(6, 5, generator=torch.Generator().manual_seed(11)))
    out["opt_b0"] = npy(torch.randn(5, generator=torch.Generator().manual_seed(12)))
    for wd in (0.0, 0.01):
        tag = "wd0" if wd == 0.0 else "wd01"
        w, b, gs = traj(lambda ps: ref_opt.AdamW(ps, lr=1e-2, weight_decay=wd))
        out[f"adam_repo_{tag}_w"], out[f"adam_repo_{tag}_b"] = w, b
        w, b, gs = traj(lambda ps: torch.optim.AdamW(ps, lr=1e-2, weight_decay=wd))
        out[f"adam_torch_{tag}_w"], out[f"adam_torch_{tag}_b"] = w, b
    w, b, _ = traj(lambda ps: ref_opt.SGD(ps, lr=1e-2, momentum=0.9, weight_decay=0.01))
    out["sgd_repo_w"], out["sgd_repo_b"] = w, b
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------
def run_steps(m, ids, am, nsteps, lr=1e-5):
    opt = torch.optim.AdamW(m.parameters(), lr=lr)          # ft_bloom.py:70
    rec = []
    first = {}
    for t in range(nsteps):
        (loss, logits, hid), _ = m(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        gn = gnorm(m)
        if t == 0:
            first = dict(loss=loss.detach().clone(), logits=logits.detach().clone(), hidden=hid.detach().clone(),
                         grads={n: p.grad.detach().clone() for n, p in m.named_parameters()})
        opt.step()
        rec.append((float(loss), gn))
    return rec, first


def gen_tiny():
    V, H, L, nh, B, S = 211, 64, 2, 8, 4, 16
    cfg, m = build(V, H, L, nh)
    sha = weights_sha(m)
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(7))
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 12:] = 0
    ids_sha = hashlib.sha256(ids.numpy().tobytes()).hexdigest()
    rec, first = run_steps(m, ids, am, 4)
    out = dict(cfg=np.array([V, H, L, nh, B, S]), ids=npy(ids), mask=npy(am),
               traj=np.array(rec, dtype=np.float64), loss0=npy(first["loss"]), logits0=npy(first["logits"]),
               hidden0=npy(first["hidden"]))
    for n, gten in first["grads"].items():
        out["g0_" + n] = npy(gten)
    for n, p in m.named_parameters():
        out["p4_" + n] = npy(p)                       # parameters after 4 AdamW steps
    # left-padded variant: rows whose causal window is all padding -> uniform softmax
    cfg2, m2 = build(V, H, L, nh)
    am2 = torch.ones(B, S, dtype=torch.long)
    am2[2, :5] = 0
    am2[0, 13:] = 0
    rec2, first2 = run_steps(m2, ids, am2, 1)
    out.update(lp_mask=npy(am2), lp_loss=npy(first2["loss"]), lp_logits=npy(first2["logits"]),
               lp_gnorm=np.array(rec2[0][1]))
    for n in ("bloom.word_embeddings.weight", "bloom.blocks.0.self_attention.query_key_value.weight",
              "bloom.blocks.1.mlp.dense_4h_to_h.bias"):
        out["lp_g_" + n] = npy(first2["grads"][n])
    # greedy decode (generation_util.py:57-119, do_sample=False) with right-aligned prompts
    cfg3, m3 = build(V, H, L, nh)
    m3.eval()
    p_ids = ids[:3, :6].clone()
    p_am = torch.ones(3, 6, dtype=torch.long)
    p_am[1, :2] = 0                                    # left padding, as inference_bloom.py:79 uses
    gen = m3.generate(p_ids, attention_mask=p_am,
                      generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    out.update(greedy_prompt=npy(p_ids), greedy_mask=npy(p_am), greedy_out=npy(gen))
    np.savez_compressed(os.path.join(HERE, "tiny_bloom.npz"), **out)
    print("tiny: sha", sha[:16], "ids", ids_sha[:16], "traj", rec)
    return sha, ids_sha


def gen_c1():
    V, H, L, nh, B, S = 250880, 1024, 2, 16, 2, 128
    cfg, m = build(V, H, L, nh)
    sha = weights_sha(m)
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(7))
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 100:] = 0
    ids_sha = hashlib.sha256(ids.numpy().tobytes()).hexdigest()
    rec, first = run_steps(m, ids, am, 4)
    lg = first["logits"]
    per_param = {n: float(g.double().pow(2).sum().sqrt()) for n, g in first["grads"].items()}
    doc = dict(cfg=dict(V=V, H=H, L=L, nh=nh, B=B, S=S, pad_row=1, pad_from=100),
               weights_sha256=sha, ids_sha256=ids_sha,
               traj=[[float(a), float(b)] for a, b in rec],
               argmax=lg.argmax(-1).tolist(),
               logits_first8=lg[:, :, :8].tolist(),
               lm_head_grad_probe=first["grads"]["bloom.word_embeddings.weight"][100:110, 100:110].tolist(),
               per_param_grad_norm=per_param,
               hidden_first4=first["hidden"][:, :, :4].tolist())
    json.dump(doc, open(os.path.join(HERE, "c1_bloom.json"), "w"))
    print("c1: sha", sha[:16], "ids", ids_sha[:16], "traj", rec)


def gen_c5():
    """Bloom-7B1 geometry (BASELINE configs[4]: H=4096, nh=32 -> hd=128, V=250880), one layer, B=1, S=512: the head-dim-128
    attention tiles and the 4096-wide GEMM / LayerNorm paths against the reference itself (scalars + slices only)."""
    V, H, L, nh, B, S = 250880, 4096, 1, 32, 1, 512
    cfg, m = build(V, H, L, nh)
    sha = weights_sha(m)
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(7))
    am = torch.ones(B, S, dtype=torch.long)
    am[0, 400:] = 0
    ids_sha = hashlib.sha256(ids.numpy().tobytes()).hexdigest()
    rec, first = run_steps(m, ids, am, 2)
    lg = first["logits"]
    per_param = {n: float(g.double().pow(2).sum().sqrt()) for n, g in first["grads"].items()}
    doc = dict(cfg=dict(V=V, H=H, L=L, nh=nh, B=B, S=S, pad_row=0, pad_from=400),
               weights_sha256=sha, ids_sha256=ids_sha,
               traj=[[float(a), float(b)] for a, b in rec],
               argmax=lg.argmax(-1).tolist(),
               logits_first8=lg[:, ::16, :8].tolist(),
               lm_head_grad_probe=first["grads"]["bloom.word_embeddings.weight"][100:110, 100:110].tolist(),
               qkv_grad_probe=first["grads"]["bloom.blocks.0.self_attention.query_key_value.weight"][:6, :6].tolist(),
               per_param_grad_norm=per_param,
               hidden_first4=first["hidden"][:, ::16, :4].tolist())
    json.dump(doc, open(os.path.join(HERE, "c5_bloom.json"), "w"))
    print("c5: sha", sha[:16], "ids", ids_sha[:16], "traj", rec)


def gen_gpt():
    """Tiny GPT-2 (pre-LN) and GPT-1 (post-LN) from the reference's modeling_gpt.py: logits, every gradient of the shifted
    cross-entropy, a 3-step torch.optim.AdamW trajectory, parameters after it, greedy tokens (KV cache)."""
    from CleanTransformer.models import modeling_gpt as rg
    V, H, L, nh, P, B, S = 173, 64, 2, 4, 64, 3, 24

    def build_gpt(version):
        cfg = rg.GPTConfig(vocab_size=V, n_embd=H, n_positions=P, n_layer=L, n_head=nh, n_ctx=P,
                           embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
        m = rg.GPTLMHeadModel(cfg, version=version)
        with torch.no_grad():
            for i, (name, prm) in enumerate(m.named_parameters()):
                r = torch.randn(prm.shape, generator=torch.Generator().manual_seed(1000 + i))
                if prm.dim() > 1:
                    prm.copy_(r * 0.02)
                elif ("norm" in name or "ln_f" in name) and name.endswith("weight"):
                    prm.copy_(1 + 0.1 * r)
                else:
                    prm.copy_(0.02 * r)
        for blk in m.gpt.blocks:
            blk.mlp[3].p = 0.0                              # the reference's trailing torch.nn.Dropout() (p = 0.5) neutralised
        return cfg, m.train()

    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(7))
    am = torch.ones(B, S, dtype=torch.long)
    am[1, 18:] = 0                                          # right padding
    ce = torch.nn.CrossEntropyLoss()
    out = dict(cfg=np.array([V, H, L, nh, P, B, S]), ids=npy(ids), mask=npy(am))
    for version in ("gpt2", "gpt"):
        cfg, m = build_gpt(version)
        out[f"{version}_names"] = np.array([n for n, _ in m.named_parameters()])
        opt = torch.optim.AdamW(m.parameters(), lr=1e-5)
        rec = []
        for t in range(3):
            (logits, hidden), _ = m(ids, attention_mask=am)
            loss = ce(logits[:, :-1, :].reshape(-1, V), ids[:, 1:].reshape(-1))
            opt.zero_grad()
            loss.backward()
            gn = gnorm(m)
            if t == 0:
                out[f"{version}_logits0"], out[f"{version}_hidden0"], out[f"{version}_loss0"] = npy(logits), npy(hidden), npy(loss)
                for n, prm in m.named_parameters():
                    out[f"{version}_g0_" + n] = npy(prm.grad)
            opt.step()
            rec.append((float(loss), gn))
        out[f"{version}_traj"] = np.array(rec, dtype=np.float64)
        for n, prm in m.named_parameters():
            out[f"{version}_p3_" + n] = npy(prm)
        print("gpt", version, "traj", rec)
    cfg, m = build_gpt("gpt2")
    m.eval()
    p_ids = ids[:2, :7].clone()
    gen = m.generate(p_ids, attention_mask=torch.ones(2, 7, dtype=torch.long),
                     generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    out.update(greedy_prompt=npy(p_ids), greedy_out=npy(gen))
    np.savez_compressed(os.path.join(HERE, "tiny_gpt.npz"), **out)


# ---------------------------------------------------------------------------------------
def gen_gpt_leftpad():
    """LEFT-padded tiny GPT-2 batch through the reference's modeling_gpt.py.  Rows whose whole causal window is padding are the one
    place the reference attends to the FUTURE (``w*b - 1e4*(1-b)`` leaves -1e4 on future keys, above the finfo.min of the padded
    visible ones: modeling_gpt.py:88-93): logits, loss and every gradient of that case."""
    from CleanTransformer.models import modeling_gpt as rg
    V, H, L, nh, P, B, S = 173, 64, 2, 4, 64, 3, 24
    cfg = rg.GPTConfig(vocab_size=V, n_embd=H, n_positions=P, n_layer=L, n_head=nh, n_ctx=P, embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    m = rg.GPTLMHeadModel(cfg, version="gpt2")
    with torch.no_grad():
        for i, (name, prm) in enumerate(m.named_parameters()):
            r = torch.randn(prm.shape, generator=torch.Generator().manual_seed(1000 + i))
            if prm.dim() > 1:
                prm.copy_(r * 0.02)
            elif ("norm" in name or "ln_f" in name) and name.endswith("weight"):
                prm.copy_(1 + 0.1 * r)
            else:
                prm.copy_(0.02 * r)
    for blk in m.gpt.blocks:
        blk.mlp[3].p = 0.0
    m.train()
    ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(7))
    am = torch.ones(B, S, dtype=torch.long)
    am[0, :5] = 0                                           # left padding
    am[2, :9] = 0
    am[2, 20:] = 0                                          # ... and a right-padded tail on the same row
    (logits, hidden), _ = m(ids, attention_mask=am)
    loss = torch.nn.CrossEntropyLoss()(logits[:, :-1, :].reshape(-1, V), ids[:, 1:].reshape(-1))
    loss.backward()
    out = dict(cfg=np.array([V, H, L, nh, P, B, S]), ids=npy(ids), mask=npy(am), logits0=npy(logits), loss0=npy(loss), gnorm0=np.array(gnorm(m)))
    for n, prm in m.named_parameters():
        out["g0_" + n] = npy(prm.grad)
    np.savez_compressed(os.path.join(HERE, "tiny_gpt_leftpad.npz"), **out)
    print("gpt leftpad loss", float(loss), "gnorm", gnorm(m))


def gen_soft_ce():
    """Probability-target branch of the reference's CrossEntropyLoss (loss.py:43-46): losses and input gradients for normalised and
    un-normalised targets, both reductions, plus the inputs of the reference's own printed self-check (seed 999, loss.py:76-91)."""
    g = torch.Generator().manual_seed(41)
    out = {}
    x = torch.randn(7, 37, generator=g) * 2
    tn = torch.softmax(torch.randn(7, 37, generator=g), dim=-1)
    tu = torch.rand(7, 37, generator=g)
    out.update(x=npy(x), t_norm=npy(tn), t_raw=npy(tu))
    for name, t in (("norm", tn), ("raw", tu)):
        for red in ("mean", "sum"):
            xi = x.clone().requires_grad_(True)
            loss = ref_loss.CrossEntropyLoss(red)(xi, t)
            loss.backward()
            out[f"loss_{name}_{red}"] = npy(loss)
            out[f"dx_{name}_{red}"] = npy(xi.grad)
    torch.manual_seed(999)
    pred, gtp = torch.rand(3, 4), torch.rand(3, 4)
    out.update(known_pred=npy(pred), known_t=npy(gtp), known_loss=npy(ref_loss.CrossEntropyLoss('mean')(pred, gtp)))
    np.savez_compressed(os.path.join(HERE, "soft_ce.npz"), **out)
    print("soft_ce:", {k: float(v) for k, v in out.items() if k.startswith("loss") or k == "known_loss"})


# ---------------------------------------------------------------------------------------
def gen_decode():
    """SURVEY §8(f)3 — the rest of the decode path: beam search (generation_util.py:121-290, do_sample=False so that token ids
    are deterministic), greedy with the n-gram ban, and the four logits processors (logits_processor.py) on seeded scores."""
    from CleanTransformer.generation import logits_processor as lp
    from CleanTransformer.models import modeling_gpt as rg
    out = {}
    # --- logits processors: data in / data out
    g = torch.Generator().manual_seed(31)
    sc = torch.randn(5, 97, generator=g) * 3.0
    sc[1, 10] = sc[1, 20]                                   # an exact tie inside the row
    hist = torch.randint(0, 9, (5, 14), generator=g)
    out["lp_scores"], out["lp_hist"] = npy(sc), npy(hist)
    for n in (2, 3):
        out[f"lp_ngram{n}"] = npy(lp.NoRepeatNGramLogitsProcessor(n)(hist, sc.clone()))
    out["lp_temp"] = npy(lp.TemperatureLogitsWrapper(0.7)(hist, sc.clone()))
    out["lp_temp_floor"] = npy(lp.TemperatureLogitsWrapper(0.0)(hist, sc.clone()))
    for k in (1, 10, 500):
        out[f"lp_topk{k}"] = npy(lp.TopKLogitsWrapper(k)(hist, sc.clone()))
    for pp in (0.3, 0.8, 1.0, 0.0):
        out[f"lp_topp{pp}"] = npy(lp.TopPLogitsWrapper(pp)(hist, sc.clone()))
    # --- Bloom beam search / n-gram greedy
    V, H, L, nh = 211, 64, 2, 8
    ids = torch.randint(0, V, (4, 16), generator=torch.Generator().manual_seed(7))
    cfg, m = build(V, H, L, nh)
    m.eval()
    p_ids = ids[:3, :6].clone()
    p_am = torch.ones(3, 6, dtype=torch.long)
    p_am[1, :2] = 0
    out["bloom_prompt"], out["bloom_mask"] = npy(p_ids), npy(p_am)
    free = m.generate(p_ids, attention_mask=p_am,
                      generation_configs=dict(beam_size=3, max_gen_len=6, do_sample=False, end_ids=[V + 5], pad_id=3))
    out["bloom_beam3_free"] = npy(free)                      # an end id that can never be produced: pure beam expansion
    # end ids taken from what the free run generated, so candidates do finish (both early_stop settings)
    ends = sorted(set(int(t) for t in free[:, 0, 7:9].reshape(-1)))
    out["bloom_ends"] = np.array(ends)
    for es in (True, False):
        r = m.generate(p_ids, attention_mask=p_am,
                       generation_configs=dict(beam_size=3, max_gen_len=6, do_sample=False, end_ids=ends, pad_id=3, early_stop=es))
        out[f"bloom_beam3_ends_es{int(es)}"] = npy(r)
    r = m.generate(p_ids, attention_mask=p_am,
                   generation_configs=dict(beam_size=2, max_gen_len=8, do_sample=False, end_ids=[V + 5], pad_id=3, no_repeat_ngram_size=2))
    out["bloom_beam2_ngram2"] = npy(r)
    rep = torch.tensor([[5, 9, 5, 9, 5, 9], [7, 7, 7, 7, 7, 7]])
    out["bloom_rep_prompt"] = npy(rep)
    for n in (0, 2):
        r = m.generate(rep, attention_mask=torch.ones_like(rep),
                       generation_configs=dict(beam_size=1, max_gen_len=8, do_sample=False, end_ids=None, pad_id=3, no_repeat_ngram_size=n))
        out[f"bloom_greedy_ngram{n}"] = npy(r)
    # --- GPT-2 beam search
    Vg, Hg, Lg, nhg, P = 173, 64, 2, 4, 64
    cfgg = rg.GPTConfig(vocab_size=Vg, n_embd=Hg, n_positions=P, n_layer=Lg, n_head=nhg, n_ctx=P, embd_pdrop=0.0, attn_pdrop=0.0,
                        resid_pdrop=0.0)
    mg = rg.GPTLMHeadModel(cfgg, version="gpt2")
    with torch.no_grad():
        for i, (name, prm) in enumerate(mg.named_parameters()):
            r = torch.randn(prm.shape, generator=torch.Generator().manual_seed(1000 + i))
            if prm.dim() > 1:
                prm.copy_(r * 0.02)
            elif ("norm" in name or "ln_f" in name) and name.endswith("weight"):
                prm.copy_(1 + 0.1 * r)
            else:
                prm.copy_(0.02 * r)
    mg.eval()
    gids = torch.randint(0, Vg, (3, 24), generator=torch.Generator().manual_seed(7))[:2, :7].clone()
    out["gpt_prompt"] = npy(gids)
    free = mg.generate(gids, attention_mask=torch.ones(2, 7, dtype=torch.long),
                       generation_configs=dict(beam_size=4, max_gen_len=6, do_sample=False, end_ids=[Vg + 1], pad_id=3))
    out["gpt_beam4_free"] = npy(free)
    ends = sorted(set(int(t) for t in free[:, 0, 8:10].reshape(-1)))
    out["gpt_ends"] = np.array(ends)
    for es in (True, False):
        r = mg.generate(gids, attention_mask=torch.ones(2, 7, dtype=torch.long),
                        generation_configs=dict(beam_size=4, max_gen_len=6, do_sample=False, end_ids=ends, pad_id=3, early_stop=es))
        out[f"gpt_beam4_ends_es{int(es)}"] = npy(r)
    # explicit position_ids / segment_ids (GPT only: modeling_gpt.py:164,184): both searches carry and extend them (:98-99, :268-271)
    gpos = torch.tensor([[3, 4, 5, 6, 7, 8, 9], [0, 1, 2, 3, 4, 5, 6]])
    gseg = torch.tensor([[1, 1, 1, 2, 2, 2, 2], [5, 5, 5, 5, 5, 5, 5]])
    out["gpt_pos"], out["gpt_seg"] = npy(gpos), npy(gseg)
    r = mg.generate(gids, attention_mask=torch.ones(2, 7, dtype=torch.long), position_ids=gpos, segment_ids=gseg,
                    generation_configs=dict(beam_size=1, max_gen_len=6, do_sample=False, end_ids=None, pad_id=3))
    out["gpt_greedy_posseg"] = npy(r)
    r = mg.generate(gids, attention_mask=torch.ones(2, 7, dtype=torch.long), position_ids=gpos, segment_ids=gseg,
                    generation_configs=dict(beam_size=3, max_gen_len=6, do_sample=False, end_ids=[Vg + 1], pad_id=3))
    out["gpt_beam3_posseg"] = npy(r)
    np.savez_compressed(os.path.join(HERE, "decode.npz"), **out)
    for k in ("gpt_greedy_posseg", "gpt_beam3_posseg", "bloom_beam3_free", "bloom_ends", "bloom_beam3_ends_es1", "bloom_beam3_ends_es0", "bloom_greedy_ngram0", "bloom_greedy_ngram2",
              "gpt_beam4_free", "gpt_ends", "gpt_beam4_ends_es1"):
        print(k, out[k].tolist())


# ---------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, ret):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    cfg, m = build(V, H, L, nh)
    if rank != 0:                                      # DDP must broadcast rank-0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    ddp = DDP(m)
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B]
    am = torch.ones(B, S, dtype=torch.long)
    if rank == 1:
        am[0, 11:] = 0
    (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
    loss.backward()
    if rank == 0:
        ret.update({n: npy(p.grad) for n, p in m.named_parameters()})
        ret["__loss0"] = npy(loss)
    dist.destroy_process_group()


def gen_ddp():
    import torch.multiprocessing as mp
    out = {}
    for world, port in ((2, 29611), (4, 29612)):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_ddp_worker, args=(world, port, ret), nprocs=world, join=True)
        for k, v in ret.items():
            out[f"w{world}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "ddp_tiny.npz"), **out)
    print("ddp_tiny.npz:", len(out), "arrays")


def gen_known():
    """The reference's own printed self-check values (seed 999; loss.py:76-100, transformer.py:134-141)."""
    vals = {}
    torch.manual_seed(999)
    pred, gt = torch.rand(3, 4), torch.randint(0, 4, (3,))
    vals["ce_index"] = float(ref_loss.CrossEntropyLoss('mean')(pred, gt))
    vals["ce_index_official"] = float(torch.nn.CrossEntropyLoss()(pred, gt))
    vals["nll"] = float(ref_loss.NLLLoss('mean')(pred, gt))
    torch.manual_seed(999)
    pred, gtp = torch.rand(3, 4), torch.rand(3, 4)
    vals["mse"] = float(ref_loss.MSELoss('mean')(pred, gtp))
    vals["ce_prob"] = float(ref_loss.CrossEntropyLoss('mean')(pred, gtp))
    torch.manual_seed(999)
    x = torch.rand((3, 4, 6))
    vals["ln_46"] = RefLayerNorm([4, 6])(x).tolist()
    json.dump(vals, open(os.path.join(HERE, "known_answers.json"), "w"))
    print("known:", {k: v for k, v in vals.items() if k != "ln_46"})


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["ops", "tiny", "c1", "c5", "gpt", "soft_ce", "decode", "ddp", "known"]
    if "ops" in which:
        gen_ops()
    if "tiny" in which:
        gen_tiny()
    if "c1" in which:
        gen_c1()
    if "c5" in which:
        gen_c5()
    if "gpt" in which:
        gen_gpt()
    if "gpt_leftpad" in which:
        gen_gpt_leftpad()
    if "soft_ce" in which:
        gen_soft_ce()
    if "decode" in which:
        gen_decode()
    if "ddp" in which:
        gen_ddp()
    if "known" in which:
        gen_known()
