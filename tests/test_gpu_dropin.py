"""GPU: the drop-in import paths.  A caller written against the reference imports `CleanTransformer.models.modeling_bloom`,
`CleanTransformer.optimizer`, `CleanTransformer.loss` ... (ft_bloom.py:12-20; the reference has no __init__.py, the package is found from the
repo root).  These tests touch NOTHING under `cleantransformer_amd` by name: they build Bloom through `CleanTransformer.*`, run the
ft_bloom.py:79-90 loop body verbatim (model(**batch) -> outputs[0] -> zero_grad / backward / step) and compare with the trajectory the
REFERENCE produced on the same inputs (tests/golden/tiny_bloom.npz, made by tests/golden/make_golden.py importing /root/reference)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
TINY = np.load(os.path.join(HERE, "golden", "tiny_bloom.npz"))


def _T(a):
    return torch.from_numpy(np.asarray(a))


def _reference_style_model():
    # ft_bloom.py:100-106: config -> BloomForCausalLM(config); the golden run started from these weights (state_dict keys are the reference's)
    from CleanTransformer.models.modeling_bloom import BloomConfig, BloomForCausalLM
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    config = BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh)
    model = BloomForCausalLM(config)
    from oracle import bloom_ref as R                                        # det_init: the deterministic weights make_golden.py loaded into the reference
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model._tie_weight()
    return model


@pytest.mark.parametrize("opt_kind", ["torch.optim.AdamW", "CleanTransformer.optimizer.AdamW"])
def test_ft_bloom_loop_through_the_reference_import_paths(opt_kind):
    """ft_bloom.py:65-90 with the reference's names only.  torch.optim.AdamW(lr=1e-5) is what ft_bloom.py:70 constructs; the repo's own
    optimizer.py AdamW (L2 form) is the other optimizer a reference caller can name — its trajectory is checked for descent and for
    agreement with the torch one at lr = 1e-5 (the two decay forms differ by O(lr * wd))."""
    device = torch.device("cuda:0")
    model = _reference_style_model().to(device)
    if opt_kind == "torch.optim.AdamW":
        from torch.optim import AdamW
        optimizer = AdamW(model.parameters(), lr=1e-5)
    else:
        from CleanTransformer.optimizer import AdamW
        optimizer = AdamW(model.parameters(), lr=1e-5)
    model.train()
    batch = {"input_ids": _T(TINY["ids"]), "attention_mask": _T(TINY["mask"]), "labels": _T(TINY["ids"]).clone()}
    losses = []
    for t in range(4):
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                batch[k] = v.to(device)
        outputs, _ = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        loss = outputs[0]
        optimizer.zero_grad()
        loss.backward()
        gn = math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in model.parameters()))
        optimizer.step()
        losses.append(loss.cpu().item())
        if opt_kind == "torch.optim.AdamW":
            assert abs(losses[-1] - TINY["traj"][t, 0]) <= 1e-5 * TINY["traj"][t, 0], (t, losses[-1])
            assert abs(gn - TINY["traj"][t, 1]) <= 1e-4 * gn, (t, gn)
        else:
            assert abs(losses[-1] - TINY["traj"][t, 0]) <= 2e-4 * TINY["traj"][t, 0], (t, losses[-1])
    assert losses[-1] < losses[0]
    if opt_kind == "torch.optim.AdamW":
        for n, p in model.named_parameters():
            ref = _T(TINY["p4_" + n]).double()
            err = (p.detach().double().cpu() - ref).abs()
            assert bool((err <= 1e-7 + 1e-5 * ref.abs()).all()), (n, float(err.max()))
    # the HIP library did the work: it is mapped into this process and the model's parameters live on the GPU
    with open("/proc/self/maps") as f:
        assert "libctmi355.so" in f.read()


def test_reference_loss_and_transformer_names_resolve_and_run_on_the_gpu():
    """loss.py:34-49 / transformer.py:71-89 through the reference's import paths, one call each against torch."""
    from CleanTransformer.loss import CrossEntropyLoss
    from CleanTransformer.transformer import LayerNorm
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 11, generator=g).cuda()
    y = torch.randint(0, 11, (6,), generator=g).cuda()
    got = CrossEntropyLoss()(x, y)
    assert abs(float(got) - float(torch.nn.functional.cross_entropy(x, y))) < 1e-5
    ln = LayerNorm(32).cuda()
    h = torch.randn(4, 5, 32, generator=g).cuda()
    assert torch.allclose(ln(h), torch.nn.functional.layer_norm(h, (32,), ln.weight, ln.bias, 1e-5), atol=1e-5)
