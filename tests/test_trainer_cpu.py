"""CPU (no GPU): the HF-free Trainer subset — step loop with gradient accumulation, device-side clipping,
logging, checkpoint-N save / rotate / resume — through the torch-CPU emulation of the kernel contracts.  The reference's
trainer.py does not import in this image (accelerate / transformers drift), so what is pinned is what it delegates
to: the golden loss / grad-norm trajectory of the reference model + torch.optim.AdamW, ``torch.nn.utils.clip_grad_norm_``, and
the on-disk layout / rotation / resume rules restated from the file."""
import json
import os

import numpy as np
import pytest
import torch

import cpu_kernel_emulation as emu
from test_host_logic_cpu import TINY, T, build, close

V, H, L, NH, B, S = [int(v) for v in TINY["cfg"]]


class Batches:
    """A ready 'dataloader': iterable with a length, no __getitem__ (the Trainer then uses it as is)."""

    def __init__(self, items):
        self.items = items

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def golden_batch():
    return {"input_ids": T(TINY["ids"]), "attention_mask": T(TINY["mask"]), "labels": T(TINY["ids"]).clone(), "prompts": ["a", "b"]}


def rand_batches(n, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ids = torch.randint(0, V, (2, 12), generator=g)
        out.append({"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids.clone()})
    return out


def make(args, data, callbacks=None, monkeypatch=None):
    from cleantransformer_amd.trainer import Trainer
    return Trainer(model=build(V, H, L, NH), args=args, train_dataset=Batches(data), callbacks=callbacks)


def test_trajectory_logging_and_grad_norm_match_reference_golden(monkeypatch, tmp_path):
    emu.install(monkeypatch)
    from cleantransformer_amd.trainer import TrainingArguments
    args = TrainingArguments(output_dir=str(tmp_path), device="cpu", max_steps=4, learning_rate=1e-5, weight_decay=0.01,
                             lr_scheduler_type="constant", max_grad_norm=1e9, logging_steps=1, save_strategy="no",
                             per_device_train_batch_size=B)
    tr = make(args, [golden_batch()] * 4)
    out = tr.train()
    assert out.global_step == 4 and len(tr.state.log_history) == 4
    for t, rec in enumerate(tr.state.log_history):
        assert rec["step"] == t + 1 and rec["learning_rate"] == 1e-5 and rec["epoch"] == (t + 1) / 4
        assert abs(rec["loss"] - float(TINY["traj"][t, 0])) <= 1.01e-4 and rec["loss"] == round(rec["loss"], 4)   # rounded to 4 places
        assert abs(rec["grad_norm"] - TINY["traj"][t, 1]) <= 1e-4 * TINY["traj"][t, 1]          # global L2 norm BEFORE clipping
    assert abs(out.training_loss - float(np.mean(TINY["traj"][:, 0]))) < 1e-5
    for n, p in tr.model.named_parameters():
        close(p, TINY["p4_" + n], 1e-5, 1e-7)


def test_clip_matches_torch_clip_grad_norm(monkeypatch):
    emu.install(monkeypatch)
    from cleantransformer_amd.trainer import clip_grad_norm_
    m = build(V, H, L, NH)
    b = golden_batch()
    (loss, _, _), _ = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
    loss.backward()
    ref = [p.grad.clone() for p in m.parameters()]
    holders = [torch.nn.Parameter(torch.zeros_like(g)) for g in ref]
    for h, g in zip(holders, ref):
        h.grad = g.clone()
    want = torch.nn.utils.clip_grad_norm_(holders, 0.5)
    got = clip_grad_norm_(list(m.parameters()), 0.5)
    assert abs(float(got) - float(want)) <= 1e-6 * float(want) and abs(float(got) - TINY["traj"][0, 1]) <= 1e-4 * float(got)
    for p, h in zip(m.parameters(), holders):
        close(p.grad, h.grad, 1e-6, 1e-12)
    again = clip_grad_norm_(list(m.parameters()), 0.5)                        # on the ball already: coef = 0.5 / (0.5 + 1e-6), as torch
    assert abs(float(again) - 0.5) < 1e-5
    for p, h in zip(m.parameters(), holders):
        close(p.grad, h.grad, 1e-5, 1e-12)


def test_gradient_accumulation_equals_the_full_batch_step(monkeypatch, tmp_path):
    """Two half-batches with gradient_accumulation_steps=2 == one step on the full batch (equal token counts), i.e. the
    golden first step — the reference's per-micro-batch zero_grad (trainer.py:468) is deliberately not reproduced."""
    emu.install(monkeypatch)
    from cleantransformer_amd.trainer import TrainingArguments
    full = golden_batch()
    halves = [{k: (v[:2] if torch.is_tensor(v) else v) for k, v in full.items()}, {k: (v[2:] if torch.is_tensor(v) else v) for k, v in full.items()}]
    common = dict(device="cpu", learning_rate=1e-5, weight_decay=0.01, lr_scheduler_type="constant", max_grad_norm=None, logging_steps=1,
                  save_strategy="no")
    a = make(TrainingArguments(output_dir=str(tmp_path / "a"), max_steps=1, gradient_accumulation_steps=2, per_device_train_batch_size=2, **common), halves)
    passes = emu.SCALE_IF_PASSES[0]
    a.train()
    # round 4: training_step announces 1/ga to the fused loss, so the backward of `loss / ga` finds its factor already in dlogits
    assert emu.SCALE_IF_PASSES[0] == passes, "the accumulation micro-steps must not rescale dlogits in a second pass"
    b = make(TrainingArguments(output_dir=str(tmp_path / "b"), max_steps=1, per_device_train_batch_size=4, **common), [full])
    b.train()
    assert abs(a.state.log_history[0]["loss"] - b.state.log_history[0]["loss"]) <= 1.01e-4
    assert abs(b.state.log_history[0]["loss"] - float(TINY["traj"][0, 0])) <= 1.01e-4
    for (n, pa), (_, pb) in zip(a.model.named_parameters(), b.model.named_parameters()):
        close(pa, pb, 1e-6, 5e-8)            # one Adam step of lr 1e-5: m/sqrt(v) amplifies fp32 summation-order noise
    assert a.state.global_step == 1 and a.state.epoch == 1.0


def test_checkpoint_layout_rotation_and_bit_exact_resume(monkeypatch, tmp_path):
    emu.install(monkeypatch)
    from cleantransformer_amd.trainer import Trainer, TrainerCallback, TrainingArguments, get_last_checkpoint
    data = rand_batches(3)

    def args(out):
        return TrainingArguments(output_dir=str(out), device="cpu", num_train_epochs=2, learning_rate=1e-3, weight_decay=0.01, warmup_steps=2,
                                 max_grad_norm=1.0, logging_steps=1, save_steps=2, save_total_limit=2, per_device_train_batch_size=2)
    full = make(args(tmp_path / "full"), data)
    full.train()
    assert full.state.global_step == 6
    assert sorted(os.listdir(tmp_path / "full")) == ["checkpoint-4", "checkpoint-6"]            # checkpoint-2 rotated out
    assert sorted(os.listdir(tmp_path / "full" / "checkpoint-6")) == sorted(
        ["pytorch_model.bin", "optimizer.pt", "scheduler.pt", "trainer_state.json", "rng_state.pth", "training_args.bin"])
    st = json.load(open(tmp_path / "full" / "checkpoint-6" / "trainer_state.json"))
    assert st["global_step"] == 6 and st["max_steps"] == 6 and len(st["log_history"]) == 6 and st["train_batch_size"] == 2
    lrs = [r["learning_rate"] for r in st["log_history"]]                                        # warm-up 2, then linear decay to 0
    assert lrs == pytest.approx([0.5e-3, 1e-3, 0.75e-3, 0.5e-3, 0.25e-3, 0.0])

    class StopAt(TrainerCallback):
        def on_step_end(self, args, state, control, **kw):
            if state.global_step == 4:
                control.should_training_stop = True

    part = make(args(tmp_path / "part"), data, callbacks=[StopAt()])
    part.train()
    assert part.state.global_step == 4 and get_last_checkpoint(str(tmp_path / "part")).endswith("checkpoint-4")
    resumed = Trainer(model=build(V, H, L, NH), args=args(tmp_path / "part"), train_dataset=Batches(data))
    resumed.train(resume_from_checkpoint=True)                 # epoch 1 resumes after its first batch (4 = 1 epoch + 1 step)
    assert resumed.state.global_step == 6
    assert resumed.state.log_history == full.state.log_history
    for (n, pa), (_, pb) in zip(full.model.named_parameters(), resumed.model.named_parameters()):
        assert torch.equal(pa, pb), n
    assert resumed.optimizer.steps[0] == full.optimizer.steps[0] == 7
    with pytest.raises(ValueError):
        Trainer(model=build(V, H, L, NH), args=args(tmp_path / "none"), train_dataset=Batches(data)).train(resume_from_checkpoint=True)


def test_safetensors_checkpoint_and_dataset_path(monkeypatch, tmp_path):
    emu.install(monkeypatch)
    from cleantransformer_amd.trainer import Trainer, TrainingArguments
    g = torch.Generator().manual_seed(9)
    samples = [{"input_ids": torch.randint(0, V, (int(n),), generator=g).tolist()} for n in (5, 9, 7, 6)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(samples)

        def __getitem__(self, i):
            return samples[i]

    def coll(items):                                       # right-pad to the longest, labels = clone, mask 1/0 (ft_bloom.py:41-55)
        n = max(len(s["input_ids"]) for s in items)
        ids = torch.tensor([s["input_ids"] + [0] * (n - len(s["input_ids"])) for s in items])
        am = torch.tensor([[1] * len(s["input_ids"]) + [0] * (n - len(s["input_ids"])) for s in items])
        return {"input_ids": ids, "attention_mask": am, "labels": ids.clone()}
    args = TrainingArguments(output_dir=str(tmp_path), device="cpu", num_train_epochs=1, per_device_train_batch_size=2, save_steps=2,
                             save_safetensors=True, save_only_model=True, logging_steps=1, include_num_input_tokens_seen=True)
    tr = Trainer(model=build(V, H, L, NH), args=args, train_dataset=DS(), data_collator=coll)
    tr.train()
    assert tr.state.global_step == 2 and tr.state.num_input_tokens_seen > 0
    assert sorted(os.listdir(tmp_path / "checkpoint-2")) == ["model.safetensors", "trainer_state.json", "training_args.bin"]
    fresh = Trainer(model=build(V, H, L, NH), args=args, train_dataset=DS(), data_collator=coll)
    fresh._load_from_checkpoint(str(tmp_path / "checkpoint-2"))
    for (n, pa), (_, pb) in zip(tr.model.named_parameters(), fresh.model.named_parameters()):
        assert torch.equal(pa, pb), n
    assert fresh.model.lm_head.weight is fresh.model.bloom.word_embeddings.weight


def test_short_last_window_does_not_leak_applied_gradients(monkeypatch, tmp_path):
    """gradient_accumulation_steps=4 with 3 batches per epoch: every epoch ends in a SHORT window (steps_in_epoch <= ga), so
    optimizer steps do not fall on multiples of ga.  Gradients that were already applied must not be accumulated into the next
    step (round-1 advisor finding): the run must equal a hand-written loop that zeroes after every optimizer step."""
    emu.install(monkeypatch)
    from cleantransformer_amd.optimizer import AdamW
    from cleantransformer_amd.trainer import TrainingArguments
    data = rand_batches(3, seed=11)
    args = TrainingArguments(output_dir=str(tmp_path), device="cpu", num_train_epochs=2, gradient_accumulation_steps=4, learning_rate=1e-3,
                             weight_decay=0.0, lr_scheduler_type="constant", max_grad_norm=None, logging_steps=1, save_strategy="no",
                             per_device_train_batch_size=2)
    tr = make(args, data)
    tr.train()
    assert tr.state.global_step == 2                                    # one (short) window per epoch
    ref = build(V, H, L, NH)
    opt = AdamW(ref.parameters(), lr=1e-3, weight_decay=0.0, decoupled=True)
    # window 1 = the three batches of epoch 1 (short last window); window 2 closes at total_batched_samples == 4, i.e. after the
    # FIRST batch of epoch 2 (the reference loop's rule, trainer.py:468-480 — kept) and must contain that batch's gradient only
    for window in (data, data[:1]):
        opt.zero_grad()
        for b in window:
            (loss, _, _), _ = ref(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
            (loss / 4).backward()
        opt.step()
    for (n, pa), (_, pb) in zip(tr.model.named_parameters(), ref.named_parameters()):
        close(pa, pb, 1e-6, 1e-7)
