"""Single-device SFT loop with the reference's step order (examples/ft_bloom.py:65-97):
forward -> optimizer.zero_grad() -> loss.backward() -> optimizer.step()."""
from __future__ import annotations

import os

import torch

from ..optimizer import AdamW


def collate(batch_texts, tokenizer, max_length, pad_to_max=False):
    """BelleDataset.collate_fn semantics (ft_bloom.py:41-55): right-pad to the longest sample, labels = input_ids
    clone INCLUDING pads, mask 1/0."""
    texts = [t + tokenizer.eos_token for t in batch_texts]
    enc = tokenizer(texts, truncation=True, padding=True, max_length=max_length, return_tensors="pt")
    out = {**enc}
    if pad_to_max and out["input_ids"].shape[-1] < max_length:
        n = max_length - out["input_ids"].shape[-1]
        out["input_ids"] = torch.cat([out["input_ids"], torch.full((len(texts), n), tokenizer.pad_token_id)], dim=-1)
        out["attention_mask"] = torch.cat([out["attention_mask"], torch.zeros(len(texts), n, dtype=out["attention_mask"].dtype)], dim=-1)
    out["labels"] = out["input_ids"].clone()
    out["prompts"] = texts
    return out


def train_step(model, batch, optimizer):
    """One SFT step (ft_bloom.py:84-90).  Returns the loss tensor (device scalar, no sync)."""
    outputs, _ = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
    loss = outputs[0]
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss


def train_step_amp(model, batch, optimizer, scaler, amp_dtype=None):
    """The reference's ``use_torch_amp`` branch (ft_bloom_DDP.py:121-127): scaled backward, scaler.step, scaler.update.
    NOTE the reference calls no ``zero_grad()`` on this branch, so gradients accumulate from step to step (each step adds
    scale * g_t to the already-unscaled sum) — kept, because it decides the numbers a side-by-side run prints; pass
    ``zero_grad=True`` for the loop one actually wants."""
    from ..amp import autocast
    with autocast(dtype=amp_dtype):                                  # amp_dtype=torch.float16: the reference's precision (torch's default autocast dtype)
        outputs, _ = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"])
        loss = outputs[0]
    scaler.scale(loss).backward()
    scaler.step(optimizer)
    scaler.update()
    return loss


def train(model, train_loader, epoches, save_interval=1000, print_interval=10, save_dir="./", optimizer=None):
    device = torch.device("cuda:0")
    model = model.to(device)
    if optimizer is None:
        optimizer = AdamW(model.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)   # == torch.optim.AdamW(lr=1e-5)
    model.train()
    steps = 0
    os.makedirs(save_dir, exist_ok=True)
    for _ in range(epoches):
        for batch in train_loader:
            batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            loss = train_step(model, batch, optimizer)
            steps += 1
            if steps % print_interval == 0:
                print("step: {}, loss: {}".format(steps, loss.cpu().item()))
            if steps % save_interval == 0:
                torch.save(model.state_dict(), os.path.join(save_dir, f"model_step_{steps}.pt"))
    return steps
