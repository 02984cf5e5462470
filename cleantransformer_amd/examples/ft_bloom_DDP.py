"""DDP SFT loop (examples/ft_bloom_DDP.py:79-156) on RCCL: one process per GPU, env RANK/LOCAL_RANK/WORLD_SIZE."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ..optimizer import AdamW
from ..trainer.ddp import DistributedDataParallel as DDP
from .ft_bloom import train_step, train_step_amp


def print_rank(value, rank_set=None):
    rank = int(os.environ.get("RANK", "0"))
    if rank_set is None or rank == rank_set:
        print(f"rank {rank}: {value}")


def train(model, train_loader, epoches, save_interval=1000, print_interval=10, save_dir="./", use_torch_amp=None, apex_level=None,
          optimizer=None, amp_zero_grad=False, amp_dtype=None):
    """ft_bloom_DDP.py:79-156.  ``use_torch_amp`` selects the GradScaler branch (:107-128) on this package's scaler; apex
    (:91-97) does not exist on ROCm builds of this stack and is refused, as the reference refuses invalid combinations."""
    if use_torch_amp and apex_level is not None:
        raise Exception("use_torch_amp and apex_level are mutually exclusive, use_torch_amp={}, apex_level={}".format(use_torch_amp, apex_level))
    if apex_level is not None:
        raise Exception("apex is not available in the MI355X build; use use_torch_amp=True or the model's compute_dtype='bf16'")
    local_rank = int(os.environ["LOCAL_RANK"])
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    model = DDP(model.to(device), device_ids=[local_rank])
    if optimizer is None:
        optimizer = AdamW(model.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    model.train()
    steps = 0
    os.makedirs(save_dir, exist_ok=True)
    scaler = None
    if use_torch_amp:
        from ..amp import GradScaler
        print_rank("using amp (GradScaler)...", rank_set=0)
        scaler = GradScaler()
    for epoch in range(epoches):
        if getattr(train_loader, "sampler", None) is not None and hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        for batch in train_loader:
            batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            if scaler is not None:
                if amp_zero_grad:
                    optimizer.zero_grad()
                loss = train_step_amp(model, batch, optimizer, scaler, amp_dtype)
            else:
                loss = train_step(model, batch, optimizer)
            steps += 1
            if steps == 1:                                           # the reference's cross-run parity probe (:145-150)
                print_rank("step{}: input_ids[0:2,10:20]={}".format(steps, batch["input_ids"][0:2, 10:20]))
                print_rank("step{}: lm_head.weigth.grad[100:110,100:110]={}".format(
                    steps, model.module.lm_head.weight.grad[100:110, 100:110]))
            if steps % print_interval == 0 and local_rank == 0:
                print_rank("step: {}, loss: {}".format(steps, loss.cpu().item()))
            if steps % save_interval == 0 and local_rank == 0:
                torch.save(model.state_dict(), os.path.join(save_dir, f"model_step_{steps}.pt"))
    return steps


def main_init():
    dist.init_process_group("nccl")                                   # "nccl" is RCCL on ROCm (ft_bloom_DDP.py:183)
