"""HF-free restatement of the subset of the reference's ``Trainer`` (CleanTransformer/trainer/trainer.py) that the SFT hot path
uses — SURVEY §8(f)4: the step loop with gradient accumulation (:468-504), gradient-norm clipping (:491-498), logging
(:1223-1243, :1287-1298), ``checkpoint-N`` save / rotate / resume (:1303-1343, :1465-1511, :349-399, :1516-1596) — over this
package's model, fused optimizers and DDP.  The reference file imports transformers / accelerate / peft internals that have
drifted (it no longer imports in this image), so the names kept are the ones a caller touches:
``TrainingArguments`` fields, ``Trainer(model, args, data_collator, train_dataset, optimizers=, callbacks=)``, ``train(
resume_from_checkpoint=)``, ``training_step``, ``compute_loss``, ``log``, ``state.log_history``, ``TrainOutput`` and the on-disk
checkpoint layout (``checkpoint-<step>/{pytorch_model.bin | model.safetensors, optimizer.pt, scheduler.pt, trainer_state.json,
rng_state.pth, training_args.bin}``).

MI355X-first pieces
  * clipping never leaves the device: ``ctmi_sumsq`` accumulates the global sum of squares in a device double, the clip
    coefficient is formed on the device and applied by ``ctmi_scale`` with a device-side multiplier; the norm is only read on
    the host when a log line is due;
  * accumulation micro-steps run under the DDP wrapper's ``no_sync()``: one bucketed all-reduce per optimizer step;
  * the running loss is a device scalar (as in the reference) — no per-step ``.item()``.

Decisions on reference behaviour (SURVEY Appendix B style)
  * trainer.py:468 calls ``model.zero_grad()`` before EVERY micro-batch, which discards all but the last micro-batch of an
    accumulation window; here gradients are cleared once per window (bug not replicated);
  * ``compute_loss`` (:558-586) takes ``outputs[0]``; this package's causal LMs return ``((loss, logits, hidden), k_v_pasts)``
    like the reference's Bloom, for which ``outputs[0]`` is a tuple — the loss is then its first element;
  * evaluation, label smoothing, NEFTune, DeepSpeed / FSDP / PEFT branches are out of scope (not on the path).
"""
from __future__ import annotations

import dataclasses
import json
import math
import os
import random
import re
import shutil
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..optimizer import AdamW
from .ddp import DistributedDataParallel

PREFIX_CHECKPOINT_DIR = "checkpoint"
WEIGHTS_NAME, SAFE_WEIGHTS_NAME = "pytorch_model.bin", "model.safetensors"
OPTIMIZER_NAME, SCHEDULER_NAME = "optimizer.pt", "scheduler.pt"
TRAINER_STATE_NAME, TRAINING_ARGS_NAME = "trainer_state.json", "training_args.bin"


@dataclass
class TrainingArguments:
    output_dir: str = "./output"
    per_device_train_batch_size: int = 8
    gradient_accumulation_steps: int = 1
    num_train_epochs: float = 3.0
    max_steps: int = -1
    learning_rate: float = 5e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: Optional[float] = 1.0
    lr_scheduler_type: str = "linear"                  # "linear" (warm-up then linear decay to 0) | "constant"
    warmup_steps: int = 0
    logging_steps: float = 500
    save_steps: float = 500
    save_strategy: str = "steps"                       # "steps" | "no"
    save_total_limit: Optional[int] = None
    save_only_model: bool = False
    save_safetensors: bool = False
    ignore_data_skip: bool = False
    include_num_input_tokens_seen: bool = False
    dataloader_drop_last: bool = False
    seed: int = 42
    disable_tqdm: bool = True
    device: Optional[str] = None                       # default: cuda:<LOCAL_RANK>

    @property
    def world_size(self) -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def process_index(self) -> int:
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def should_save(self) -> bool:
        return self.process_index == 0

    @property
    def train_batch_size(self) -> int:
        return self.per_device_train_batch_size


@dataclass
class TrainerState:
    """trainer_state.json — the fields of transformers' TrainerState the reference reads back (:366-378)."""
    epoch: Optional[float] = None
    global_step: int = 0
    max_steps: int = 0
    logging_steps: float = 500
    eval_steps: float = 500
    save_steps: float = 500
    train_batch_size: Optional[int] = None
    num_train_epochs: int = 0
    num_input_tokens_seen: int = 0
    total_flos: float = 0
    log_history: List[Dict[str, float]] = field(default_factory=list)
    best_metric: Optional[float] = None
    best_model_checkpoint: Optional[str] = None
    is_local_process_zero: bool = True
    is_world_process_zero: bool = True
    is_hyper_param_search: bool = False
    trial_name: Optional[str] = None
    trial_params: Optional[Dict[str, Any]] = None

    def save_to_json(self, json_path: str):
        with open(json_path, "w", encoding="utf-8") as f:
            f.write(json.dumps(dataclasses.asdict(self), indent=2, sort_keys=True) + "\n")

    @classmethod
    def load_from_json(cls, json_path: str):
        with open(json_path, "r", encoding="utf-8") as f:
            return cls(**json.loads(f.read()))


@dataclass
class TrainerControl:
    should_training_stop: bool = False
    should_epoch_stop: bool = False
    should_save: bool = False
    should_log: bool = False


class TrainerCallback:
    """Event sink with the (args, state, control, **kwargs) signature of transformers' callbacks; return None or the control."""

    def on_train_begin(self, args, state, control, **kw): pass
    def on_step_begin(self, args, state, control, **kw): pass
    def on_substep_end(self, args, state, control, **kw): pass
    def on_step_end(self, args, state, control, **kw): pass
    def on_epoch_begin(self, args, state, control, **kw): pass
    def on_epoch_end(self, args, state, control, **kw): pass
    def on_log(self, args, state, control, logs=None, **kw): pass
    def on_save(self, args, state, control, **kw): pass
    def on_train_end(self, args, state, control, **kw): pass


class TrainOutput(NamedTuple):
    global_step: int
    training_loss: float
    metrics: Optional[Dict[str, float]]


class _Schedule:
    """``get_scheduler("linear" | "constant")`` of the reference's create_scheduler (:854-865): multiplicative LR factor per
    optimizer step, with ``step() / get_last_lr() / state_dict()``."""

    def __init__(self, optimizer, kind: str, warmup: int, total: int):
        assert kind in ("linear", "constant"), kind
        self.optimizer, self.kind, self.warmup, self.total = optimizer, kind, int(warmup), int(total)
        self.base_lr = _get_lr(optimizer)
        self.last_epoch = 0
        _set_lr(optimizer, self.base_lr * self._factor(0))

    def _factor(self, s: int) -> float:
        if s < self.warmup:
            return s / max(1, self.warmup)
        if self.kind == "constant":
            return 1.0
        return max(0.0, (self.total - s) / max(1, self.total - self.warmup))

    def step(self):
        self.last_epoch += 1
        _set_lr(self.optimizer, self.base_lr * self._factor(self.last_epoch))

    def get_last_lr(self):
        return [_get_lr(self.optimizer)]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr, "kind": self.kind, "warmup": self.warmup, "total": self.total}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lr = int(sd["last_epoch"]), float(sd["base_lr"])
        _set_lr(self.optimizer, self.base_lr * self._factor(self.last_epoch))


def _get_lr(optimizer) -> float:
    return optimizer.lr if hasattr(optimizer, "lr") else optimizer.param_groups[0]["lr"]


def _set_lr(optimizer, lr: float) -> None:
    if hasattr(optimizer, "lr"):
        optimizer.lr = lr
    if hasattr(optimizer, "param_groups"):
        for g in optimizer.param_groups:
            if "lr" in g or not hasattr(optimizer, "lr"):
                g["lr"] = lr


def clip_grad_norm_(parameters, max_norm: float, norm_sq_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_`` (what accelerator.clip_grad_norm_ resolves to, trainer.py:497) without leaving the
    device: returns the total L2 norm BEFORE clipping as a 0-dim device tensor and scales the gradients in place by
    ``min(1, max_norm / (norm + 1e-6))``."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    acc = norm_sq_ws if norm_sq_ws is not None else torch.zeros(1, dtype=torch.float64, device=grads[0].device)
    for i, g in enumerate(grads):
        ops.sumsq(g.reshape(-1), out=acc, accumulate=i > 0)
    norm = acc.sqrt()
    coef = torch.clamp(float(max_norm) / (norm + 1e-6), max=1.0).to(torch.float32)
    for g in grads:
        ops.scale_(g.reshape(-1), 1.0, s_dev=coef)
    return norm.to(torch.float32).reshape(())


def get_last_checkpoint(folder: str) -> Optional[str]:
    if not os.path.isdir(folder):
        return None
    found = [(int(m.group(1)), d) for d in os.listdir(folder)
             for m in [re.fullmatch(PREFIX_CHECKPOINT_DIR + r"-(\d+)", d)] if m and os.path.isdir(os.path.join(folder, d))]
    return os.path.join(folder, max(found)[1]) if found else None


class Trainer():
    def __init__(self, model=None, args: TrainingArguments = None, data_collator=None, train_dataset=None, eval_dataset=None,
                 tokenizer=None, optimizers=(None, None), callbacks=None):
        self.args = args if args is not None else TrainingArguments()
        dev = self.args.device or f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        self.device = torch.device(dev)
        self.model = model.to(self.device)
        self.model_wrapped = self.model
        self.tokenizer = tokenizer
        self.data_collator = data_collator
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.optimizer, self.lr_scheduler = optimizers
        self.callbacks = list(callbacks or [])
        self.state = TrainerState(is_local_process_zero=self.is_local_process_zero(), is_world_process_zero=self.is_world_process_zero())
        self.control = TrainerControl()
        self.is_in_train = False
        self._globalstep_last_logged = 0
        self._total_loss_scalar = 0.0

    # ------------------------------------------------------------------------------------------------ small helpers
    def is_local_process_zero(self):
        return int(os.environ.get("LOCAL_RANK", "0")) == 0

    def is_world_process_zero(self):
        return self.args.process_index == 0

    def _fire(self, event: str, **kw):
        for cb in self.callbacks:
            cb = cb() if isinstance(cb, type) else cb
            r = getattr(cb, event)(self.args, self.state, self.control, model=self.model, optimizer=self.optimizer, **kw)
            if r is not None:
                self.control = r

    def get_train_dataloader(self):
        """trainer.py:912-940.  A ready DataLoader (anything with ``__iter__``/``__len__`` that is not a Dataset) is used as is."""
        ds = self.train_dataset
        if isinstance(ds, torch.utils.data.DataLoader) or not hasattr(ds, "__getitem__"):
            return ds
        sampler = None
        if self.args.world_size > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, seed=self.args.seed)
        else:
            sampler = torch.utils.data.RandomSampler(ds, generator=torch.Generator().manual_seed(self.args.seed))
        return torch.utils.data.DataLoader(ds, batch_size=self.args.per_device_train_batch_size, sampler=sampler,
                                           collate_fn=self.data_collator, drop_last=self.args.dataloader_drop_last)

    def create_optimizer_and_scheduler(self, num_training_steps: int):
        if self.optimizer is None:                          # create_optimizer (:816-846): torch-AdamW semantics, fused
            a = self.args
            self.optimizer = AdamW(self.model.parameters(), lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), eps=a.adam_epsilon,
                                   weight_decay=a.weight_decay, decoupled=True)
        if self.lr_scheduler is None:
            self.lr_scheduler = _Schedule(self.optimizer, self.args.lr_scheduler_type, self.args.warmup_steps, num_training_steps)

    def _wrap_model(self, model):
        if self.args.world_size > 1 and not isinstance(model, DistributedDataParallel):
            return DistributedDataParallel(model, device_ids=[self.device.index] if self.device.type == "cuda" else None)
        return model

    def _prepare_inputs(self, inputs):
        return {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in inputs.items()}

    # ------------------------------------------------------------------------------------------------ one micro-batch
    def compute_loss(self, model, inputs, return_outputs=False):
        inputs = {k: v for k, v in inputs.items() if isinstance(v, torch.Tensor)}     # e.g. 'prompts' is a list of str
        outputs = model(**inputs)
        if isinstance(outputs, dict):
            if "loss" not in outputs:
                raise ValueError("the model did not return a loss")
            loss = outputs["loss"]
        else:
            loss = outputs[0]
            if isinstance(loss, (tuple, list)):             # ((loss, logits, hidden), k_v_pasts): the reference's causal LMs
                loss = loss[0]
        return (loss, outputs) if return_outputs else loss

    def training_step(self, model, inputs):
        """trainer.py:543-556; the division by gradient_accumulation_steps is accelerate's ``backward`` (:555)."""
        model.train()
        ga = self.args.gradient_accumulation_steps
        # the fused loss writes dlogits in its forward pass: tell it the upstream gradient this step will send (1/ga), so the backward of
        # `loss / ga` does not pay a second pass over [T,V] (and bf16 dlogits are rounded once, not twice)
        with ops.expected_loss_grad(1.0 / ga):
            loss = self.compute_loss(model, self._prepare_inputs(inputs))
        (loss / ga if ga > 1 else loss).backward()
        return loss.detach() / ga

    # ------------------------------------------------------------------------------------------------ the loop
    def train(self, resume_from_checkpoint=None, **kwargs):
        self.is_in_train = True
        try:
            return self._inner_training_loop(self.args.train_batch_size, self.args, kwargs.pop("model_path", resume_from_checkpoint))
        finally:
            self.is_in_train = False

    def _inner_training_loop(self, batch_size, args, resume_from_checkpoint):
        self._train_batch_size = batch_size
        loader = self.get_train_dataloader()
        has_len = hasattr(loader, "__len__")
        assert has_len or args.max_steps > 0, "a dataloader of unknown length needs max_steps"
        assert args.num_train_epochs > 0 or args.max_steps > 0
        ga = args.gradient_accumulation_steps
        if has_len:                                         # :287-305
            updates_per_epoch = max(len(loader) // ga, 1)
            if args.max_steps > 0:
                max_steps = args.max_steps
                num_train_epochs = max_steps // updates_per_epoch + int(max_steps % updates_per_epoch > 0)
            else:
                max_steps = math.ceil(args.num_train_epochs * updates_per_epoch)
                num_train_epochs = math.ceil(args.num_train_epochs)
        else:
            num_train_epochs, updates_per_epoch, max_steps = 2 ** 62, args.max_steps, args.max_steps

        self.create_optimizer_and_scheduler(max_steps)
        model = self.model_wrapped = self._wrap_model(self.model)

        if isinstance(resume_from_checkpoint, bool) and resume_from_checkpoint:          # :349-362
            resume_from_checkpoint = get_last_checkpoint(args.output_dir)
            if resume_from_checkpoint is None:
                raise ValueError(f"No valid checkpoint found in output directory ({args.output_dir})")
        epochs_trained, skip_in_epoch = 0, 0
        if resume_from_checkpoint is not None:
            self._load_from_checkpoint(resume_from_checkpoint)
            self._load_optimizer_and_scheduler(resume_from_checkpoint)
            st = os.path.join(resume_from_checkpoint, TRAINER_STATE_NAME)
            if os.path.isfile(st):                           # :366-378
                self.state = TrainerState.load_from_json(st)
                assert self.state.train_batch_size == self._train_batch_size, "per-device batch size changed since the checkpoint"
                epochs_trained = self.state.global_step // updates_per_epoch
                if not args.ignore_data_skip:
                    skip_in_epoch = (self.state.global_step % updates_per_epoch) * ga

        def abs_or_ratio(v, default):                       # :401-403
            return default if v is None else (max_steps * v if v < 1 else v)
        self.state.logging_steps = abs_or_ratio(args.logging_steps, self.state.logging_steps)
        self.state.save_steps = abs_or_ratio(args.save_steps, self.state.save_steps)
        self.state.max_steps, self.state.num_train_epochs = max_steps, num_train_epochs
        self.state.train_batch_size = self._train_batch_size
        self.state.epoch = self.state.epoch or 0
        self._fire("on_train_begin")

        tr_loss = torch.zeros((), device=self.device)
        self._globalstep_last_logged = self.state.global_step
        total_batched_samples = 0
        grad_norm = None
        norm_ws = torch.zeros(1, dtype=torch.float64, device=self.device) if self.device.type == "cuda" else None
        params = [p for p in self.model.parameters() if p.requires_grad]
        need_zero = True
        for epoch in range(epochs_trained, num_train_epochs):
            self._fire("on_epoch_begin")
            if hasattr(getattr(loader, "sampler", None), "set_epoch"):
                loader.sampler.set_epoch(epoch)
            steps_in_epoch = len(loader) if has_len else args.max_steps * ga
            it = iter(loader)
            steps_skipped = 0
            if epoch == epochs_trained and resume_from_checkpoint is not None:           # :445-451
                for _ in range(skip_in_epoch):
                    next(it)
                steps_skipped = skip_in_epoch
                self._load_rng_state(resume_from_checkpoint)
            step = -1
            for step, inputs in enumerate(it):
                total_batched_samples += 1
                if args.include_num_input_tokens_seen and "input_ids" in inputs:
                    self.state.num_input_tokens_seen += int(inputs["input_ids"].numel())
                window_start = (total_batched_samples - 1) % ga == 0
                last_short = steps_in_epoch <= ga and (step + 1 + steps_skipped) == steps_in_epoch
                window_end = total_batched_samples % ga == 0 or last_short
                if window_start:
                    self._fire("on_step_begin")
                if need_zero:                               # once per window (see module docstring): set by EVERY window end,
                    self.optimizer.zero_grad()              # including the short last window of an epoch (steps_in_epoch <= ga)
                    need_zero = False
                if not window_end and isinstance(model, DistributedDataParallel):
                    with model.no_sync():
                        tr_loss += self.training_step(model, inputs)
                else:
                    tr_loss += self.training_step(model, inputs)
                if window_end:
                    if args.max_grad_norm is not None and args.max_grad_norm > 0:
                        grad_norm = clip_grad_norm_(params, args.max_grad_norm, norm_ws)
                    self.optimizer.step()
                    self.lr_scheduler.step()
                    need_zero = True                        # applied gradients must never reach the next optimizer step
                    self.state.global_step += 1
                    self.state.epoch = epoch + (step + 1 + steps_skipped) / steps_in_epoch
                    self._default_flow()
                    self._fire("on_step_end")
                    self._maybe_log_save_evaluate(tr_loss, grad_norm, model, epoch)
                else:
                    self._fire("on_substep_end")
                if self.control.should_epoch_stop or self.control.should_training_stop:
                    break
            if step < 0:
                self.control.should_training_stop = True
            self._fire("on_epoch_end")
            self._maybe_log_save_evaluate(tr_loss, grad_norm, model, epoch)
            if self.control.should_training_stop:
                break
        self._fire("on_train_end")
        self._total_loss_scalar += float(tr_loss)
        return TrainOutput(self.state.global_step, self._total_loss_scalar / max(self.state.global_step, 0.001), None)

    def _default_flow(self):
        """transformers' DefaultFlowCallback.on_step_end: raise should_log / should_save / should_training_stop."""
        s, c = self.state, self.control
        c.should_log = s.logging_steps > 0 and s.global_step % max(1, int(s.logging_steps)) == 0
        c.should_save = self.args.save_strategy == "steps" and s.save_steps > 0 and s.global_step % max(1, int(s.save_steps)) == 0
        if s.global_step >= s.max_steps:
            c.should_training_stop = True

    # ------------------------------------------------------------------------------------------------ log / save
    def _maybe_log_save_evaluate(self, tr_loss, grad_norm, model, epoch, ignore_keys_for_eval=None):
        if self.control.should_log and self.state.global_step > self._globalstep_last_logged:       # :1225-1243
            mean = tr_loss.detach().clone()
            if self.args.world_size > 1:
                dist.all_reduce(mean)
                mean /= self.args.world_size
            scalar = mean.item()
            logs = {"loss": round(scalar / (self.state.global_step - self._globalstep_last_logged), 4),
                    "learning_rate": self.lr_scheduler.get_last_lr()[0]}
            if grad_norm is not None:
                logs["grad_norm"] = grad_norm.detach().item() if isinstance(grad_norm, torch.Tensor) else grad_norm
            tr_loss -= tr_loss
            self._total_loss_scalar += scalar
            self._globalstep_last_logged = self.state.global_step
            self.log(logs)
        self.control.should_log = False
        if self.control.should_save:
            self._save_checkpoint(model)
            self.control.should_save = False
            self._fire("on_save")

    def log(self, logs):
        if self.state.epoch is not None:
            logs["epoch"] = self.state.epoch
        if self.args.include_num_input_tokens_seen:
            logs["num_input_tokens_seen"] = self.state.num_input_tokens_seen
        self.state.log_history.append({**logs, "step": self.state.global_step})
        if not self.args.disable_tqdm or os.environ.get("CTMI_TRAINER_PRINT"):
            if self.is_world_process_zero():
                print(logs)
        self._fire("on_log", logs=logs)

    def _get_output_dir(self):
        return self.args.output_dir

    def _save_checkpoint(self, model, metrics=None):
        run_dir = self._get_output_dir()
        output_dir = os.path.join(run_dir, f"{PREFIX_CHECKPOINT_DIR}-{self.state.global_step}")
        self.save_model(output_dir)
        if self.args.should_save:
            if not self.args.save_only_model:
                torch.save(self.optimizer.state_dict(), os.path.join(output_dir, OPTIMIZER_NAME))
                torch.save(self.lr_scheduler.state_dict(), os.path.join(output_dir, SCHEDULER_NAME))
            self.state.save_to_json(os.path.join(output_dir, TRAINER_STATE_NAME))
            if not self.args.save_only_model:
                self._save_rng_state(output_dir)                      # inside the should_save guard, as trainer.py:1318-1324
        if self.args.should_save:
            self._rotate_checkpoints(use_mtime=False, output_dir=run_dir)

    def save_model(self, output_dir=None):
        if self.args.should_save:
            self._save(output_dir or self.args.output_dir)

    def _save(self, output_dir, state_dict=None):
        os.makedirs(output_dir, exist_ok=True)
        state_dict = state_dict or self.model.state_dict()
        if self.args.save_safetensors:
            import safetensors.torch
            flat = {k: v.detach().contiguous().clone() for k, v in state_dict.items()}      # tied tensors may not share storage on disk
            safetensors.torch.save_file(flat, os.path.join(output_dir, SAFE_WEIGHTS_NAME), metadata={"format": "pt"})
        else:
            torch.save(state_dict, os.path.join(output_dir, WEIGHTS_NAME))
        if self.tokenizer is not None and hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(output_dir)
        torch.save(dataclasses.asdict(self.args), os.path.join(output_dir, TRAINING_ARGS_NAME))

    def _save_rng_state(self, output_dir):
        rng = {"python": random.getstate(), "numpy": np.random.get_state(), "cpu": torch.random.get_rng_state()}
        if self.device.type == "cuda":
            rng["cuda"] = torch.cuda.random.get_rng_state(self.device)
        os.makedirs(output_dir, exist_ok=True)
        name = "rng_state.pth" if self.args.world_size <= 1 else f"rng_state_{self.args.process_index}.pth"
        torch.save(rng, os.path.join(output_dir, name))

    def _load_rng_state(self, checkpoint):
        name = "rng_state.pth" if self.args.world_size <= 1 else f"rng_state_{self.args.process_index}.pth"
        path = os.path.join(checkpoint, name)
        if not os.path.isfile(path):
            return
        rng = torch.load(path, weights_only=False)
        random.setstate(rng["python"])
        np.random.set_state(rng["numpy"])
        torch.random.set_rng_state(rng["cpu"])
        if self.device.type == "cuda" and "cuda" in rng:
            torch.cuda.random.set_rng_state(rng["cuda"], self.device)

    def _sorted_checkpoints(self, output_dir, checkpoint_prefix=PREFIX_CHECKPOINT_DIR, use_mtime=False):
        found = []
        for x in Path(output_dir).glob(f"{checkpoint_prefix}-*"):
            if not os.path.isdir(x):
                continue
            if use_mtime:
                found.append((os.path.getmtime(x), str(x)))
            else:
                m = re.match(f".*{checkpoint_prefix}-([0-9]+)", str(x))
                if m is not None:
                    found.append((int(m.groups()[0]), str(x)))
        ordered = [p for _, p in sorted(found)]
        best = self.state.best_model_checkpoint                     # never rotate the best model out (:1504-1510)
        if best is not None and str(Path(best)) in ordered:
            i = ordered.index(str(Path(best)))
            for j in range(i, len(ordered) - 2):
                ordered[j], ordered[j + 1] = ordered[j + 1], ordered[j]
        return ordered

    def _rotate_checkpoints(self, use_mtime, output_dir):
        limit = self.args.save_total_limit
        if limit is None or limit <= 0:
            return
        ordered = self._sorted_checkpoints(use_mtime=use_mtime, output_dir=output_dir)
        if len(ordered) <= limit:
            return
        if self.state.best_model_checkpoint is not None and limit == 1 and ordered[-1] != self.state.best_model_checkpoint:
            limit = 2
        for stale in ordered[:max(0, len(ordered) - limit)]:
            shutil.rmtree(stale, ignore_errors=True)

    # ------------------------------------------------------------------------------------------------ resume
    def _load_from_checkpoint(self, resume_from_checkpoint, model=None):
        model = model or self.model
        weights, safe = os.path.join(resume_from_checkpoint, WEIGHTS_NAME), os.path.join(resume_from_checkpoint, SAFE_WEIGHTS_NAME)
        if not (os.path.isfile(weights) or os.path.isfile(safe)):
            raise ValueError(f"Can't find a valid checkpoint at {resume_from_checkpoint}")
        if self.args.save_safetensors and os.path.isfile(safe) or not os.path.isfile(weights):
            import safetensors.torch
            sd = safetensors.torch.load_file(safe, device="cpu")
        else:
            sd = torch.load(weights, map_location="cpu", weights_only=True)
        result = model.load_state_dict(sd, strict=False)             # :1586
        if hasattr(model, "_tie_weight"):
            model._tie_weight()
        elif hasattr(model, "_tie_weights"):
            model._tie_weights()
        return result

    def _load_optimizer_and_scheduler(self, checkpoint):
        if checkpoint is None:
            return
        opt, sch = os.path.join(checkpoint, OPTIMIZER_NAME), os.path.join(checkpoint, SCHEDULER_NAME)
        if os.path.isfile(opt) and os.path.isfile(sch):
            self.optimizer.load_state_dict(torch.load(opt, map_location=self.device, weights_only=False))
            self.lr_scheduler.load_state_dict(torch.load(sch, weights_only=False))
