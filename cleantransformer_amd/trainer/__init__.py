from .ddp import DistributedDataParallel  # noqa: F401
from .trainer import (Trainer, TrainerCallback, TrainerControl, TrainerState, TrainingArguments, TrainOutput,  # noqa: F401
                      clip_grad_norm_, get_last_checkpoint)
