from cleantransformer_amd.trainer.trainer import (Trainer, TrainerCallback, TrainerControl, TrainerState,  # noqa: F401
                                                  TrainingArguments, TrainOutput, clip_grad_norm_, get_last_checkpoint)
