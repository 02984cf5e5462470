from cleantransformer_amd.trainer.ddp import DistributedDataParallel  # noqa: F401
