from .ddp import DistributedDataParallel  # noqa: F401
