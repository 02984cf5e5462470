from cleantransformer_amd.optimizer import AdamW, SGD  # noqa: F401
