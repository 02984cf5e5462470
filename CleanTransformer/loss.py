from cleantransformer_amd.loss import CrossEntropyLoss  # noqa: F401
