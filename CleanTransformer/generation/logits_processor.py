from cleantransformer_amd.generation.logits_processor import (NoRepeatNGramLogitsProcessor, TemperatureLogitsWrapper,  # noqa: F401
                                                              TopKLogitsWrapper, TopPLogitsWrapper)
