from cleantransformer_amd.transformer import *  # noqa: F401,F403
from cleantransformer_amd.transformer import LayerNorm, AttentionLayer, TransformerBlock, ExampleConfig  # noqa: F401
