from cleantransformer_amd.models.modeling_gpt import *  # noqa: F401,F403
from cleantransformer_amd.models.modeling_gpt import (GPTConfig, GPTLMHeadModel, GPTModel, TransformerBlock, AttentionLayer,  # noqa: F401
                                                      Conv1D, NewGELUActivation)
