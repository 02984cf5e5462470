from cleantransformer_amd.models.modeling_bloom import *  # noqa: F401,F403
from cleantransformer_amd.models.modeling_bloom import (BloomConfig, BloomForCausalLM, BloomModel, BloomBlock, BloomAttentionLayer,  # noqa: F401
                                                        BloomMLP, BloomGelu)
