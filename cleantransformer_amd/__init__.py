"""cleantransformer_amd — MI355X (gfx950) native implementation of the CleanTransformer data-parallel SFT hot path.

Public names mirror the reference package layout:
    cleantransformer_amd.transformer.{LayerNorm, AttentionLayer, TransformerBlock}
    cleantransformer_amd.loss.CrossEntropyLoss
    cleantransformer_amd.optimizer.{AdamW, SGD}
    cleantransformer_amd.models.modeling_bloom.{BloomConfig, BloomForCausalLM, ...}
    cleantransformer_amd.trainer.DistributedDataParallel
The top-level ``CleanTransformer`` package in this repository re-exports them under the reference's import paths.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def build(force: bool = False) -> str:
    """Compile libctmi355.so in-tree with hipcc for gfx950."""
    from ._build import build as _b
    return _b(force=force)
