"""Checkpoint / config loading with the reference's key contract (examples/inference_bloom.py:16-63).

``load_model`` accepts a state dict in this repo's own layout, in HuggingFace layout (with or without the
``transformer.`` prefix) and — unlike the reference — also one saved from the DDP wrapper (``module.``)."""
from __future__ import annotations

import json
from collections import OrderedDict

import torch

from ..models.modeling_bloom import BloomConfig, BloomForCausalLM

_PER_BLOCK = ("input_layernorm", "self_attention.query_key_value", "self_attention.dense",
              "post_attention_layernorm", "mlp.dense_h_to_4h", "mlp.dense_4h_to_h")


def map_state_dict(state_dict, n_layer: int):
    if any(k.startswith("module.") for k in state_dict):
        state_dict = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in state_dict.items())
    if "bloom.word_embeddings.weight" in state_dict:
        return state_dict
    pre = "transformer." if "transformer.word_embeddings.weight" in state_dict else ""
    out = OrderedDict()
    out["bloom.word_embeddings.weight"] = state_dict[pre + "word_embeddings.weight"]
    for t in ("weight", "bias"):
        out[f"bloom.word_embeddings_layernorm.{t}"] = state_dict[f"{pre}word_embeddings_layernorm.{t}"]
    for i in range(n_layer):
        for t in ("weight", "bias"):
            for name in _PER_BLOCK:
                out[f"bloom.blocks.{i}.{name}.{t}"] = state_dict[f"{pre}h.{i}.{name}.{t}"]
    for t in ("weight", "bias"):
        out[f"bloom.ln_f.{t}"] = state_dict[f"{pre}ln_f.{t}"]
    out["lm_head.weight"] = state_dict.get("lm_head.weight", state_dict[pre + "word_embeddings.weight"])
    return out


def load_state(model: BloomForCausalLM, state_dict) -> BloomForCausalLM:
    model.load_state_dict(map_state_dict(state_dict, model.config.n_layer), strict=True)
    model.eval()
    model._tie_weight()
    return model


def load_model(config, ckpt_path):
    state_dict = torch.load(ckpt_path, map_location="cpu")
    if not isinstance(state_dict, dict):
        state_dict = state_dict.state_dict()
    return load_state(BloomForCausalLM(config), state_dict)


def config_from_dict(d: dict) -> BloomConfig:
    d = dict(d)
    for syn in (("n_embed", "hidden_size"), ("n_head", "num_attention_heads")):
        src = next((k for k in syn if k in d), None)
        if src is not None:
            for k in syn:
                d[k] = d[src]
    return BloomConfig(**d)


def load_config(config_fn):
    return config_from_dict(json.load(open(config_fn, "r")))
