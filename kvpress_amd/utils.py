"""Host-side helpers of the hot path: the query projection and cache access.

Mirrors kvpress/utils.py of the reference (get_prerope_query_states :12-53,
extract_keys_and_values :104-114).  These call model-owned modules (``q_proj`` may be
quantised / LoRA-wrapped) and therefore stay PyTorch calls on the model's device.
"""
from __future__ import annotations

import torch
from torch import nn


def _is_instance_named(module: nn.Module, *names: str) -> bool:
    return any(cls.__name__ in names for cls in type(module).__mro__)


def get_prerope_query_states(module: nn.Module, hidden_states: torch.Tensor) -> torch.Tensor:
    """Query states before RoPE, [B, num_heads, q_len, head_dim] (reference utils.py:12-53).

    Phi3 fuses q/k/v in ``qkv_proj`` (queries are the first num_heads*head_dim features);
    Llama-like layers have ``q_proj``; Qwen3 / Gemma3 apply ``q_norm`` per head.
    """
    bsz, q_len, _ = hidden_states.shape
    num_heads = module.config.num_attention_heads
    head_dim = module.head_dim
    if _is_instance_named(module, "Phi3Attention"):
        query_states = module.qkv_proj(hidden_states)[..., : num_heads * head_dim]
    elif hasattr(module, "q_proj"):
        query_states = module.q_proj(hidden_states)
    else:
        raise NotImplementedError(f"Press not yet implemented for {module.__class__}.")
    query_states = query_states.view(bsz, q_len, num_heads, head_dim).transpose(1, 2)
    if _is_instance_named(module, "Qwen3Attention", "Gemma3Attention"):
        query_states = module.q_norm(query_states)
    return query_states


def _is_quantized(cache) -> bool:
    try:
        from transformers import QuantizedCache
    except Exception:  # pragma: no cover
        return False
    return isinstance(cache, QuantizedCache)


def extract_keys_and_values(cache, layer_idx: int) -> tuple[torch.Tensor, torch.Tensor]:
    """K and V of a cache layer; dequantised for a QuantizedCache (reference utils.py:98-114)."""
    layer = cache.layers[layer_idx]
    if _is_quantized(cache):
        return layer._dequantize(layer._quantized_keys), layer._dequantize(layer._quantized_values)
    return layer.keys, layer.values
