"""BasePress: the forward-hook boundary of the hot path.

Same protocol as the reference's kvpress/presses/base_press.py (BasePress :43-207): a context
manager registers ``forward_hook`` (with_kwargs=True) on every ``self_attn``; after each
attention forward during prefill the hook pulls K/V out of the HF cache, calls ``compress`` and
writes the compressed K/V back.  Differences, all host-side:
  * prefill detection never synchronises: ``cache_position[-1] + 1 == q_len`` (:37-40, a
    device->host ``.item()``) is equivalent to "the cache held nothing before this forward",
    i.e. ``kv_len <= q_len`` after the layer's update, which is read from tensor shapes;  transformers >= 5.x no longer
    passes ``cache_position`` to the attention layer at all;
  * layers are also located for decoder-only models that keep them under ``model.decoder``
    (OPT: BASELINE config 1), and ``rotary_emb`` is attached only when the model has one.
"""
from __future__ import annotations

import logging
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Generator

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.utils import _is_quantized, extract_keys_and_values

logger = logging.getLogger(__name__)

SUPPORTED_MODEL_NAMES = (
    "LlamaForCausalLM",
    "MistralForCausalLM",
    "Phi3ForCausalLM",
    "Qwen2ForCausalLM",
    "Qwen3ForCausalLM",
    "Gemma3ForConditionalGeneration",
)


# cache layers whose stored length IS their logical length: the tensor shapes alone tell a prefill from a later step
_SHAPE_DECIDES = frozenset({"DynamicLayer", "QuantizedLayer", "QuantoQuantizedLayer", "HQQQuantizedLayer"})


def is_prefilling(kv_len: int, q_len: int, kwargs: dict | None = None, cache_layer=None) -> bool:
    """True for the initial prefill.
    * ``cache_layer`` is a DynamicCache / QuantizedCache layer: the shapes decide, without a device sync -- the cache held
      nothing before this forward, i.e. after the layer's update it holds the q_len tokens of this forward, or fewer when an
      earlier press of a ComposedPress has already pruned them (a continuation or decoding step leaves kv_len = past + q_len >
      q_len).  This is the path of every decoded token under a DecodingPress: no ``.item()`` per layer and token.
    * any other layer (pre-allocated static cache, sliding window, unknown) with ``cache_position`` passed by the model
      (transformers < 5.3): the reference's own rule, ``cache_position[-1] + 1 == q_len`` (base_press.py:37-40; one sync).
    * otherwise (transformers >= 5.3 passes no cache_position): the shape rule."""
    if cache_layer is not None and type(cache_layer).__name__ in _SHAPE_DECIDES:
        return int(kv_len) <= int(q_len)
    cache_position = None if kwargs is None else kwargs.get("cache_position")
    if cache_position is not None:
        return int(cache_position[-1]) + 1 == int(q_len)
    return int(kv_len) <= int(q_len)


def _language_model(model):
    inner = model.model if hasattr(model, "model") else model
    if hasattr(inner, "language_model"):
        inner = inner.language_model
    if not hasattr(inner, "layers") and hasattr(inner, "decoder"):
        inner = inner.decoder  # OPT-style
    return inner


@dataclass
class BasePress:
    """Base class of all KV-cache compression methods (reference base_press.py:43-53).
    Compression is applied during pre-filling only."""

    def post_init_from_model(self, model):
        """Optional hook to initialise press parameters from the model (base_press.py:55-59)."""
        pass

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        """Core logic: return the compressed (keys, values) (base_press.py:61-99)."""
        raise NotImplementedError("compress method must be implemented in subclass")

    def forward_hook(self, module: nn.Module, input: list[torch.Tensor], kwargs: dict, output: list):
        """Forward hook of an attention layer (base_press.py:101-162)."""
        hidden_states = kwargs["hidden_states"]
        cache = kwargs["past_key_values"]
        cache_layer = cache.layers[module.layer_idx]
        q_len = hidden_states.shape[1]

        # Don't compress after pre-filling
        kv_len = cache.get_seq_length(module.layer_idx) if _is_quantized(cache) else cache_layer.keys.shape[2]
        if not is_prefilling(kv_len, q_len, kwargs, cache_layer):
            return output

        keys, values = extract_keys_and_values(cache, module.layer_idx)
        attentions = output[1] if isinstance(output, (tuple, list)) and len(output) > 1 else None
        keys, values = self.compress(module, hidden_states, keys, values, attentions, kwargs)

        if _is_quantized(cache):
            cache_layer._quantized_keys = cache_layer._quantize(keys, axis=cache_layer.axis_key)
            cache_layer._quantized_values = cache_layer._quantize(values, axis=cache_layer.axis_value)
            cache_layer.keys = torch.zeros(0, dtype=keys.dtype, device=keys.device)
            cache_layer.values = torch.zeros(0, dtype=keys.dtype, device=keys.device)
            cache_layer.cumulative_length = keys.shape[2]
        else:
            cache_layer.keys = keys
            cache_layer.values = values
        return output

    @contextmanager
    def __call__(self, model) -> Generator:
        """Context manager applying the press to ``model`` (base_press.py:164-207):

        >>> with press(model):
        ...     model(input_ids, past_key_values=cache)
        """
        if type(model).__name__ not in SUPPORTED_MODEL_NAMES:
            logger.warning(f"Model {type(model)} not tested, supported models: {SUPPORTED_MODEL_NAMES}")
        is_gemma3 = type(model).__name__ == "Gemma3ForConditionalGeneration"
        if is_gemma3:
            logger.warning("Compression in Gemma3 is only applied to layer without sliding window attention")

        self.post_init_from_model(model)
        hooks = []
        try:
            language_model = _language_model(model)
            for layer in language_model.layers:
                if is_gemma3 and getattr(layer.self_attn, "is_sliding", False):
                    continue
                if hasattr(language_model, "rotary_emb"):
                    layer.self_attn.rotary_emb = language_model.rotary_emb
                hooks.append(layer.self_attn.register_forward_hook(self.forward_hook, with_kwargs=True))
            yield
        finally:
            for hook in hooks:
                hook.remove()
        # A select kernel that found at RUN time that it could not produce valid indices (include/kvpress_hip.h, KVP_EASYNC) must not
        # pass silently: poll the library's status word on the way out (reached only when the body did not raise).  The poll never
        # waits for the GPU -- a prefill's kernels may still be running here -- so it is BEST EFFORT for the last layers: a failure
        # they report later surfaces at the next library call (every select / compress / gather entry point checks first), or at
        # ``kvpress_amd._native.async_error_check()`` after the caller's own synchronisation.  What it always catches is everything
        # that has already run -- in particular all but the last layers of a long prefill -- and a poisoned cache cannot be USED
        # silently either way: its rows are NaN.
        _native.async_error_check()
