"""DecodingPress / PrefillDecodingPress (kvpress/presses/decoding_press.py:21-239, prefill_decoding_press.py:19-91):
compression DURING decoding.

The hook does nothing while the context is pre-filled; afterwards it buffers each layer's hidden states and, every
``compression_interval`` forward calls (or at once when a call brings ``target_size`` tokens or more), prunes that layer's
cache to ``target_size`` tokens with the base press -- whose ``compress`` is the same HIP path as in prefill, now in its
small-S, latency-bound regime (one fused library call for Knorm)."""
from __future__ import annotations

import logging
from collections import defaultdict
from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch import nn

from kvpress_amd.presses.base_press import BasePress, is_prefilling
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.utils import _is_quantized, extract_keys_and_values

logger = logging.getLogger(__name__)


def _kv_len(cache, layer_idx: int) -> int:
    return cache.get_seq_length(layer_idx) if _is_quantized(cache) else cache.layers[layer_idx].keys.shape[2]


@dataclass
class DecodingPress(BasePress):
    """Parameters
    ----------
    base_press : ScorerPress
        Scores the tokens when a compression is due (its own ``compression_ratio`` is overridden).
    compression_interval : int, default=512
        Forward calls (decoding steps) between two compressions of a layer.
    target_size : int, default=2048
        Tokens a layer keeps after a compression.
    hidden_states_buffer_size : int, default=256
        Hidden states kept per layer between compressions (0: only the current step's).
    """

    base_press: ScorerPress
    compression_interval: int = 512
    target_size: int = 2048
    hidden_states_buffer_size: int = 256

    def __post_init__(self):
        assert isinstance(self.base_press, ScorerPress), "DecodingPress requires a ScorerPress as input"
        assert self.compression_interval > 0, "compression_interval must be greater than 0"
        assert self.target_size > 0, "target_size must be greater than 0"
        self.reset()
        if self.base_press.compression_ratio:
            logger.warning(f"compression_ratio is set for base press ({self.base_press.compression_ratio}). "
                           f"This will be overridden by the decoding press.")

    def post_init_from_model(self, model):
        self.base_press.post_init_from_model(model)

    def reset(self):
        """Forget the buffered hidden states and step counters (also done when the context manager exits)."""
        self.hidden_states_buffer = defaultdict(list)
        self.layer_step_counts = defaultdict(int)

    def _resolve_target_size(self, kwargs: dict) -> int:
        return self.target_size

    @staticmethod
    def _find_target_compression_ratio(q_len: int, target_tokens: int) -> float:
        """A ratio r with ``int(q_len * (1 - r)) == target_tokens`` (decoding_press.py:196-236): the exact quotient, then a
        bisection (at most 20 halvings) against the truncation of ScorerPress's n_kept formula."""
        if q_len <= target_tokens:
            return 0.0
        ratio, low, high = 1.0 - target_tokens / q_len, 0.0, 1.0
        for _ in range(20):
            n_kept = int(q_len * (1 - ratio))
            if n_kept == target_tokens:
                break
            if n_kept > target_tokens:   # compress more
                low, ratio = ratio, (ratio + high) / 2
            else:                        # compress less
                high, ratio = ratio, (low + ratio) / 2
        if int(q_len * (1 - ratio)) != target_tokens:
            logger.warning(f"Binary search failed: q_len={q_len}, target={target_tokens}, got={int(q_len * (1 - ratio))}, ratio={ratio}")
        return ratio

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        """``hidden_states`` are the buffered ones (buffer_len tokens), keys / values the whole history
        (decoding_press.py:68-112)."""
        ratio = self._find_target_compression_ratio(keys.shape[2], self._resolve_target_size(kwargs))
        saved = self.base_press.compression_ratio
        self.base_press.compression_ratio = ratio
        try:
            return self.base_press.compress(module, hidden_states, keys, values, attentions, kwargs)
        finally:
            self.base_press.compression_ratio = saved

    def forward_hook(self, module: nn.Module, input: list[torch.Tensor], kwargs: dict, output: list):
        hidden_states = kwargs["hidden_states"]
        cache = kwargs["past_key_values"]
        q_len = hidden_states.shape[1]
        layer_idx = module.layer_idx
        if is_prefilling(_kv_len(cache, layer_idx), q_len, kwargs, cache.layers[layer_idx]):
            return output                                   # still pre-filling: nothing to do

        self.hidden_states_buffer[layer_idx].append(hidden_states.detach().clone())
        self.layer_step_counts[layer_idx] += 1
        if self.layer_step_counts[layer_idx] >= self.compression_interval or q_len >= self._resolve_target_size(kwargs):
            cache_layer = cache.layers[layer_idx]
            keys, values = extract_keys_and_values(cache, layer_idx)
            attentions = output[1] if isinstance(output, (tuple, list)) and len(output) > 1 and output[1] is not None else None
            buffered = torch.cat(self.hidden_states_buffer[layer_idx], dim=1)
            keys, values = self.compress(module, buffered, keys, values, attentions, kwargs)
            if _is_quantized(cache):
                cache_layer._quantized_keys = cache_layer._quantize(keys, axis=cache_layer.axis_key)
                cache_layer._quantized_values = cache_layer._quantize(values, axis=cache_layer.axis_value)
                cache_layer.keys = torch.zeros(0, dtype=keys.dtype, device=keys.device)
                cache_layer.values = torch.zeros(0, dtype=keys.dtype, device=keys.device)
                cache_layer.cumulative_length = keys.shape[2]
            else:
                cache_layer.keys = keys
                cache_layer.values = values
            self.layer_step_counts[layer_idx] = 0
            self.hidden_states_buffer[layer_idx] = []      # buffer and cache must describe the same tokens

        n = self.hidden_states_buffer_size
        self.hidden_states_buffer[layer_idx] = self.hidden_states_buffer[layer_idx][-n:] if n > 0 else []
        return output

    @contextmanager
    def __call__(self, model):
        try:
            with super().__call__(model):
                yield
        finally:
            self.reset()


@dataclass
class CompressionRatioDecodingPress(DecodingPress):
    """A decoding press that keeps a fixed FRACTION of all tokens seen so far instead of an absolute ``target_size``
    (kvpress/presses/compression_ratio_decoding_press.py:9-50).  Needs the logical ``position_ids`` among the attention
    layer's kwargs (the pipeline and ``generate`` pass them).

    Parameters
    ----------
    base_press : ScorerPress
    target_compression_ratio : float, default=0.5
        Fraction of all tokens seen so far that is removed at a compression.
    compression_interval, hidden_states_buffer_size : as DecodingPress
    """

    target_compression_ratio: float = 0.5
    target_size: int = field(default=1, init=False)

    def __post_init__(self):
        super().__post_init__()
        assert 0 <= self.target_compression_ratio < 1, "target_compression_ratio must be between 0 and 1"

    def _resolve_target_size(self, kwargs: dict) -> int:
        return max(1, int(self._resolve_total_tokens_seen(kwargs) * (1 - self.target_compression_ratio)))

    def _resolve_total_tokens_seen(self, kwargs: dict) -> int:
        if kwargs.get("position_ids") is not None:
            return int(kwargs["position_ids"].max().item()) + 1
        raise NotImplementedError("CompressionRatioDecodingPress requires logical position_ids in kwargs")


@dataclass
class PrefillDecodingPress(BasePress):
    """One press object, two phases: ``prefilling_press`` while the context is pre-filled, ``decoding_press`` afterwards.

    Parameters
    ----------
    prefilling_press : BasePress, optional
    decoding_press : DecodingPress, optional
    """

    prefilling_press: Optional[BasePress] = None
    decoding_press: Optional[DecodingPress] = None

    def post_init_from_model(self, model):
        for p in (self.prefilling_press, self.decoding_press):
            if p is not None:
                p.post_init_from_model(model)

    def _phase_press(self, kv_len: int, q_len: int, kwargs: dict | None = None):
        if is_prefilling(kv_len, q_len, kwargs) and self.prefilling_press is not None:
            return self.prefilling_press
        return self.decoding_press

    def compress(self, module, hidden_states, keys, values, attentions, kwargs):
        press = self._phase_press(keys.shape[2], hidden_states.shape[1], kwargs)
        if press is None:
            logger.warning("No compression applied during prefill or decoding phase")
            return keys, values
        return press.compress(module, hidden_states, keys, values, attentions, kwargs)

    def forward_hook(self, module: nn.Module, input: list[torch.Tensor], kwargs: dict, output: list):
        if not getattr(self, "_order_checked", False) and self.prefilling_press is not None and self.decoding_press is not None:
            # the decoding press works on what the prefill press left: an order-dependent one must see the reference's score order
            from kvpress_amd.presses.scorer_press import resolve_chain_kept_order

            self._order_checked = True
            resolve_chain_kept_order(self.prefilling_press, [self.decoding_press], "PrefillDecodingPress")
        press = self._phase_press(_kv_len(kwargs["past_key_values"], module.layer_idx), kwargs["hidden_states"].shape[1])
        return output if press is None else press.forward_hook(module, input, kwargs, output)

    @contextmanager
    def __call__(self, model):
        try:
            with super().__call__(model):
                yield
        finally:
            if self.decoding_press is not None:
                self.decoding_press.reset()
