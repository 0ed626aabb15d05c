"""DMSPress (kvpress/presses/dms_press.py:16-130): threshold-based eviction with a sliding window of protected tokens.

Every token gets a score from the wrapped ScorerPress (library) when it enters the cache; once it leaves the sliding
window it is evicted if its score is below ``threshold``.  Evicted tokens stay in the cache and are masked during
attention (``module.masked_key_indices`` + kvpress_amd.attention_patch, as AdaKVPress).  Host logic around the score:
the comparison against the threshold is one elementwise op on the [B, H, n] scores that leave the window."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
from torch import nn

from kvpress_amd.attention_patch import patch_attention_functions
from kvpress_amd.presses.base_press import BasePress, is_prefilling
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.utils import _is_quantized, extract_keys_and_values


@dataclass
class DMSPress(BasePress):
    """Dynamic Memory Sparsification inspired eviction (https://arxiv.org/abs/2506.05345) on any ScorerPress.

    Parameters
    ----------
    press : ScorerPress
        Scores every token when it enters the cache.
    threshold : float, optional
        Tokens whose score is below it are evicted once they leave the sliding window.
    sliding_window_size : int, default=128
        Most recent tokens that are never evicted.
    decoding : bool, default=False
        Also evict during decoding (the pipeline then keeps the hook for the answers).
    """

    press: ScorerPress
    threshold: Optional[float] = None
    sliding_window_size: int = 128
    decoding: bool = False
    scores_buffer: dict = field(default_factory=dict, init=False, repr=False)
    compression_ratios: dict = field(default_factory=dict, init=False, repr=False)

    def __post_init__(self):
        patch_attention_functions()

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    @property
    def compression_ratio(self):
        assert len(self.compression_ratios) > 0, "Forward pass must be run to compute the compression ratio"
        return sum(self.compression_ratios.values()) / len(self.compression_ratios)

    @compression_ratio.setter
    def compression_ratio(self, value):
        raise AttributeError(f"compression ratio cannot be set for {type(self).__name__}")

    def forward_hook(self, module: nn.Module, input: list[torch.Tensor], kwargs: dict, output: list):
        hidden_states = kwargs["hidden_states"]
        cache = kwargs["past_key_values"]
        q_len = hidden_states.shape[1]
        layer_idx = module.layer_idx
        # tokens in the cache after this forward (the reference reads cache_position[-1] + 1, :85)
        cache_len = cache.get_seq_length(layer_idx) if _is_quantized(cache) else cache.layers[layer_idx].keys.shape[2]
        prefilling = is_prefilling(cache_len, q_len, kwargs, cache.layers[layer_idx])
        if prefilling and layer_idx == 0:
            self.scores_buffer.clear()
            self.compression_ratios.clear()
        if not prefilling and not self.decoding:
            return output

        keys, values = extract_keys_and_values(cache, layer_idx)
        scores = self.press.score(module, hidden_states, keys[:, :, -q_len:], values[:, :, -q_len:], None, kwargs)
        self.scores_buffer[layer_idx] = scores if prefilling else torch.cat([self.scores_buffer[layer_idx], scores], dim=-1)

        if self.scores_buffer[layer_idx].shape[-1] > self.sliding_window_size:
            n_to_evict = self.scores_buffer[layer_idx].shape[-1] - self.sliding_window_size
            scores_to_evict = self.scores_buffer[layer_idx][..., :n_to_evict]
            self.scores_buffer[layer_idx] = self.scores_buffer[layer_idx][..., n_to_evict:]
            new = list(torch.where(scores_to_evict < self.threshold))
            if len(new[0]) > 0:
                new[-1] = new[-1] + (cache_len - scores_to_evict.shape[2] - self.sliding_window_size)   # positions in the cache (:105-106)
                if getattr(module, "masked_key_indices", None) is None:
                    module.masked_key_indices = new
                else:
                    module.masked_key_indices = [torch.cat([i, j]) for i, j in zip(module.masked_key_indices, new)]

        if getattr(module, "masked_key_indices", None) is not None:
            bsz, num_key_value_heads, n, _ = keys.shape
            self.compression_ratios[layer_idx] = len(module.masked_key_indices[0]) / (bsz * num_key_value_heads * n)
        else:
            self.compression_ratios[layer_idx] = 0
        return output
