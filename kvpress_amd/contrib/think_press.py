"""ThinKPress (kvpress/presses/think_press.py:16-98): prunes key CHANNELS instead of tokens.

kvp_think_channel_scores (window-query energy x key energy per channel) -> kvp_topk_select | KVP_TOPK_SMALLEST over the
[B*H, D] rows -> kvp_zero_channels, in place on the cache's key tensor, as the reference's ``keys.scatter_`` (:82)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.snapkv_press import SnapKVPress


@dataclass
class ThinKPress(BasePress):
    """ThinK (https://arxiv.org/abs/2407.21018): the key channels with the lowest query-key interaction are zeroed.
    The cache keeps its shape; the memory saving needs a kernel that skips the zeroed channels (not part of this press,
    as in the reference).

    Parameters
    ----------
    key_channel_compression_ratio : float, default=0.0
        Fraction of key channels that is pruned.
    window_size : int, default=32
        Number of recent tokens whose queries weigh the channels.
    """

    key_channel_compression_ratio: float = 0.0
    window_size: int = 32

    def compute_window_queries(self, module, hidden_states, position_embeddings):
        """RoPE'd queries of the last ``window_size`` tokens (think_press.py:40-54)."""
        return SnapKVPress.compute_window_queries(module, hidden_states, self.window_size, position_embeddings)

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if self.key_channel_compression_ratio == 0:
            return keys, values
        head_dim = keys.shape[-1]
        queries = self.compute_window_queries(module, kwargs["hidden_states"], kwargs["position_embeddings"])
        key_scores = _native.think_channel_scores(queries, keys)                                       # [B, H, D]  (:72-76)
        n_pruned = int(head_dim * self.key_channel_compression_ratio)                                  # (:79)
        indices = _native.topk_select(key_scores, n_pruned, _native.ORDER_POSITION | _native.TOPK_SMALLEST)
        _native.zero_channels_(keys, indices)                                                          # in place (:82)
        return keys, values

    @property
    def compression_ratio(self):
        return self.key_channel_compression_ratio / 2

    @compression_ratio.setter
    def compression_ratio(self, value):
        raise AttributeError(f"compression ratio cannot be set for {type(self).__name__}")
