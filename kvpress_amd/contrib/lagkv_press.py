"""LagKVPress (kvpress/presses/lagkv_press.py:13-97) on kvp_lagkv_score."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class LagKVPress(ScorerPress):
    """LagKV (https://arxiv.org/abs/2504.04704): attention-free scores from the next partition's statistics -- tokens of a
    partition are normalised with the channel-wise min / max of the FOLLOWING partition; the spread of the normalised
    vector is the token's importance.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    n_sink : int, default=4
        Leading tokens that are always kept (score 1).
    lag_size : int, default=128
        Partition length.
    cross_scoring : bool, default=False
        Keep the raw (softmax) scores, comparable across partitions, instead of the rank inside the partition.
    """

    compression_ratio: float = 0.0
    n_sink: int = 4
    lag_size: int = 128
    cross_scoring: bool = False

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        bsz, num_key_value_heads, q_len, _ = keys.shape
        if q_len < self.n_sink + 2 * self.lag_size:   # too short to compare partitions: keep the sinks and the most recent (:57-63)
            score = torch.ones((bsz, num_key_value_heads, q_len), dtype=torch.float32, device=keys.device)
            if q_len > self.n_sink:
                score[:, :, self.n_sink:] = torch.arange(q_len - self.n_sink, device=keys.device) / (q_len - self.n_sink)
            return score
        return _native.lagkv_score(keys, values, self.n_sink, self.lag_size, self.cross_scoring)
