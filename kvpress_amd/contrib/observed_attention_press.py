"""ObservedAttentionPress (kvpress/presses/observed_attention_press.py:13-48) on kvp_observed_attention_score."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class ObservedAttentionPress(ScorerPress):
    """Scores from the attention weights observed during the forward pass: the average weight a key receives from the
    queries that can see it (related to H2O, https://arxiv.org/abs/2306.14048).

    Requires ``attn_implementation="eager"`` (the layer must return its attention weights).

    Parameters
    ----------
    compression_ratio : float, default=0.0
    """

    compression_ratio: float = 0.0

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        assert attentions is not None, 'Set attn_implementation="eager" to use this hook'
        return _native.observed_attention_score(attentions, keys.shape[1])
