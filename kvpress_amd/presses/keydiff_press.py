"""KeyDiffPress (kvpress/presses/keydiff_press.py:15-46) on kvp_keydiff_score."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class KeyDiffPress(ScorerPress):
    """KeyDiff (https://arxiv.org/abs/2504.15364): keys most similar (cosine) to the head's average normalised key are
    evicted first.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    """

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        return _native.keydiff_score(keys)
