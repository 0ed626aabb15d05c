"""KnormPress (kvpress/presses/knorm_press.py:13-38): score = -||k||_2, on kvp_rownorm_score / kvp_knorm_compress."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class KnormPress(ScorerPress):
    """Key-norm based KV cache compression (https://arxiv.org/pdf/2406.11430): keys with a
    small L2 norm are kept.

    Parameters
    ----------
    compression_ratio : float, default=0.0
        Fraction of key-value pairs to remove during compression.
    """

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        return _native.rownorm_score(keys, -1.0)  # -keys.norm(dim=-1)  (knorm_press.py:38), float32

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        """ScorerPress.compress (scorer_press.py:76-102) as ONE library call: norm + first select pass, select, gather.
        A subclass that overrides ``score`` gets the generic three-call sequence."""
        if self.compression_ratio == 0:
            return keys, values
        if type(self).score is not KnormPress.score:
            return super().compress(module, hidden_states, keys, values, attentions, kwargs)
        order = _native.ORDER_SCORE if self.kept_order == "score" else _native.ORDER_POSITION   # "score": the reference's row order
        return _native.knorm_compress(keys, values, self.n_kept(module, keys.shape[2]), order)
