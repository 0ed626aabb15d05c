"""CURPress (kvpress/presses/cur_press.py:15-67) on kvp_cur_score."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Literal

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class CURPress(ScorerPress):
    """CUR-style pruning (https://arxiv.org/abs/2509.15038): approximate leverage scores -- squared l2 norms of keys and
    values, normalised inside local windows -- decide which tokens stay.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    num_sinks : int, default=4
        The first tokens always get score 1.
    leverage_type : "key" | "value" | "kv_avg" | "kv_product", default "kv_product"
    use_random_leverage : bool, default=False
        Project keys and values onto 20 random directions first (a torch GEMM with torch's generator, as in the reference).
    use_local_approximation : bool, default=True
    local_window_size : int, default=16
    """

    num_sinks: int = 4
    leverage_type: Literal["key", "value", "kv_avg", "kv_product"] = "kv_product"
    use_random_leverage: bool = False
    use_local_approximation: bool = True
    local_window_size: int = 16

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        if self.use_random_leverage:  # cur_press.py:34-38
            r = 20
            G = torch.randn(keys.shape[-1], r, device=keys.device) / math.sqrt(r)
            keys = keys @ G.to(keys.dtype)
            values = values @ G.to(values.dtype)
        return _native.cur_score(keys, values, self.leverage_type, self.local_window_size if self.use_local_approximation else 0,
                                 self.num_sinks)
