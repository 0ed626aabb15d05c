"""TOVAPress (kvpress/presses/tova_press.py:16-61) on the SnapKV kernels with a one-token window."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.utils import get_prerope_query_states


@dataclass
class TOVAPress(ScorerPress):
    """TOVA (https://arxiv.org/abs/2401.06104): the attention the LAST token pays to the earlier keys, averaged over
    all heads, is every kv-head's score; the last token itself is never pruned.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    """

    compression_ratio: float = 0.0

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        k_len = keys.shape[2]
        if attentions is not None:
            attn = attentions[..., -1:, :-1]                         # tova_press.py:45-46
            scores = _native.snapkv_score_from_attn(attn, keys.shape[1], k_len, 1)
        else:
            # window attention with window 1, no pooling (tova_press.py:48-50): per-kv-group means, last column padded
            q_pre = get_prerope_query_states(module, hidden_states[:, -1:])
            cos, sin = kwargs["position_embeddings"]
            scores = _native.snapkv_score_rope(q_pre, cos[:, -1:], sin[:, -1:], keys, 1)
        # mean over all heads, repeated for every kv-head (:52-53); the pad column (max + 1, :58) is the same in
        # every group, so it stays the row maximum
        return _native.scores_head_mean_(scores)
