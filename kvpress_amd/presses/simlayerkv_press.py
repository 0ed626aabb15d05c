"""SimLayerKVPress (kvpress/presses/simlayerkv_press.py:18-116): "lazy" layers keep only the first and the most recent tokens.

A layer is lazy when its last ``n_last`` queries put more than ``lazy_threshold`` of their attention on the first
``n_initial`` and the last ``n_recent`` keys.  The window attention is the SnapKV kernels' (``kvp_snapkv_score_rope`` with
W = n_last and no pooling: the mean over window rows and GQA group per kv-head); its mean over batch and heads and the two
partial sums are a reduction over one [B, H, S] float tensor (torch glue, one host read for the decision, as in the
reference's ``score.item()``); a lazy layer's cache is cut with ``kvp_gather_kv``."""
from __future__ import annotations

import logging
from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.utils import get_prerope_query_states

logger = logging.getLogger(__name__)


@dataclass
class SimLayerKVPress(BasePress):
    """SimLayerKV (https://arxiv.org/abs/2410.13846).

    Parameters
    ----------
    lazy_threshold : float, default=1.0
        Attention mass on initial + recent tokens above which a layer is lazy (1.0: never).
    n_last : int, default=1
        Queries (last tokens) that are inspected.
    n_recent : int, default=1024
        Recent tokens a lazy layer keeps.
    n_initial : int, default=4
        Initial (sink) tokens a lazy layer keeps.
    """

    lazy_threshold: float = 1.0
    n_last: int = 1
    n_recent: int = 1024
    n_initial: int = 4

    def __post_init__(self):
        assert 0.0 <= self.lazy_threshold <= 1.0, "lazy_threshold should be in [0, 1]"
        self.compression_ratios = []

    def lazy_score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, position_embeddings) -> torch.Tensor:
        """Attention mass of the last queries on the initial and recent keys, a 0-dim float32 tensor (:49-56)."""
        W = self.n_last
        cos, sin = position_embeddings
        q_pre = get_prerope_query_states(module, hidden_states[:, -W:])
        sc = _native.snapkv_score_rope(q_pre, cos[:, -W:], sin[:, -W:], keys, 1)[..., :-W]   # [B, H, S - W]: mean over window and group
        w = sc.mean(dim=(0, 1))
        return w[: self.n_initial].sum() + w[-self.n_recent:].sum()

    def is_lazy(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, position_embeddings) -> bool:
        return self.lazy_score(module, hidden_states, keys, position_embeddings).item() > self.lazy_threshold

    @property
    def compression_ratio(self):
        if len(self.compression_ratios) > 0:
            return sum(self.compression_ratios) / len(self.compression_ratios)
        raise ValueError("Forward pass must be run to compute the compression ratio")

    @compression_ratio.setter
    def compression_ratio(self, value):
        raise AttributeError(f"compression ratio cannot be set for {type(self).__name__}")

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if module.layer_idx == 0:
            self.compression_ratios = []
        B, H, k_len, _ = keys.shape
        min_length = self.n_initial + self.n_recent + self.n_last
        if k_len <= min_length:
            logger.warning(f"Sequence length is shorter than {min_length}: no compression applied")
        if self.lazy_threshold == 1.0 or k_len <= min_length:
            self.compression_ratios.append(0.0)
            return keys, values
        if self.is_lazy(module, hidden_states, keys, kwargs["position_embeddings"]):
            # keys[:, :, :n_initial] + keys[:, :, -n_recent + n_last:]  (:104-105)
            tail_start = k_len - self.n_recent + self.n_last
            pos = torch.cat([torch.arange(self.n_initial, device=keys.device), torch.arange(tail_start, k_len, device=keys.device)])
            keys, values = _native.gather_kv(keys, values, pos.to(torch.int32)[None, None, :].expand(B, H, -1).contiguous())
            self.compression_ratios.append((k_len - self.n_initial - self.n_recent + 1) / k_len)
        else:
            self.compression_ratios.append(0.0)
        return keys, values
