"""StreamingLLMPress (kvpress/presses/streaming_llm_press.py:15-54): sinks + most recent tokens."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class StreamingLLMPress(ScorerPress):
    """StreamingLLM (https://arxiv.org/abs/2309.17453): keep the first ``n_sink`` tokens and the most recent ones.

    The score is a 0/1 mask built on the host side (no arithmetic on K or V): ones everywhere, zeros on the
    ``n_pruned`` positions after the sinks; selection and gather run in kvp_topk_select / kvp_gather_kv like for any
    other ScorerPress (exactly ``n_kept`` ones exist, so ties never decide anything).

    Parameters
    ----------
    compression_ratio : float, default=0.0
    n_sink : int, default=4
    """

    compression_ratio: float = 0.0
    n_sink: int = 4

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        k_len = keys.shape[2]
        assert k_len > self.n_sink, f"Input should contain more tokens than n_sink={self.n_sink}"
        n_pruned = k_len - int(k_len * (1 - self.compression_ratio))
        scores = torch.ones(keys.shape[:-1], dtype=torch.float32, device=keys.device)
        scores[:, :, self.n_sink: self.n_sink + n_pruned] = 0
        return scores
