"""SnapKVPress (kvpress/presses/snapkv_press.py:16-105) on kvp_snapkv_score."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.utils import get_prerope_query_states


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


@dataclass
class SnapKVPress(ScorerPress):
    """SnapKV (https://arxiv.org/abs/2404.14469): the attention the last ``window_size`` queries
    pay to the earlier keys estimates their importance.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    window_size : int, default=64
        Number of recent tokens whose queries score the earlier keys (never pruned themselves).
    kernel_size : int, default=5
        Size of the average-pooling kernel smoothing the scores along the sequence.
    """

    compression_ratio: float = 0.0
    window_size: int = 64
    kernel_size: int = 5

    @staticmethod
    def compute_window_queries(module, hidden_states, window_size, position_embeddings) -> torch.Tensor:
        """RoPE'd queries of the last ``window_size`` tokens, [B, Hq, W, D] (snapkv_press.py:53-58).
        q_proj is a model-owned nn.Linear (64 x hidden GEMM) and stays a torch call."""
        query_states = get_prerope_query_states(module, hidden_states[:, -window_size:])
        cos, sin = position_embeddings
        cos, sin = cos[:, -window_size:], sin[:, -window_size:]
        return (query_states * cos.unsqueeze(1)) + (_rotate_half(query_states) * sin.unsqueeze(1))

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        k_len = keys.shape[2]
        assert (
            hidden_states.shape[1] > self.window_size
        ), f"Query length {hidden_states.shape[1]} should be greater than the window size {self.window_size}"

        if attentions is not None:
            attn = attentions[..., -self.window_size:, : -self.window_size]  # snapkv_press.py:88-89
            return _native.snapkv_score_from_attn(attn, keys.shape[1], k_len, self.kernel_size)
        cos, sin = kwargs["position_embeddings"]
        cos, sin = cos[:, -self.window_size:], sin[:, -self.window_size:]
        if _native.qproj_rope_supported(module, hidden_states, self.window_size):
            # plain bf16 / f16 nn.Linear q_proj: projection of the last W tokens + RoPE in one library kernel
            return _native.snapkv_score_hidden(hidden_states[:, -self.window_size:], module.q_proj.weight, cos, sin, keys, self.kernel_size)
        # otherwise q_proj stays the model's own call (quantised / LoRA / q_norm ...); RoPE + the rest in the library
        q_pre = get_prerope_query_states(module, hidden_states[:, -self.window_size:])
        return _native.snapkv_score_rope(q_pre, cos, sin, keys, self.kernel_size)

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        """ScorerPress.compress (scorer_press.py:76-102) as ONE library call when the scores come from the window
        queries (``attentions is None``): RoPE, two attention passes, pooling + first select pass, select, gather; the
        window tokens are appended to the selection instead of being scored.  Given attention weights, or a
        subclass with its own ``score``, the generic three-call sequence runs."""
        if self.compression_ratio == 0:
            return keys, values
        if attentions is not None or type(self).score is not SnapKVPress.score:
            return super().compress(module, hidden_states, keys, values, attentions, kwargs)
        order = _native.ORDER_SCORE if self.kept_order == "score" else _native.ORDER_POSITION   # "score": the reference's row order
        assert (
            hidden_states.shape[1] > self.window_size
        ), f"Query length {hidden_states.shape[1]} should be greater than the window size {self.window_size}"
        cos, sin = kwargs["position_embeddings"]
        cos, sin = cos[:, -self.window_size:], sin[:, -self.window_size:]
        n_kept = self.n_kept(module, keys.shape[2])
        if _native.qproj_rope_supported(module, hidden_states, self.window_size):
            return _native.snapkv_compress_hidden(hidden_states[:, -self.window_size:], module.q_proj.weight, cos, sin, keys, values,
                                                  self.kernel_size, n_kept, order)
        q_pre = get_prerope_query_states(module, hidden_states[:, -self.window_size:])
        return _native.snapkv_compress_rope(q_pre, cos, sin, keys, values, self.kernel_size, n_kept, order)
