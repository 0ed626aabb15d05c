"""ChunkKVPress (kvpress/presses/chunkkv_press.py:14-125): whole CHUNKS of tokens are kept or dropped.

Global scores from the wrapped ScorerPress (library), per-chunk score = mean over the chunk of the head-summed scores (a
[B, S] -> [B, n_chunks] reduction, torch glue on a tiny tensor), ``kvp_topk_select`` over the chunk scores, then ONE
``kvp_gather_kv`` with the positions of the kept chunks -- the same positions for every head and batch element, taken
from batch element 0 exactly as the reference does (:104)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class ChunkKVPress(BasePress):
    """ChunkKV (https://arxiv.org/abs/2502.00299): semantic-preserving, chunk-wise token selection.

    Parameters
    ----------
    press : ScorerPress
        Computes the global importance scores.
    chunk_length : int, default=20
    """

    press: ScorerPress
    chunk_length: int = 20

    def __post_init__(self):
        assert isinstance(self.press, ScorerPress), "ChunkKVPress requires a ScorerPress as input"

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    @property
    def compression_ratio(self):
        return self.press.compression_ratio

    @compression_ratio.setter
    def compression_ratio(self, value):
        self.press.compression_ratio = value

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if self.press.compression_ratio == 0:
            return keys, values
        assert attentions is None, "ChunkPress does not support attentions."
        B, H, kv_len, _ = keys.shape
        L = self.chunk_length
        n_full, tail = divmod(kv_len, L)
        if n_full == 0:   # no complete chunk: the wrapped press decides alone (:77-78)
            return self.press.compress(module, hidden_states, keys, values, attentions, kwargs)
        scores = self.press.score(module, hidden_states, keys, values, attentions, kwargs).to(torch.float32)
        per_token = scores.sum(dim=1)                                                    # [B, S]  (:82-83)
        chunk_scores = per_token[:, : n_full * L].view(B, n_full, L).mean(dim=-1)
        if tail:
            chunk_scores = torch.cat([chunk_scores, per_token[:, -tail:].mean(dim=-1, keepdim=True)], dim=-1)   # (:89-92)
        n_chunks = n_full + (1 if tail else 0)
        n_kept = max(1, int(n_chunks * (1 - self.press.compression_ratio)))             # (:97)
        top = _native.topk_select(chunk_scores[:1].contiguous(), n_kept)[0].long()      # chunk ids of batch element 0, ascending
        pos = (top[:, None] * L + torch.arange(L, device=keys.device)[None, :]).flatten()
        pos = pos[pos < kv_len]                                                          # the partial last chunk (:110-112)
        indices = pos.to(torch.int32)[None, None, :].expand(B, H, -1).contiguous()
        return _native.gather_kv(keys, values, indices)
