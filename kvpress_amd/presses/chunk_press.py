"""ChunkPress (kvpress/presses/chunk_press.py:17-100): a ScorerPress applied to every chunk of the sequence on its own.

The reference loops over the chunks in Python (one ``score`` + ``topk`` per chunk, :67-81).  Here all full chunks of a
batch element are scored in ONE call of the wrapped press -- the chunks become its batch dimension through strided
views (``keys[b, :, c*L:(c+1)*L]`` is element ``c`` of a ``[n_chunks, H, L, D]`` view; no copy: the kernels take
element strides) -- and selected in ONE segmented top-k (``kvp_topk_select_segmented``).  A shorter last chunk takes one
more call each.  Scores per chunk are what the reference computes per chunk (the wrapped press never looks across its
batch dimension, apart from SnapKV's pad constant, which only has to exceed the chunk's own maximum)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class ChunkPress(BasePress):
    """Uniform compression through independent chunk processing (FINCH).

    Parameters
    ----------
    press : ScorerPress
        The scoring method applied to each chunk.
    chunk_length : int, default=1024
    """

    press: ScorerPress
    chunk_length: int = 1024

    def __post_init__(self):
        assert isinstance(self.press, ScorerPress), "ChunkPress requires a ScorerPress as input"

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    @property
    def compression_ratio(self):
        return self.press.compression_ratio

    @compression_ratio.setter
    def compression_ratio(self, value):
        self.press.compression_ratio = value

    def _n_kept(self, length: int) -> int:
        return max(1, int(length * (1 - self.press.compression_ratio)))  # chunk_press.py:79

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if self.press.compression_ratio == 0:
            return keys, values
        assert attentions is None, "ChunkPress does not support attentions."

        B, H, S, D = keys.shape
        L = self.chunk_length
        n_full, tail = divmod(S, L)
        pe = kwargs.get("position_embeddings")
        rows = []
        for b in range(B):
            kw = kwargs
            if pe is not None and pe[0].shape[0] == B and B > 1:  # per-element rotary tables
                kw = dict(kwargs, position_embeddings=(pe[0][b:b + 1], pe[1][b:b + 1]))
            parts = []
            if n_full:
                # chunks as the batch dimension of the wrapped press: [n_full, H, L, D] / [n_full, L, hidden] views
                kc = keys[b, :, : n_full * L].unflatten(1, (n_full, L)).transpose(0, 1)
                vc = values[b, :, : n_full * L].unflatten(1, (n_full, L)).transpose(0, 1)
                hc = hidden_states[b, : n_full * L].unflatten(0, (n_full, L))
                sc = self.press.score(module, hc, kc, vc, None, kw)                   # [n_full, H, L]
                sc = sc.to(torch.float32).transpose(0, 1).reshape(H, n_full * L)      # rows = heads, chunks side by side
                parts.append(_native.topk_select_segmented(sc, L, self._n_kept(L)))
            if tail:
                s0 = n_full * L
                sc = self.press.score(module, hidden_states[b:b + 1, s0:], keys[b:b + 1, :, s0:], values[b:b + 1, :, s0:], None, kw)
                parts.append(_native.topk_select_segmented(sc[0].to(torch.float32), tail, self._n_kept(tail), pos_base=s0))
            rows.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1))
        indices = torch.stack(rows, dim=0)                                             # [B, H, n] int32, ascending
        return _native.gather_kv(keys, values, indices)
