"""KeyRerotationPress (kvpress/presses/key_rerotation_press.py:17-162): prune with a ScorerPress, then re-rotate the kept
keys so that they carry the RoPE phases of positions 0..n_kept-1 (as StreamingLLM does in the paper).

score (wrapped press) -> kvp_topk_select (ascending positions = the reference's ``torch.sort(indices)``, :157) ->
kvp_gather_kv_rerotate (the gather and the re-rotation in one pass: the kept keys are written once, already rotated)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class KeyRerotationPress(BasePress):
    """Parameters
    ----------
    press : ScorerPress
        Decides which tokens are kept; the rerotation is applied afterwards.
    """

    press: ScorerPress

    def __post_init__(self):
        assert isinstance(self.press, ScorerPress)

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
        scores = self.press.score(module, hidden_states, keys, values, attentions, kwargs)
        n_kept = int(keys.shape[2] * (1 - self.press.compression_ratio))      # key_rerotation_press.py:154-155
        indices = _native.topk_select(scores, n_kept)                         # ascending positions (:157)
        return _native.gather_kv_rerotate(keys, values, indices, module.rotary_emb.inv_freq)   # :157-160 + :107-128 in one pass
