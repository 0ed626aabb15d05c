"""AdaKVPress (kvpress/presses/adakv_press.py:14-78): head-wise budgets -- the pruned tokens are the lowest scores ACROSS
the heads of a layer, with a per-head safeguard.

On the library: ``score`` (wrapped press) -> ``kvp_topk_select`` per head (the n_safe best, protected with
``kvp_scores_fill_at``) -> ``kvp_topk_select`` with ``KVP_TOPK_SMALLEST`` over the flattened ``[B, H * S]`` rows (the
cross-head bottom-k).  K and V stay in the cache; the pruned (batch, head, position) triples go to
``module.masked_key_indices`` and are masked during decoding by kvpress_amd.attention_patch."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.attention_patch import patch_attention_functions
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class AdaKVPress(BasePress):
    """AdaKV (https://arxiv.org/abs/2407.11550).

    Parameters
    ----------
    press : ScorerPress
    alpha_safeguard : float, default=0.20
        Every head keeps at least this fraction of its ``n_kept`` best tokens.
    """

    press: ScorerPress
    alpha_safeguard: float = 0.20

    def __post_init__(self):
        assert isinstance(self.press, ScorerPress), "AdaKVPress requires a ScorerPress as input"
        assert 0 <= self.alpha_safeguard <= 1, "alpha_safeguard should be in [0, 1]"
        patch_attention_functions()

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
        if self.compression_ratio == 0:
            return keys, values
        assert module.config._attn_implementation != "eager", "eager mode not supported"

        scores = self.press.score(module, hidden_states, keys, values, attentions, kwargs)
        scores = scores.to(torch.float32).contiguous()
        if scores.data_ptr() in (keys.data_ptr(), values.data_ptr()):
            scores = scores.clone()
        bsz, num_kv, k_len = scores.shape

        n_kept = int(k_len * (1 - self.compression_ratio))          # ScorerPress definition
        n_safe = int(n_kept * self.alpha_safeguard)                 # adakv_press.py:60
        if n_safe:
            _native.scores_fill_at_(scores, _native.topk_select(scores, n_safe), torch.finfo(torch.float32).max)

        n_pruned = num_kv * (k_len - n_kept)                        # :66
        flat = _native.topk_select(scores.view(bsz, num_kv * k_len), n_pruned, _native.ORDER_POSITION | _native.TOPK_SMALLEST)
        flat = flat.to(torch.int64).flatten()
        batch_indices = torch.arange(bsz, device=flat.device).repeat_interleave(n_pruned)
        module.masked_key_indices = (batch_indices, flat // k_len, flat % k_len)
        return keys, values
