"""ScorerPress: score -> top-k -> gather, on the MI355X kernels.

Mirror of kvpress/presses/scorer_press.py (ScorerPress :17-102).  ``compress`` keeps the
reference semantics -- ratio 0 returns the inputs untouched (:86-87), ``n_kept =
int(k_len * (1 - ratio))`` in Python double arithmetic (:93-94), new contiguous [B,H,n_kept,D]
outputs in the input dtype, inputs never modified -- but top-k and gather are the HIP kernels
``kvp_topk_select`` / ``kvp_gather_kv`` (include/kvpress_hip.h).

Defined deviations (DESIGN.md "Parity contract"):
  * scores are float32 (the reference keeps the model dtype; bf16 scores tie ~1000-fold at the
    threshold and torch.topk's choice among ties is unspecified);
  * among equal scores the lowest position is kept;
  * the retained tokens are stored in ascending position order, the reference stores them in descending
    score order, which no reference test observes.  CONSEQUENCE for chains: a position-dependent press running AFTER a
    ScorerPress on the already-pruned cache (ComposedPress([Knorm, SnapKV / StreamingLLM / ExpectedAttention]),
    PrefillDecodingPress with such a decoding press) sees the survivors in position order here and in score order in the
    reference, so its "last W tokens" / sinks are different tokens: such chains are NOT reference-equivalent by default.
    ``kept_order = "score"`` (class attribute, settable per instance) stores the survivors in the reference's order
    (descending score, ties by position: ``KVP_ORDER_SCORE``) for reference-exact chaining; pinned by
    tests/test_host_hook.py::test_kept_order_switch.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress

logger = logging.getLogger(__name__)


@dataclass
class ScorerPress(BasePress):
    """Base class for score-based KV cache compression.

    Parameters
    ----------
    compression_ratio : float, default=0.0
        Fraction of key-value pairs to remove during compression.
    """

    compression_ratio: float = 0.0
    kept_order = "position"   # "position" (ascending, default) | "score" (the reference's torch.topk order); not a dataclass field

    def __post_init__(self):
        assert 0 <= self.compression_ratio < 1, "Compression ratio must be between 0 and 1"

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        """Importance score per KV pair, shape [B, num_kv_heads, seq_len]; higher = keep
        (scorer_press.py:35-74)."""
        raise NotImplementedError

    def n_kept(self, module: nn.Module, k_len: int) -> int:
        """Tokens kept per head: ``int(k_len * (1 - compression_ratio))`` in Python double arithmetic
        (scorer_press.py:93-94); presses with per-layer budgets override this."""
        return int(k_len * (1 - self.compression_ratio))

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if self.compression_ratio == 0:
            return keys, values

        scores = self.score(module, hidden_states, keys, values, attentions, kwargs)
        order = _native.ORDER_SCORE if self.kept_order == "score" else _native.ORDER_POSITION
        indices = _native.topk_select(scores, self.n_kept(module, keys.shape[2]), order)  # int32 [B,H,n_kept]
        return _native.gather_kv(keys, values, indices)                             # contiguous [B,H,n_kept,D]
