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

_ORDER_WARNED = False


def _wrapped(press):
    """`press` and every press it wraps (ChunkPress.press, DecodingPress.base_press, ComposedPress.presses, ...)"""
    out, todo = [], [press]
    while todo:
        p = todo.pop()
        if not isinstance(p, BasePress) or any(p is q for q in out):
            continue
        out.append(p)
        for name in ("press", "base_press", "prefilling_press", "decoding_press"):
            todo.append(getattr(p, name, None))
        todo += list(getattr(p, "presses", None) or [])
    return out


def warn_if_chain_depends_on_kept_order(first, later, where: str) -> bool:
    """One-time ``logger.warning`` when a press that looks at token ORDER (anything but the order-blind row scorers Knorm / KeyDiff /
    QFilter / CUR-without-sinks is treated as order-dependent: "last W tokens", sinks, chunks, re-rotation, recency) runs on a cache that
    an earlier ScorerPress pruned with ``kept_order = "position"``: the reference hands that press the survivors in descending SCORE
    order (scorer_press.py:95-100), this package in ascending position order, so the chain keeps different tokens than the reference
    unless ``kept_order = "score"`` is set on the earlier press (composed_press.py:56-62).  Returns whether the
    situation was detected (tests)."""
    global _ORDER_WARNED
    from kvpress_amd.presses.keydiff_press import KeyDiffPress
    from kvpress_amd.presses.knorm_press import KnormPress
    from kvpress_amd.presses.qfilter_press import QFilterPress

    order_blind = (KnormPress, KeyDiffPress, QFilterPress)
    pruners = [p for p in _wrapped(first) if isinstance(p, ScorerPress) and p.kept_order == "position" and p.compression_ratio != 0]
    dependents = [p for q in later for p in _wrapped(q)]
    dependents = [p for p in dependents if not isinstance(p, order_blind) and type(p).__name__ not in ("ComposedPress", "PerLayerCompressionPress", "DecodingPress")]
    if not pruners or not dependents:
        return False
    if not _ORDER_WARNED:
        _ORDER_WARNED = True
        logger.warning(
            "%s: %s runs on a cache that %s has already pruned with kept_order='position' (survivors in ascending position order). The "
            "reference stores them in descending score order, so this order-dependent press sees different 'last' / 'first' tokens than "
            "in NVIDIA/kvpress; set kept_order='score' on the earlier press for reference-identical chains. (Shown once.)",
            where, type(dependents[0]).__name__, type(pruners[0]).__name__)
    return True


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
