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
    reference, so its "last W tokens" / sinks would be different tokens.
    ``kept_order = "score"`` (class attribute, settable per instance) stores the survivors in the reference's order
    (descending score, ties by position: ``KVP_ORDER_SCORE``); pinned by tests/test_host_hook.py::test_kept_order_switch.
    ComposedPress / PrefillDecodingPress switch the earlier presses of such a chain to it automatically (round 6:
    ``resolve_chain_kept_order``), so chains ARE reference-equivalent unless an instance was explicitly set to "position".
"""
from __future__ import annotations

import logging
from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress

logger = logging.getLogger(__name__)

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


def _order_blind(p) -> bool:
    """row scorers whose result does not depend on where a token sits in the (already pruned) cache"""
    from kvpress_amd.presses.cur_press import CURPress
    from kvpress_amd.presses.keydiff_press import KeyDiffPress
    from kvpress_amd.presses.knorm_press import KnormPress
    from kvpress_amd.presses.qfilter_press import QFilterPress

    if isinstance(p, CURPress):     # sinks and the local normalisation windows are positional
        return p.num_sinks == 0 and not (p.use_local_approximation and p.local_window_size)
    return isinstance(p, (KnormPress, KeyDiffPress, QFilterPress))


def resolve_chain_kept_order(first, later, where: str):
    """A press that looks at token ORDER ("last W tokens", sinks, chunks, re-rotation, recency: anything but the order-blind row scorers
    Knorm / KeyDiff / QFilter / CUR without sinks and windows) running on a cache that an earlier ScorerPress has pruned sees the
    survivors in descending SCORE order in the reference (scorer_press.py:95-100; composed_press.py:56-62).  This package's default is
    ascending position order, so such a chain would keep different tokens.  Round 6: the chain makes
    itself reference-identical -- every earlier ScorerPress whose ``kept_order`` was not set on the INSTANCE is switched to "score" (the
    fused score-order path, +14 % on that press only) and an info line says so; an instance the user explicitly set to "position" is
    respected and gets a warning per chain object (not per process).  Returns "switched", "warned" or None (tests)."""
    pruners = [p for p in _wrapped(first) if isinstance(p, ScorerPress) and p.kept_order == "position" and p.compression_ratio != 0]
    dependents = [p for q in later for p in _wrapped(q)]
    dependents = [p for p in dependents if not _order_blind(p) and type(p).__name__ not in ("ComposedPress", "PerLayerCompressionPress", "DecodingPress")]
    if not pruners or not dependents:
        return None
    explicit = [p for p in pruners if "kept_order" in vars(p)]
    for p in pruners:
        if "kept_order" not in vars(p):
            p.kept_order = "score"
    if explicit:
        logger.warning(
            "%s: %s runs on a cache that %s has already pruned with kept_order='position' (set on the instance: survivors in ascending "
            "position order). The reference stores them in descending score order, so this order-dependent press sees different 'last' / "
            "'first' tokens than in NVIDIA/kvpress; set kept_order='score' (or leave it unset) for reference-identical chains.",
            where, type(dependents[0]).__name__, type(explicit[0]).__name__)
        return "warned"
    logger.info("%s: %s now keeps its survivors in the reference's order (kept_order='score'): %s behind it depends on token order.",
                where, type(pruners[0]).__name__, type(dependents[0]).__name__)
    return "switched"


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
