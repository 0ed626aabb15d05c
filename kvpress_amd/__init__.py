"""kvpress_amd: MI355X-native score -> top-k -> gather hot path of NVIDIA/kvpress.

Public API mirrors the reference for this path (same class names, dataclass fields and
method signatures): BasePress, ScorerPress, KnormPress, SnapKVPress, ExpectedAttentionPress.
Everything below ``ScorerPress.compress`` runs in hand-written HIP kernels (gfx950) reached
through the C ABI of include/kvpress_hip.h; there is no CPU or pure-PyTorch fallback.
"""
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.expected_attention_press import ExpectedAttentionPress
from kvpress_amd.presses.knorm_press import KnormPress
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.presses.snapkv_press import SnapKVPress

__version__ = "0.1.0"
__all__ = ["BasePress", "ScorerPress", "KnormPress", "SnapKVPress", "ExpectedAttentionPress"]
