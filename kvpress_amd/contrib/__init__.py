"""kvpress_amd.contrib: presses OUTSIDE the hot-path scope of this package.

Built in round 1, kept because they have kernels and reference-made goldens, not extended since: LagKVPress, ObservedAttentionPress,
ThinKPress, ChunkKVPress, BlockPress.  Their entry points are declared in include/kvpress_hip_extra.h, not in the boundary header.
Nothing in ``kvpress_amd`` proper imports this sub-package.
"""
from kvpress_amd.contrib.block_press import BlockPress
from kvpress_amd.contrib.chunkkv_press import ChunkKVPress
from kvpress_amd.contrib.lagkv_press import LagKVPress
from kvpress_amd.contrib.observed_attention_press import ObservedAttentionPress
from kvpress_amd.contrib.think_press import ThinKPress

__all__ = ["BlockPress", "ChunkKVPress", "LagKVPress", "ObservedAttentionPress", "ThinKPress"]
