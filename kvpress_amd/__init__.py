"""kvpress_amd: MI355X-native score -> top-k -> gather hot path of NVIDIA/kvpress.

The public API mirrors the reference's (same class names, dataclass fields and method signatures): BasePress / ScorerPress,
the three core scorers (KnormPress, SnapKVPress, ExpectedAttentionPress), the scorers and wrappers that build on the same
kernels (see __all__ and DESIGN.md section 8) and the "kv-press-text-generation" pipeline, registered on import like the
reference's.  Everything below ``compress()`` runs in hand-written HIP kernels (gfx950) reached through the C ABI of
include/kvpress_hip.h; there is no CPU or pure-PyTorch fallback.
"""
from kvpress_amd.presses.adakv_press import AdaKVPress
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.block_press import BlockPress
from kvpress_amd.presses.chunk_press import ChunkPress
from kvpress_amd.presses.chunkkv_press import ChunkKVPress
from kvpress_amd.presses.composed_press import ComposedPress
from kvpress_amd.presses.criticalkv_press import CriticalAdaKVPress, CriticalKVPress
from kvpress_amd.presses.cur_press import CURPress
from kvpress_amd.presses.dms_press import DMSPress
from kvpress_amd.presses.duo_attention_press import DuoAttentionPress
from kvpress_amd.presses.decoding_press import CompressionRatioDecodingPress, DecodingPress, PrefillDecodingPress
from kvpress_amd.presses.expected_attention_press import ExpectedAttentionPress
from kvpress_amd.presses.expected_attention_with_stats import ExpectedAttentionStatsPress
from kvpress_amd.presses.finch_press import FinchPress
from kvpress_amd.presses.key_rerotation_press import KeyRerotationPress
from kvpress_amd.presses.keydiff_press import KeyDiffPress
from kvpress_amd.presses.knorm_press import KnormPress
from kvpress_amd.presses.lagkv_press import LagKVPress
from kvpress_amd.presses.observed_attention_press import ObservedAttentionPress
from kvpress_amd.presses.per_layer_compression_press import PerLayerCompressionPress
from kvpress_amd.presses.pyramidkv_press import PyramidKVPress
from kvpress_amd.presses.qfilter_press import QFilterPress
from kvpress_amd.presses.random_press import RandomPress
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.presses.simlayerkv_press import SimLayerKVPress
from kvpress_amd.presses.snapkv_press import SnapKVPress
from kvpress_amd.presses.streaming_llm_press import StreamingLLMPress
from kvpress_amd.presses.think_press import ThinKPress
from kvpress_amd.presses.tova_press import TOVAPress

__version__ = "0.1.0"
__all__ = ["BasePress", "ScorerPress", "KnormPress", "SnapKVPress", "ExpectedAttentionPress", "ExpectedAttentionStatsPress", "PyramidKVPress", "TOVAPress",
           "KeyDiffPress", "LagKVPress", "QFilterPress", "ObservedAttentionPress", "CURPress", "StreamingLLMPress", "SimLayerKVPress", "ThinKPress", "RandomPress", "ChunkPress", "ChunkKVPress", "BlockPress", "KeyRerotationPress", "FinchPress", "AdaKVPress", "CriticalKVPress", "CriticalAdaKVPress", "DMSPress", "DuoAttentionPress", "ComposedPress", "PerLayerCompressionPress", "DecodingPress",
           "CompressionRatioDecodingPress", "PrefillDecodingPress", "KVPressTextGenerationPipeline"]


# importing the package registers the "kv-press-text-generation" task, as `import kvpress` does (kvpress/__init__.py, pipeline.py:326-331)
from kvpress_amd.pipeline import KVPressTextGenerationPipeline  # noqa: E402
