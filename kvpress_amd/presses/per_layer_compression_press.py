"""PerLayerCompressionPress (kvpress/presses/per_layer_compression_press.py:19-69): one ScorerPress, one ratio per layer."""
from __future__ import annotations

import inspect
import logging
from dataclasses import dataclass
from typing import List

import torch
from torch import nn

from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.scorer_press import ScorerPress

logger = logging.getLogger(__name__)


@dataclass
class PerLayerCompressionPress(BasePress):
    """Layer ``i`` is compressed with ``compression_ratios[i]`` (the wrapped press's own ratio is restored after every
    hook call).  Experimental in the reference: layers end up with different cache lengths, which needs an attention
    implementation that takes the per-layer K/V length from the tensors (flash / sdpa).

    Parameters
    ----------
    press : ScorerPress
    compression_ratios : List[float]
    """

    press: ScorerPress
    compression_ratios: List[float]

    def __post_init__(self):
        logger.warning("Per layer compression wrapper is an experimental feature and only works with flash attention. "
                       "Please make sure that the model uses flash attention.")
        assert "compression_ratio" in inspect.signature(self.press.__init__).parameters, \
            f"compression_ratio can't be set in the provided press: {self.press.__class__}"
        assert isinstance(self.press, ScorerPress), "PerLayerCompressionPress requires a ScorerPress as input"

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    def forward_hook(self, module: nn.Module, input: list[torch.Tensor], kwargs: dict, output: list):
        saved = self.press.compression_ratio
        self.press.compression_ratio = self.compression_ratios[module.layer_idx]
        try:
            return self.press.forward_hook(module, input, kwargs, output)
        finally:
            self.press.compression_ratio = saved

    @property
    def compression_ratio(self):
        return sum(self.compression_ratios) / len(self.compression_ratios)

    @compression_ratio.setter
    def compression_ratio(self, value):
        raise AttributeError(f"compression ratio cannot be set for {type(self).__name__}")
