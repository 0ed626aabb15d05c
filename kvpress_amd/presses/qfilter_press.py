"""QFilterPress (kvpress/presses/qfilter_press.py:13-83) on kvp_rowdot_score."""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import cache

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress


def _qfilters_class():
    from huggingface_hub import PyTorchModelHubMixin

    class QFilters(torch.nn.Module, PyTorchModelHubMixin):
        """The published container of the learned filters: one vector per (layer, kv-head) (qfilter_press.py:13-16)."""

        def __init__(self, num_layers: int, num_kv_heads: int, kv_head_dim: int):
            super().__init__()
            self.q_filters = torch.nn.Parameter(torch.randn(num_layers, num_kv_heads, kv_head_dim))

    return QFilters


@dataclass
class QFilterPress(ScorerPress):
    """Q-Filter (https://arxiv.org/abs/2503.02812): keys are scored by their projection on a learned, model-specific
    direction per (layer, kv-head); the lowest responses are pruned.

    The filters are fetched from the Hugging Face hub by model name in ``post_init_from_model`` (needs network access or a
    populated hub cache); ``q_filters`` may also be assigned directly as a ``[num_layers, num_kv_heads, head_dim]`` tensor.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    """

    q_filters: torch.Tensor = field(init=False, default=None)

    def post_init_from_model(self, model):
        model_name = model.config.name_or_path.split("/")[-1]
        self.q_filters = self.load_q_filters(model_name)
        self.q_filters = self.q_filters.to(model.dtype)

    @staticmethod
    @cache
    def load_q_filters(model_name):
        model_name = model_name if "Meta-Llama-3.1-405B" in model_name else model_name.replace("Meta-Llama", "Llama")
        try:
            return _qfilters_class().from_pretrained(f"nthngdy/{model_name}_qfilt").q_filters
        except TypeError:
            raise ValueError(f"Could not load Q-filters for {model_name}. Available models: {QFilterPress.available_qfilters()}")

    @staticmethod
    def available_qfilters():
        from huggingface_hub import get_collection

        collection = get_collection("nthngdy/q-filters-67a4994dcb302a3d37f3d119", token=False)
        return [x.item_id.split("/")[-1][:-6] for x in collection.items]

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        if self.q_filters is None:
            raise ValueError("Q-filters not loaded. If you are using a wrapper press, make sure to call post_init_from_model.")
        return _native.rowdot_score(keys, self.q_filters[module.layer_idx], -1.0)
