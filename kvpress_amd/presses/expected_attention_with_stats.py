"""ExpectedAttentionStatsPress (kvpress/presses/expected_attention_with_stats.py:22-106): ExpectedAttentionPress with
query statistics that were computed OFFLINE on a calibration set instead of on the prompt's own hidden states -- so the
4.4 TFLOP query projection of the whole prompt disappears and only ``kvp_ea_score`` runs per layer.

``mu`` [num_layers, H_q, D] and ``cov`` [num_layers, H_q, D, D] come from the Hugging Face hub (``stats_folder`` or the
id derived from the model and the calibration settings; needs network access) or are assigned directly.  The script that
computes and uploads new statistics (the reference's ``main`` / ``collect_queries``) is not part of this package."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.expected_attention_press import ExpectedAttentionPress


def _stats_class():
    from huggingface_hub import PyTorchModelHubMixin

    class ExpectedAttentionStats(torch.nn.Module, PyTorchModelHubMixin):
        """Container of the published statistics (expected_attention_with_stats.py:108-139)."""

        def __init__(self, num_layers: int, num_heads: int, head_dim: int, dataset_name: str, model_name: str, num_samples: int,
                     sample_seq_len: int, n_sink: int):
            super().__init__()
            self.query_mean = torch.nn.Parameter(torch.zeros(num_layers, num_heads, head_dim))
            self.query_cov = torch.nn.Parameter(torch.zeros(num_layers, num_heads, head_dim, head_dim))
            self.dataset_name, self.model_name, self.num_samples = dataset_name, model_name, num_samples
            self.sample_seq_len, self.n_sink = sample_seq_len, n_sink

        def stats_id(self) -> str:
            return (f"alessiodevoto/exp_att_stats_{self.model_name.replace('/', '_')}_{self.dataset_name.replace('/', '_')}_"
                    f"{self.num_samples}_{self.sample_seq_len}_{self.n_sink}")

    return ExpectedAttentionStats


@dataclass
class ExpectedAttentionStatsPress(ExpectedAttentionPress):
    """Parameters as ExpectedAttentionPress, plus the calibration settings that identify the published statistics:

    dataset_name : str, default="kmfoda/booksum"
    num_samples : int, default=100
    sample_seq_len : int, default=1000
    stats_folder : str, optional
        Load the statistics from this hub id / folder instead of the derived id.
    """

    sample_seq_len: int = 1000
    num_samples: int = 100
    dataset_name: str = "kmfoda/booksum"
    stats_folder: Optional[str] = None
    mu: torch.Tensor = field(init=False, default=None)
    cov: torch.Tensor = field(init=False, default=None)

    def get_query_statistics(self, module: nn.Module, hidden_states: torch.Tensor):
        """The layer's stored statistics through the averaged RoPE (:58-65); one set for every batch element."""
        q_len = hidden_states.shape[1]
        i = module.layer_idx
        dev = hidden_states.device   # statistics assigned by hand may still sit on the host
        mu, cov = self.apply_avg_rope(module, self.mu[i].to(dev, torch.float32), self.cov[i].to(dev, torch.float32) if self.use_covariance else None, q_len)
        return mu.unsqueeze(0), (cov.unsqueeze(0) if cov is not None else None)

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        assert keys.size(2) > self.n_sink, f"Input should contain more tokens than n_sink={self.n_sink}"
        mu, cov = self.get_query_statistics(module, hidden_states)
        B = keys.shape[0]
        return _native.ea_score(keys, values, mu.expand(B, -1, -1), cov.expand(B, -1, -1, -1) if cov is not None else None, self.n_sink,
                                self.use_vnorm, self.epsilon)

    @staticmethod
    def available_stats():
        from huggingface_hub import get_collection

        return [x.item_id for x in get_collection("alessiodevoto/expectedattentionstats-68b0248d519303713320e2cf").items]

    def post_init_from_model(self, model):
        """Load the statistics unless they were assigned (:72-82)."""
        if self.mu is None and self.cov is None:
            stats = _stats_class().from_pretrained(self.stats_folder) if self.stats_folder is not None else self._maybe_load_stats_from_hub(model)
            self.mu = stats.query_mean.data.to(model.device, dtype=model.dtype)
            self.cov = stats.query_cov.data.to(model.device, dtype=model.dtype)

    def _maybe_load_stats_from_hub(self, model):
        cls = _stats_class()
        cfg = model.config
        stats_id = cls(model_name=cfg.name_or_path, num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads, head_dim=cfg.head_dim,
                       dataset_name=self.dataset_name, num_samples=self.num_samples, sample_seq_len=self.sample_seq_len, n_sink=self.n_sink).stats_id()
        try:
            return cls.from_pretrained(stats_id)
        except ValueError:
            raise ValueError(f"No statistics found for model {stats_id} on the Hub. Please compute them first "
                             "(the reference's expected_attention_with_stats.py --model_name <model_name>).")
