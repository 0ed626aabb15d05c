"""DuoAttentionPress (kvpress/presses/duo_attention_press.py:31-130): retrieval heads keep the whole cache, streaming heads
only the first ``sink_size`` and the last ``recent_size`` tokens.

Pure host logic: the streaming heads' middle tokens become ``module.masked_key_indices`` and are masked during attention
by kvpress_amd.attention_patch (as for AdaKVPress); nothing is scored, so no kernel of the library runs.  The head scores
come from the published attention patterns (network) -- or from a subclass overriding ``load_attention_pattern``, which
is how the reference's own tests (and ours) run offline.  The experimental on-the-fly scoring of the reference needs the
BookSum dataset and is not provided."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from kvpress_amd.attention_patch import patch_attention_functions
from kvpress_amd.presses.base_press import BasePress

def _pattern_dirs() -> dict:
    """checkpoint name -> directory of its published attention pattern in the DuoAttention repository
    (duo_attention_press.py:18-25): ``<model>/lr=0.02-reg=0.05-ctx=1000_<max ctx>-multi_passkey10`` (the older entries
    are stored URL-encoded there)."""
    table = [("togethercomputer/", "Llama-2-7B-32K-Instruct", 32000, True), ("gradientai//", "Llama-3-8B-Instruct-Gradient-1048k", 32000, True),
             ("gradientai//", "Llama-3-8B-Instruct-Gradient-4194k", 32000, True), ("meta-llama/", "Meta-Llama-3.1-8B-Instruct", 128000, False),
             ("mistralai/", "Mistral-7B-Instruct-v0.2", 32000, True), ("mistralai/", "Mistral-7B-Instruct-v0.3", 32000, True)]
    out = {}
    for org, model, ctx, encoded in table:
        eq = "%3D" if encoded else "="
        out[org + model] = f"{model}/lr{eq}0.02-reg{eq}0.05-ctx{eq}1000_{ctx}-multi_passkey10"
    return out


PATTERNS_DICT = _pattern_dirs()


_PATTERN_CACHE: dict = {}   # checkpoint name -> (sink_size, recent_size, head_scores)


@dataclass
class DuoAttentionPress(BasePress):
    """DuoAttention (https://arxiv.org/abs/2410.10819).

    Parameters
    ----------
    head_compression_ratio : float, default=0.0
        Fraction of the (layer, kv-head) pairs that become streaming heads (lowest retrieval scores first).
    """

    head_compression_ratio: float = 0.0
    compression_ratio_: float = field(init=False, default=None)
    recent_size: int = field(init=False, default=None)
    sink_size: int = field(init=False, default=None)
    streaming_mask: torch.Tensor = field(init=False, default=None)

    def __post_init__(self):
        patch_attention_functions()

    def post_init_from_model(self, model):
        """sink_size, recent_size and the streaming mask from the model's attention pattern (:72-85)."""
        self.sink_size, self.recent_size, head_scores = self.load_attention_pattern(model)
        n_pruned = round(head_scores.size * self.head_compression_ratio)
        self.streaming_mask = torch.zeros(head_scores.shape, dtype=bool, device=model.device)
        if n_pruned > 0:
            indices = np.argsort(head_scores, axis=None)[:n_pruned]
            self.streaming_mask[np.unravel_index(indices, head_scores.shape)] = True

    @property
    def compression_ratio(self) -> float:
        assert self.compression_ratio_ is not None, "Forward pass must be run to compute the compression ratio"
        return self.compression_ratio_

    @compression_ratio.setter
    def compression_ratio(self, value):
        raise AttributeError(f"compression ratio cannot be set for {type(self).__name__}")

    def compress(self, module, hidden_states, keys, values, attentions, kwargs):
        assert module.config._attn_implementation != "eager", "eager mode not supported"
        if self.streaming_mask is None:
            raise ValueError("Streaming mask not initialized. Make sure to call post_init_from_model to initialize this press.")
        k_len = keys.shape[2]
        if (self.head_compression_ratio > 0) or (k_len > (self.sink_size + self.recent_size)):
            masked_keys = torch.zeros_like(keys[..., 0], dtype=torch.bool)
            masked_keys[:, self.streaming_mask[module.layer_idx], self.sink_size: -self.recent_size] = True
            module.masked_key_indices = torch.nonzero(masked_keys, as_tuple=True)
        self.compression_ratio_ = self.streaming_mask.float().mean().item()
        self.compression_ratio_ *= 1 - (self.sink_size + self.recent_size) / k_len
        return keys, values

    @staticmethod
    def load_attention_pattern(model):
        """(sink_size, recent_size, head_scores [n_layers, n_kv_heads]) from the DuoAttention repository (:105-122); fetched once per
        checkpoint name (the reference memoises with ``@cached``), 30 s timeout per request.  The reference's
        ``on_the_fly_scoring`` (calibrating the patterns on the spot) is not part of this package."""
        name = model.config.name_or_path
        assert name in PATTERNS_DICT, f"Checkpoint {name} not in {list(PATTERNS_DICT.keys())}"
        if name not in _PATTERN_CACHE:
            import json
            from io import StringIO
            from urllib.request import urlopen

            url = f"https://raw.githubusercontent.com/mit-han-lab/duo-attention/refs/heads/main/attn_patterns/{PATTERNS_DICT[name]}/"
            config = json.loads(urlopen(url + "config.json", timeout=30).read().decode())
            text = urlopen(url + "full_attention_heads.tsv", timeout=30).read().decode()
            head_scores = np.clip(np.loadtxt(StringIO(text), dtype=float, delimiter="\t"), 0, 1)
            _PATTERN_CACHE[name] = (config["sink_size"], config["recent_size"], head_scores)
        return _PATTERN_CACHE[name]
