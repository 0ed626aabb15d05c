"""ExpectedAttentionPress (kvpress/presses/expected_attention_press.py:16-165) on
kvp_ea_qstats / kvp_ea_score."""
from __future__ import annotations

import weakref
from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.scorer_press import ScorerPress
from kvpress_amd.utils import get_prerope_query_states


@dataclass
class ExpectedAttentionPress(ScorerPress):
    """Expected attention: E[exp(q.k/sqrt(d))] for future queries q ~ N(mu, cov) rotated by the
    average RoPE of the next ``n_future_positions`` positions, rescaled by ||v||.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    n_future_positions : int, default=512
    n_sink : int, default=4
        Initial tokens excluded from the statistics and never pruned.
    use_covariance : bool, default=True
    use_vnorm : bool, default=True
        Rescale by ``(scores + epsilon) * ||V||_2``.
    epsilon : float, default=0.0
    """

    compression_ratio: float = 0.0
    n_future_positions: int = 512
    n_sink: int = 4
    use_covariance: bool = True
    use_vnorm: bool = True
    epsilon: float = 0.0

    def get_query_statistics(self, module: nn.Module, hidden_states: torch.Tensor):
        """Mean and covariance of the pre-RoPE queries (float32), then the averaged RoPE
        (expected_attention_press.py:62-86).  The full-sequence q_proj is a model-owned GEMM."""
        q_len = hidden_states.shape[1]
        h = hidden_states[:, self.n_sink:]
        query_states = get_prerope_query_states(module, h)
        if query_states.shape[2] == 0:
            # no query beyond the sinks (a DecodingPress buffer of <= n_sink hidden states): the reference's mean and
            # covariance of an empty set are NaN (:74-80) and it carries on; so do we (the scores become NaN and the
            # selection falls back to the tie rule) instead of failing in the statistics kernel
            B, Hq, _, D = query_states.shape
            mu = torch.full((B, Hq, D), float("nan"), dtype=torch.float32, device=query_states.device)
            cov = torch.full((B, Hq, D, D), float("nan"), dtype=torch.float32, device=query_states.device) if self.use_covariance else None
            return self.apply_avg_rope(module, mu, cov, q_len)
        mu, cov = _native.ea_qstats(query_states, self.use_covariance)
        return self.apply_avg_rope(module, mu, cov, q_len)

    def _avg_rope_matrix(self, module: nn.Module, q_len: int, device, dtype) -> torch.Tensor:
        """R = mean over positions q_len .. q_len + n_future_positions - 1 of the RoPE matrix (expected_attention_press.py:110-120).
        It depends on the rotary embedding (shared by all layers), q_len and n_future_positions only, so the 32 layers of a
        forward pass share ONE computation (the reference rebuilds it per layer: ~10 small launches each): cached per press
        instance, a handful of 64 KiB matrices at most."""
        rot = module.rotary_emb
        inv = getattr(rot, "inv_freq", None)
        # the table depends on the module's CURRENT frequencies: dynamic / NTK rope re-derives inv_freq in place, users edit the
        # scaling -- so the buffer's identity, its in-place version counter and the attention scaling are part of the key
        key = (id(rot), int(q_len), int(self.n_future_positions), int(module.head_dim), str(device), dtype,
               inv.data_ptr() if inv is not None else 0, inv._version if inv is not None else 0, float(getattr(rot, "attention_scaling", 1.0)),
               str(getattr(rot, "rope_type", "default")))
        cache = self.__dict__.setdefault("_rope_cache", {})
        hit = cache.get(key)
        R = hit[1] if hit is not None and hit[0]() is module.rotary_emb else None   # (the id of a dead module may be reused)
        if R is None:
            position_ids = torch.arange(q_len, q_len + self.n_future_positions, device=device).unsqueeze(0)
            head_dim = module.head_dim
            cos, sin = module.rotary_emb(torch.empty(0, dtype=dtype, device=device), position_ids)
            cos, sin = cos[0].to(dtype), sin[0].to(dtype)
            half = head_dim // 2
            eye_h = torch.eye(half, device=device, dtype=dtype)
            P = torch.zeros((head_dim, head_dim), device=device, dtype=dtype)
            P[half:, :half] = eye_h
            P[:half, half:] = -eye_h
            # mean_p (diag(cos_p) + P * sin_p[:, None]) == diag(mean cos) + P * mean(sin)[:, None]
            R = torch.diag(cos.mean(dim=0)) + sin.mean(dim=0).unsqueeze(1) * P
            if len(cache) >= 8:
                cache.clear()
            cache[key] = (weakref.ref(module.rotary_emb), R)
        return R

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_rope_cache", None)   # weak references to the rotary module: a used press must stay picklable
        return state

    def apply_avg_rope(self, module: nn.Module, mu: torch.Tensor, cov: torch.Tensor, q_len: int):
        """mu <- mu R^T, cov <- R cov R^T with R the RoPE matrix averaged over positions
        q_len .. q_len + n_future_positions - 1 (expected_attention_press.py:88-124).
        D x D host-side math in float32 on the device of mu."""
        R = self._avg_rope_matrix(module, q_len, mu.device, mu.dtype)
        mu = torch.matmul(mu, R.T)
        if cov is not None:
            cov = torch.matmul(R, torch.matmul(cov, R.T))
        return mu, cov

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        assert keys.size(2) > self.n_sink, f"Input should contain more tokens than n_sink={self.n_sink}"
        mean_query, cov_query = self.get_query_statistics(module, hidden_states)
        return _native.ea_score(keys, values, mean_query, cov_query, self.n_sink, self.use_vnorm, self.epsilon)
