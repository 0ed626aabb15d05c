"""FinchPress (kvpress/presses/finch_press.py:16-165): SnapKV-style scores from the QUESTION that follows the context.

The input is ``context + delimiter token + question``; a hook on the embedding layer finds the delimiter, takes the
tokens after it as the observation window and removes the delimiter from the sequence (:124-137).  Scores come from
``kvp_finch_score`` (window attention for any window length, rows weighted by their number of visible keys, no
pooling); the selection is one global ``kvp_topk_select`` or one ``kvp_topk_select_segmented`` over the chunks, the kept
keys are gathered in position order and -- by default -- re-rotated to positions 0..n-1 (``kvp_gather_kv_rerotate``).
The reference's non-rerotating variant keeps torch.topk's order inside the cache; here it is position order, like every
other press of this package (retained order)."""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass, field

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.utils import get_prerope_query_states


@dataclass
class FinchPress(BasePress):
    """FINCH: prompt-guided KV cache compression (https://direct.mit.edu/tacl/article/doi/10.1162/tacl_a_00716/125280).

    Parameters
    ----------
    compression_ratio : float, default=0.0
    chunk_length : int, optional
        Select per chunk of this many tokens instead of over the whole sequence.
    normalize_scores : bool, default=True
        Weight every window row by its number of non-masked attention weights.
    rerotate_keys : bool, default=True
        Re-rotate the kept keys to consecutive positions.
    delimiter_token, delimiter_token_id, window_size : set by ``update_model_and_tokenizer`` / the embedding hook.
    """

    compression_ratio: float = 0.0
    chunk_length: int = None
    normalize_scores: bool = True
    rerotate_keys: bool = True
    delimiter_token: str = field(default=None, init=False)
    delimiter_token_id: int = field(default=None, init=False)
    window_size: int = field(default=None, init=False)

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        """[B, H_kv, S] float32 (finch_press.py:56-83)."""
        W, k_len = self.window_size, keys.shape[2]
        if attentions is not None:
            attn = attentions[..., -W:, :-W]
            if self.normalize_scores:
                attn = attn * torch.arange(k_len - W, k_len, device=attn.device)[None, None, :, None].to(attn.dtype)
            return _native.snapkv_score_from_attn(attn, keys.shape[1], k_len, 1)
        cos, sin = kwargs["position_embeddings"]
        q_pre = get_prerope_query_states(module, hidden_states[:, -W:])
        return _native.finch_score(q_pre, cos[:, -W:], sin[:, -W:], keys, self.normalize_scores)

    def compress(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
                 attentions: torch.Tensor, kwargs: dict) -> tuple[torch.Tensor, torch.Tensor]:
        if self.compression_ratio == 0:
            return keys, values
        assert self.window_size is not None, "window_size must be provided"
        scores = self.score(module, hidden_states, keys, values, attentions, kwargs)
        k_len = keys.shape[2]
        if self.chunk_length is None:
            indices = _native.topk_select(scores, int(k_len * (1 - self.compression_ratio)))
        else:
            assert self.chunk_length > self.window_size / (1 - self.compression_ratio)
            L = self.chunk_length
            n_full, tail = divmod(k_len, L)
            B, H, _ = scores.shape
            parts = []
            if n_full:   # finch_press.py:104-110, all full chunks in one segmented select
                sc = scores[..., : n_full * L].reshape(B * H, n_full * L)
                parts.append(_native.topk_select_segmented(sc, L, max(1, int(L * (1 - self.compression_ratio)))).view(B, H, -1))
            if tail:
                sc = scores[..., n_full * L:].reshape(B * H, tail)
                parts.append(_native.topk_select_segmented(sc, tail, max(1, int(tail * (1 - self.compression_ratio))),
                                                           pos_base=n_full * L).view(B, H, -1))
            indices = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
        if self.rerotate_keys:                                        # ascending positions (:114), gather + re-rotation in one pass
            return _native.gather_kv_rerotate(keys, values, indices, module.rotary_emb.inv_freq)
        return _native.gather_kv(keys, values, indices)

    def embed_token_forward_hook(self, module, input, output):
        """Find the delimiter between context and question, set the window, drop the delimiter (finch_press.py:124-137)."""
        if input[0].shape[1] > 1 and self.delimiter_token_id in input[0][0]:  # prefilling
            assert len(input[0]) == 1, "Only batch size 1 is supported."
            delim_tokens = input[0][0] == self.delimiter_token_id
            assert delim_tokens.sum() == 1, "Only one delimiter token should be present."
            context_length = int(torch.nonzero(delim_tokens)[0].item())
            self.window_size = len(input[0][0]) - 1 - context_length
            assert self.window_size > 0, "No window detected (window size must be > 0)."
            output = output[:, ~delim_tokens]
        return output

    def update_model_and_tokenizer(self, model, tokenizer, delimiter_token: str = "<|finch_sep|>"):
        """Register the delimiter token with the tokenizer and the embedding table (finch_press.py:139-151)."""
        self.delimiter_token = delimiter_token
        if delimiter_token not in tokenizer.get_vocab():
            tokenizer.add_special_tokens({"additional_special_tokens": [delimiter_token]})
        self.delimiter_token_id = tokenizer.convert_tokens_to_ids(delimiter_token)
        model.resize_token_embeddings(len(tokenizer))
        return tokenizer

    @contextmanager
    def __call__(self, model):
        if self.delimiter_token_id is None:
            raise ValueError("No delimiter token ID provided. Use the update_model_and_tokenizer method before calling the press.")
        with super().__call__(model):
            hook = None
            try:
                hook = model.model.embed_tokens.register_forward_hook(self.embed_token_forward_hook)
                yield
            finally:
                if hook is not None:
                    hook.remove()
