"""Head-wise key masking for presses that prune a different token set per head (AdaKVPress).

Same mechanism as the reference's kvpress/attention_patch.py (:8-110): such a press leaves K/V in the cache and stores
``module.masked_key_indices = (batch_idx, head_idx, seq_idx)``; every registered transformers attention function is
wrapped so that, while decoding, the keys at those indices are overwritten with a "fake" key ``k`` whose logit with every
current query is hugely negative (``exp(<q, k>) == 0``): the tokens are as good as removed, at no memory saving.

The reference patches at import time; here ``patch_attention_functions()`` runs (once) when the first press that needs it
is constructed.  This is decode-time host logic (torch ops on one decoding step's queries), not part of the hot path."""
from __future__ import annotations

import torch
from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

_PATCH_FLAG = "_kvpress_amd_head_masking"


def search_hyperplane(X: torch.Tensor, max_iter: int = 1000) -> torch.Tensor:
    """For X [n, m, d] find Y [n, d] with <X[i, j], Y[i]> > 0 for all j, and return ``-1e5 * Y / ||Y||^2`` so that
    ``<x, result> <= -1e5 * <x, Y> / ||Y||^2`` is a hugely negative logit for every row x (attention_patch.py:8-41).
    Perceptron-style: start from the mean row and keep adding the mean of the rows still on the wrong side."""
    Y = X.mean(dim=1)
    for _ in range(max_iter):
        wrong = torch.bmm(X, Y.unsqueeze(-1)) <= 0          # [n, m, 1]
        if not wrong.any():
            return -1e5 * Y / Y.norm(dim=-1, keepdim=True) ** 2
        Y = Y + (X * wrong).sum(dim=1) / wrong.sum(dim=1).clamp(min=1)
    raise ValueError("Could not find fake keys such that for every query q, exp(<q, k>) = 0")


def attention_patch(func):
    """Wrap one attention function (attention_patch.py:44-86)."""
    if getattr(func, _PATCH_FLAG, False):
        return func

    def wrapper(module, query, key, value, attention_mask, dropout, **kwargs):
        if query.shape[2] == key.shape[2]:
            module.masked_key_indices = None                 # prefill: nothing is masked yet
        elif getattr(module, "masked_key_indices", None) is not None:
            bsz, num_heads, q_len, head_dim = query.shape
            num_kv = key.shape[1]
            groups = num_heads // num_kv
            # one fake key per (batch, kv head): negative logit with every query of its group in this step
            q = query.view(bsz, num_kv, groups, q_len, head_dim).reshape(bsz * num_kv, groups * q_len, head_dim)
            fake = search_hyperplane(q).view(bsz, num_kv, head_dim)
            b_idx, h_idx, s_idx = module.masked_key_indices
            key[b_idx, h_idx, s_idx] = fake[b_idx, h_idx]
        if "cu_seq_lens_k" in kwargs:                        # only with model.generate (kvpress PR 115)
            kwargs["cu_seq_lens_k"][-1] = key.shape[-2]
        return func(module, query, key, value, attention_mask, dropout, **kwargs)

    setattr(wrapper, _PATCH_FLAG, True)
    return wrapper


def patch_attention_functions():
    """Wrap every attention function transformers has registered (idempotent)."""
    for name, func in list(ALL_ATTENTION_FUNCTIONS.items()):
        ALL_ATTENTION_FUNCTIONS[name] = attention_patch(func)
