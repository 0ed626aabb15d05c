"""CriticalKVPress / CriticalAdaKVPress (kvpress/presses/criticalkv_press.py:20-194): two-stage selection; the second
stage rescales the scores with the L1 norm of every value vector after the output projection, ||Wo_h v||_1.

Per q-head the projection V_h Wo_h is one plain library GEMM on the model's own ``o_proj`` weight (torch / hipBLASLt, the
same standing as the model's q_proj), its row-wise L1 norm is ``kvp_rowl1_score``; stage budgets use ``kvp_topk_select``
and ``kvp_scores_fill_at``; the cross-head bottom-k of the Ada variant is ``kvp_topk_select | KVP_TOPK_SMALLEST``."""
from __future__ import annotations

import logging
from dataclasses import dataclass

import torch
from torch import nn

from kvpress_amd import _native
from kvpress_amd.attention_patch import patch_attention_functions
from kvpress_amd.presses.base_press import BasePress
from kvpress_amd.presses.expected_attention_press import ExpectedAttentionPress
from kvpress_amd.presses.scorer_press import ScorerPress

logger = logging.getLogger(__name__)
_FMAX = torch.finfo(torch.float32).max   # scores are float32 here (the reference uses the max of the model dtype)


def vwl1norm(values: torch.Tensor, module: nn.Module) -> torch.Tensor:
    """[B, H_kv, S] float32: mean over the kv-head's q-heads of ||Wo_hq^T v||_1 (criticalkv_press.py:57-77)."""
    B, H, S, D = values.shape
    Hq = module.config.num_attention_heads
    G = Hq // H
    Wo = module.o_proj.weight.transpose(0, 1).reshape(Hq, D, -1)              # [Hq, D, hidden]
    out = torch.zeros((B, H, S), dtype=torch.float32, device=values.device)
    for hq in range(Hq):                                                       # head-wise, like the reference: one [B,S,hidden] temp at a time
        proj = torch.matmul(values[:, hq // G], Wo[hq].to(values.dtype))       # library GEMM, model dtype
        out[:, hq // G] += _native.rowl1_score(proj, 1.0 / G)
    return out


class CriticalKVPress(ScorerPress):
    """CriticalKV (https://arxiv.org/abs/2502.03805) on any ScorerPress.

    Parameters
    ----------
    press : ScorerPress
    epsilon : float, default=1e-4
        Added to the scores before the rescaling.
    first_stage_ratio : float, default=0.5
        Share of the budget that the wrapped press's own ranking fills.
    """

    def __init__(self, press: ScorerPress, epsilon: float = 1e-4, first_stage_ratio: float = 0.5):
        self.press = press
        self.epsilon = epsilon
        self.first_stage_ratio = first_stage_ratio
        assert isinstance(self.press, ScorerPress), "CriticalKVPress requires a ScorerPress as input"
        if isinstance(self.press, ExpectedAttentionPress) and self.press.use_vnorm:
            logger.warning("use_vnorm should be disabled for CriticalKVPress")

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    @property
    def compression_ratio(self):
        return self.press.compression_ratio

    @compression_ratio.setter
    def compression_ratio(self, value):
        self.press.compression_ratio = value

    vwl1norm = staticmethod(vwl1norm)

    def score(self, module, hidden_states, keys, values, attentions, kwargs):
        scores = self.press.score(module, hidden_states, keys, values, attentions, kwargs).to(torch.float32)   # stage 1 (:81)
        k_len = keys.shape[2]
        budget = int((1 - self.compression_ratio) * k_len * self.first_stage_ratio)                           # (:83)
        first = _native.topk_select(scores, budget)
        scores = (scores + self.epsilon) * vwl1norm(values, module)                                            # stage 2 (:87-88)
        return _native.scores_fill_at_(scores.contiguous(), first, _FMAX)                                      # merge (:91)


@dataclass
class CriticalAdaKVPress(BasePress):
    """CriticalAdaKV: the two-stage selection inside AdaKV's head-wise budgets (criticalkv_press.py:96-194).

    Parameters
    ----------
    press : ScorerPress
    alpha_safeguard : float, default=0.20
    epsilon : float, default=1e-4
    first_stage_ratio : float, default=0.5
    """

    press: ScorerPress = None
    alpha_safeguard: float = 0.20
    epsilon: float = 1e-4
    first_stage_ratio: float = 0.5

    def __post_init__(self):
        assert 0 <= self.alpha_safeguard <= 1, "alpha_safeguard should be in 0, 1]"
        assert isinstance(self.press, ScorerPress), "CriticalAdaKVPress requires a ScorerPress as input"
        if isinstance(self.press, ExpectedAttentionPress) and self.press.use_vnorm:
            logger.warning("use_vnorm should be disabled for CriticalAdaKVPress")
        patch_attention_functions()

    def post_init_from_model(self, model):
        self.press.post_init_from_model(model)

    @property
    def compression_ratio(self):
        return self.press.compression_ratio

    @compression_ratio.setter
    def compression_ratio(self, value):
        self.press.compression_ratio = value

    @staticmethod
    def _protect_best(scores: torch.Tensor, counts: list) -> None:
        """scores[b, h, j] = max for the counts[h] best j of every (b, h), in place: one descending-score select of the largest
        count, entries beyond a head's own count turned into -1 (which kvp_scores_fill_at skips), one fill."""
        n = max(counts)
        if n <= 0:
            return
        order = _native.topk_select(scores, n, _native.ORDER_SCORE)                                           # [B, H, n]
        beyond = torch.arange(n, device=scores.device)[None, None, :] >= torch.tensor(counts, device=scores.device)[None, :, None]
        _native.scores_fill_at_(scores, order.masked_fill(beyond, -1), _FMAX)

    def compress(self, module, hidden_states, keys, values, attentions, kwargs):
        if self.compression_ratio == 0:
            return keys, values
        assert module.config._attn_implementation != "eager", "eager mode not supported"
        N = _native
        scores = self.press.score(module, hidden_states, keys, values, attentions, kwargs).to(torch.float32).contiguous()
        bsz, H, k_len = scores.shape
        n_kept = int(k_len * (1 - self.compression_ratio))
        n_safe = int(n_kept * self.alpha_safeguard)
        N.scores_fill_at_(scores, N.topk_select(scores, n_safe), _FMAX)                                        # safeguard (:141-142)

        # head budgets: how many of the n_kept * H best (token, head) pairs of the layer fall to each head (:149-154)
        flat_top = N.topk_select(scores.view(bsz, H * k_len), n_kept * H)
        head_budgets = torch.bincount((flat_top // k_len).flatten().long(), minlength=H)                       # summed over the batch, as the reference
        stage1 = (head_budgets * self.first_stage_ratio).to(torch.int64).tolist()
        budgets = head_budgets.tolist()

        # stage 1: each head's own `stage1[h]` best tokens are protected (:157-160)
        self._protect_best(scores, stage1)
        # stage 2: rescale, then each head's `budgets[h]` best of the rescaled scores (:163-168)
        scores = ((scores + self.epsilon) * vwl1norm(values, module)).contiguous()
        self._protect_best(scores, budgets)

        # bottom-k across heads -> masked during attention (:174-183)
        n_pruned = H * (k_len - n_kept)
        idx = N.topk_select(scores.view(bsz, H * k_len), n_pruned, N.ORDER_POSITION | N.TOPK_SMALLEST).flatten().long()
        batch_indices = torch.arange(bsz, device=idx.device).repeat_interleave(n_pruned)
        module.masked_key_indices = (batch_indices, idx // k_len, idx % k_len)
        return keys, values
