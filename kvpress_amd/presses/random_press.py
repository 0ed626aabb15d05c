"""RandomPress (kvpress/presses/random_press.py:15-46): uniformly random scores (baseline)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from kvpress_amd.presses.scorer_press import ScorerPress


@dataclass
class RandomPress(ScorerPress):
    """Prunes a random subset.  The scores come from torch's generator (float32, on the keys' device; seeded when
    ``seed`` is given); selection and gather are the HIP kernels.

    Parameters
    ----------
    compression_ratio : float, default=0.0
    seed : int, optional
    """

    compression_ratio: float = 0.0
    seed: Optional[int] = None

    def score(self, module: nn.Module, hidden_states: torch.Tensor, keys: torch.Tensor, values: torch.Tensor,
              attentions: torch.Tensor, kwargs) -> torch.Tensor:
        generator = None
        if self.seed is not None:
            generator = torch.Generator(device=keys.device)
            generator.manual_seed(self.seed)
        return torch.rand(*keys.shape[:-1], generator=generator, device=keys.device, dtype=torch.float32)
