"""ComposedPress (kvpress/presses/composed_press.py:12-62): several presses applied one after the other."""
from __future__ import annotations

from dataclasses import dataclass

from kvpress_amd.presses.base_press import BasePress


@dataclass
class ComposedPress(BasePress):
    """Chain compression methods: each press's forward hook runs on the cache the previous one left; the overall
    ``compression_ratio`` (1 - product of the retained fractions) is updated after every hook call.

    Parameters
    ----------
    presses : list[BasePress]
    """

    presses: list[BasePress]

    def __post_init__(self):
        self.compression_ratio = None
        from kvpress_amd.presses.adakv_press import AdaKVPress

        # the reference's own rule (composed_press.py:47-50; KVzipPress is not part of this package)
        assert not any(isinstance(press, AdaKVPress) for press in self.presses), "ComposedPress cannot contains AdaKVPress or KVzipPress"
        self._order_checked = False

    def _check_kept_order(self):
        # once per ComposedPress, at its first hook call (ratios / kept_order may be set after construction): a ScorerPress in front
        # of an order-dependent press hands over its survivors in the reference's order (scorer_press.resolve_chain_kept_order)
        from kvpress_amd.presses.scorer_press import resolve_chain_kept_order

        self._order_checked = True
        for i, press in enumerate(self.presses[:-1]):
            resolve_chain_kept_order(press, self.presses[i + 1:], "ComposedPress")

    def post_init_from_model(self, model):
        for press in self.presses:
            press.post_init_from_model(model)

    def forward_hook(self, module, input, kwargs, output):
        if not getattr(self, "_order_checked", False):
            self._check_kept_order()
        retained = 1.0
        for press in self.presses:
            output = press.forward_hook(module, input, kwargs, output)
            retained *= 1 - press.compression_ratio
        self.compression_ratio = 1 - retained
        return output
