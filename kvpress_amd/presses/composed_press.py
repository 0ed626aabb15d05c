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
        # composed_press.py:31-34 forbids AdaKVPress / KVzipPress here.  The same holds for every press that records
        # ``module.masked_key_indices`` relative to the cache it saw: a later pruning press would silently invalidate them.
        from kvpress_amd.presses.adakv_press import AdaKVPress
        from kvpress_amd.presses.criticalkv_press import CriticalAdaKVPress
        from kvpress_amd.presses.dms_press import DMSPress
        from kvpress_amd.presses.duo_attention_press import DuoAttentionPress

        masking = (AdaKVPress, CriticalAdaKVPress, DMSPress, DuoAttentionPress)
        assert not any(isinstance(press, masking) for press in self.presses), \
            "ComposedPress cannot contain presses that mask keys through module.masked_key_indices (AdaKVPress, CriticalAdaKVPress, DMSPress, DuoAttentionPress)"

    def post_init_from_model(self, model):
        for press in self.presses:
            press.post_init_from_model(model)

    def forward_hook(self, module, input, kwargs, output):
        retained = 1.0
        for press in self.presses:
            output = press.forward_hook(module, input, kwargs, output)
            retained *= 1 - press.compression_ratio
        self.compression_ratio = 1 - retained
        return output
