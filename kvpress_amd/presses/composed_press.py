"""ComposedPress (kvpress/presses/composed_press.py:12-62): several presses applied one after the other."""
from __future__ import annotations

from dataclasses import dataclass

from kvpress_amd.presses.base_press import BasePress


def _inner_presses(press):
    out = []
    for name in ("press", "prefill_press", "decoding_press", "base_press"):
        p = getattr(press, name, None)
        if isinstance(p, BasePress):
            out.append(p)
    out += [p for p in getattr(press, "presses", None) or [] if isinstance(p, BasePress)]
    return out


def _masks_keys(press) -> bool:
    """press (or a press it wraps) records module.masked_key_indices"""
    from kvpress_amd.presses.adakv_press import AdaKVPress
    from kvpress_amd.presses.criticalkv_press import CriticalAdaKVPress
    from kvpress_amd.presses.dms_press import DMSPress
    from kvpress_amd.presses.duo_attention_press import DuoAttentionPress

    return isinstance(press, (AdaKVPress, CriticalAdaKVPress, DMSPress, DuoAttentionPress)) or any(_masks_keys(p) for p in _inner_presses(press))


def _masking_only_types():
    from kvpress_amd.presses.criticalkv_press import CriticalAdaKVPress
    from kvpress_amd.presses.dms_press import DMSPress
    from kvpress_amd.presses.duo_attention_press import DuoAttentionPress

    return (CriticalAdaKVPress, DMSPress, DuoAttentionPress)


def _prunes_positions(press) -> bool:
    """press removes or reorders cache positions.  Classified by the OUTERMOST type: the channel-pruning ThinKPress and the
    masking-only presses (CriticalAdaKVPress, DMSPress, DuoAttentionPress -- whatever scorer they wrap, they only record
    module.masked_key_indices) leave the positions alone."""
    from kvpress_amd.presses.think_press import ThinKPress

    return not isinstance(press, (ThinKPress,) + _masking_only_types())


def _overwrites_mask(press) -> bool:
    """press ASSIGNS module.masked_key_indices (an earlier press's mask is lost); DMSPress merges into the existing indices."""
    from kvpress_amd.presses.dms_press import DMSPress

    return isinstance(press, _masking_only_types()) and not isinstance(press, DMSPress)


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
        # Beyond the reference: presses that record ``module.masked_key_indices`` relative to the cache they saw (CriticalAdaKVPress,
        # DMSPress, DuoAttentionPress, also nested inside a wrapper) are fine on their own, in last position, or followed only by a
        # channel-pruning ThinKPress -- the reference's test suite composes a lone DuoAttentionPress -- but a later press that
        # removes or reorders positions silently invalidates those indices: warn about exactly that case.
        for i, press in enumerate(self.presses):
            if _masks_keys(press) and any(_prunes_positions(later) for later in self.presses[i + 1:]):
                import warnings

                warnings.warn(f"ComposedPress: {type(press).__name__} masks keys through module.masked_key_indices, but a later press "
                              f"prunes positions: the masked indices will no longer refer to the same tokens", stacklevel=2)
            if _masks_keys(press) and any(_overwrites_mask(later) for later in self.presses[i + 1:]):
                import warnings

                warnings.warn(f"ComposedPress: {type(press).__name__} records module.masked_key_indices, but a later masking press assigns "
                              f"them anew: the earlier mask is discarded", stacklevel=2)

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
