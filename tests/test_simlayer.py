"""SimLayerKVPress: the laziness statistic against the oracle restatement, and the press's bookkeeping
(kvpress/presses/simlayerkv_press.py:41-116).  The end-to-end behaviour is pinned by the pipeline goldens
(pipe_simlayer_lazy / pipe_simlayer_busy in tests/test_pipeline.py)."""
import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O


def _check(dev, dt, name="sk_257_A"):
    import kvpress_amd as P

    s = _inputs.make_case(name)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, dev)
    keys = torch.from_numpy(s["keys"]).to(device=dev, dtype=dt)
    values = torch.from_numpy(s["values"]).to(device=dev, dtype=dt)
    kwargs = {"position_embeddings": pe}
    for n_last, n_recent, n_initial in ((1, 32, 4), (3, 100, 0), (2, 8, 16)):
        press = P.SimLayerKVPress(lazy_threshold=0.5, n_last=n_last, n_recent=n_recent, n_initial=n_initial)
        with torch.no_grad():
            got = float(press.lazy_score(att, hidden, keys, pe))
            q = P.SnapKVPress.compute_window_queries(att, hidden, n_last, pe)
        want = O.simlayer_lazy_score(q.float().cpu().numpy(), s["keys"], n_initial, n_recent)
        assert abs(got - want) <= 1e-3 * abs(want), (n_last, n_recent, n_initial, got, want)
        # a threshold on either side of the statistic decides the layer
        for thr, lazy in ((want * 0.9, True), (min(1.0, want * 1.1 + 1e-6), False)):
            if thr >= 1.0:
                continue
            p = P.SimLayerKVPress(lazy_threshold=thr, n_last=n_last, n_recent=n_recent, n_initial=n_initial)
            att.layer_idx = 0
            with torch.no_grad():
                ko, vo = p.compress(att, hidden, keys, values, None, kwargs)
            S = s["S"]
            if lazy:
                n = n_initial + n_recent - n_last
                assert tuple(ko.shape) == (s["B"], s["H"], n, s["D"]) and ko.is_contiguous()
                ref = torch.cat([keys[:, :, :n_initial], keys[:, :, S - n_recent + n_last:]], dim=2)
                assert torch.equal(ko, ref) and torch.equal(vo, torch.cat([values[:, :, :n_initial], values[:, :, S - n_recent + n_last:]], dim=2))
                assert p.compression_ratio == pytest.approx((S - n_initial - n_recent + 1) / S)
            else:
                assert ko is keys and vo is values and p.compression_ratio == 0.0


def test_lazy_score_and_compress_cpu(fake_native):
    _check("cpu", torch.float32)


def test_bookkeeping(fake_native):
    import kvpress_amd as P

    p = P.SimLayerKVPress()
    with pytest.raises(ValueError):
        p.compression_ratio
    with pytest.raises(AttributeError):
        p.compression_ratio = 0.5
    with pytest.raises(AssertionError):
        P.SimLayerKVPress(lazy_threshold=1.5)
    # threshold 1.0 or a sequence no longer than n_initial + n_recent + n_last: untouched
    k = torch.zeros(1, 2, 30, 8)

    class M:
        layer_idx = 0
    assert p.compress(M(), None, k, k, None, {})[0] is k and p.compression_ratio == 0.0
    q = P.SimLayerKVPress(lazy_threshold=0.1, n_recent=40)
    assert q.compress(M(), None, k, k, None, {})[0] is k


@pytest.mark.gpu
def test_lazy_score_and_compress_gpu():
    _check("cuda:0", torch.float32)
    _check("cuda:0", torch.bfloat16)
