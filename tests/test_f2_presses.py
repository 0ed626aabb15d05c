"""Host logic of the SURVEY §8 f-2 presses (PyramidKV, TOVA, KeyDiff, StreamingLLM, Random) on CPU: the classes run
through the oracle-backed fakes of conftest.py and are compared with the REAL reference's outputs (tests/golden)."""
import os

import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
F2 = [n for n, c in _inputs.CASES.items() if c["kind"] in ("pyramid", "tova", "keydiff", "streaming", "cur", "qfilter", "observed", "lagkv")]


def make_press(s, ratio):
    import kvpress_amd as P

    k = s["kind"]
    if k == "pyramid":
        return P.PyramidKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"], beta=s["beta"])
    if k == "tova":
        return P.TOVAPress(compression_ratio=ratio)
    if k == "keydiff":
        return P.KeyDiffPress(compression_ratio=ratio)
    if k == "lagkv":
        return P.contrib.LagKVPress(compression_ratio=ratio, n_sink=s["n_sink"], lag_size=s["lag"], cross_scoring=s.get("cross", False))
    if k == "observed":
        return P.contrib.ObservedAttentionPress(compression_ratio=ratio)
    if k == "qfilter":
        p = P.QFilterPress(compression_ratio=ratio)
        p.q_filters = torch.from_numpy(_inputs.make_qfilters(s))
        return p
    if k == "cur":
        return P.CURPress(compression_ratio=ratio, num_sinks=s.get("sinks", 4), leverage_type=s["leverage"],
                          use_local_approximation=s.get("local", True), local_window_size=s.get("window", 16))
    return P.StreamingLLMPress(compression_ratio=ratio, n_sink=s["n_sink"])


@pytest.mark.parametrize("name", F2)
def test_press_matches_reference_cpu(name, fake_native):
    s = _inputs.make_case(name)
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
    if s["kind"] == "qfilter":
        att.layer_idx = _inputs.QF_LAYER
    keys, values = torch.from_numpy(s["keys"]), torch.from_numpy(s["values"])
    kwargs = {"position_embeddings": pe}
    attn = torch.from_numpy(_inputs.make_attentions(s)) if s["kind"] == "observed" else None
    with torch.no_grad():
        sc = make_press(s, 0.5).score(att, hidden, keys, values, attn, kwargs).numpy()
        ref = g["scores_f32"]
        if s["kind"] == "lagkv":
            _inputs.assert_lag_scores_close(sc, ref, s, name)
        elif s["kind"] == "observed":
            np.testing.assert_allclose(sc, ref, rtol=2e-5, atol=1e-30)
        elif s["kind"] == "qfilter":   # a signed dot product crossing zero: absolute tolerance
            np.testing.assert_allclose(sc, ref, rtol=2e-4, atol=2e-5)
        elif s["kind"] in ("pyramid", "tova"):
            W = s["W"]
            np.testing.assert_allclose(sc[..., :-W], ref[..., :-W], rtol=2e-4)
            assert (sc[..., -W:] > sc[..., :-W].max()).all()
        elif s["kind"] == "keydiff":
            np.testing.assert_allclose(sc, ref, rtol=0, atol=2e-6)
        elif s["kind"] == "cur":
            np.testing.assert_allclose(sc, ref, rtol=2e-4, atol=1e-30)
        else:
            assert np.array_equal(sc, ref)
        for i, r in enumerate(s["ratios"]):
            ko, vo = make_press(s, r).compress(att, hidden, keys, values, attn, kwargs)
            n = int(g[f"nkept_{i}"])
            assert tuple(ko.shape) == tuple(vo.shape) == (s["B"], s["H"], n, s["D"])
            if s["kind"] == "streaming":  # sinks + most recent tokens, exactly the reference's set
                idx = g[f"idx_f32_{i}"]
                wk, wv = O.gather_kv(s["keys"], s["values"], idx)
                assert np.array_equal(ko.numpy(), wk) and np.array_equal(vo.numpy(), wv)
                n_pruned = s["S"] - n
                assert idx[0, 0].tolist() == list(range(s["n_sink"])) + list(range(s["n_sink"] + n_pruned, s["S"]))
        k0, v0 = make_press(s, 0.0).compress(att, hidden, keys, values, attn, kwargs)
        assert k0 is keys and v0 is values


class _Cfg:
    def __init__(self, n):
        self.num_hidden_layers = n


class _Mod(torch.nn.Module):
    def __init__(self, n, i):
        super().__init__()
        self.config, self.layer_idx = _Cfg(n), i


# the reference's own test of the budget (tests/presses/test_pyramidkv_press.py:27-52), same grid
@pytest.mark.parametrize("num_hidden_layers", [32, 64, 128])
@pytest.mark.parametrize("compression_ratio", [0.1, 0.25, 0.3, 0.5, 0.6, 0.75, 0.8])
@pytest.mark.parametrize("q_len", [1024, 2787, 4096, 6591, 8192])
def test_pyramid_mean_layer_budget(num_hidden_layers, compression_ratio, q_len):
    import kvpress_amd as P

    press = P.PyramidKVPress()
    press.compression_ratio = compression_ratio
    budgets = [press.get_layer_budget(_Mod(num_hidden_layers, i), q_len) for i in range(num_hidden_layers)]
    assert sum(budgets) / num_hidden_layers == pytest.approx(q_len * (1 - compression_ratio), rel=1e-3)
    assert budgets == sorted(budgets, reverse=True)
    # and the class agrees with the oracle restatement that the golden fixtures pin (py_* cases)
    assert budgets == [O.pyramidkv_budget(q_len, compression_ratio, 64, 20, num_hidden_layers, i) for i in range(num_hidden_layers)]


def test_pyramid_falls_back_to_snapkv_budget():
    import kvpress_amd as P

    # short prompt: the ramp would drop below the window -> round(q_len * (1 - r)) (pyramidkv_press.py:76-78)
    press = P.PyramidKVPress(compression_ratio=0.9, window_size=64, beta=20)
    assert press.get_layer_budget(_Mod(8, 3), 200) == round(200 * 0.1)


def test_streaming_asserts_on_short_input(fake_native):
    import kvpress_amd as P

    k = torch.zeros(1, 1, 4, 8)
    with pytest.raises(AssertionError):
        P.StreamingLLMPress(0.5, n_sink=4).score(None, None, k, k, None, {})


def test_tiny_llama_end_to_end_lengths(fake_native):
    """The new presses under the forward hook of a real (tiny) model: per-layer cache lengths."""
    import kvpress_amd as P
    from transformers import DynamicCache

    model = _inputs.make_tiny_llama()
    ids = torch.randint(3, 59, (1, 120))
    for press, want in ((P.TOVAPress(0.5), [60, 60]), (P.KeyDiffPress(0.25), [90, 90]), (P.StreamingLLMPress(0.5), [60, 60]),
                        (P.RandomPress(0.5, seed=1), [60, 60]),
                        (P.PyramidKVPress(0.5, window_size=8), None)):
        cache = DynamicCache()
        with torch.no_grad(), press(model):
            model.model(input_ids=ids, past_key_values=cache)
        got = [cache.get_seq_length(i) for i in range(2)]
        if want is None:
            want = [press.get_layer_budget(model.model.layers[i].self_attn, 120) for i in range(2)]
            assert want[0] > want[1] and sum(want) == 120
        assert got == want
