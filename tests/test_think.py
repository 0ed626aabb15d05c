"""ThinKPress (key-channel pruning) against the REAL reference's outputs (tests/golden/think_*.npz, oracle/gen_golden_think.py).
CPU: the oracle restatement and the press's host logic over oracle-backed entry points; GPU (marked): the kernels."""
import os

import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = list(_inputs.THINK_CASES)
DEV = "cuda:0"


def gold(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    s = _inputs.make_think_case(name)
    g = gold(name)
    sc = O.think_channel_scores(g["qwin_f32"], s["keys"])
    for i, r in enumerate(s["ratios"]):
        idx, k = O.think_prune(s["keys"], sc, r)
        assert np.array_equal(idx, g[f"pruned_f32_{i}"]), f"{name} r={r}"
        assert (np.take_along_axis(k, np.broadcast_to(idx[:, :, None, :].astype(np.int64), k.shape[:3] + (idx.shape[-1],)), -1) == 0).all()


def _run(s, dev, dt):
    import kvpress_amd as P

    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, dev)
    values = torch.from_numpy(s["values"]).to(device=dev, dtype=dt)
    kwargs = {"position_embeddings": pe, "hidden_states": hidden}
    out = []
    with torch.no_grad():
        for i, r in enumerate(s["ratios"]):
            press = P.contrib.ThinKPress(key_channel_compression_ratio=r, window_size=s["W"])
            assert press.compression_ratio == r / 2
            keys = torch.from_numpy(s["keys"]).to(device=dev, dtype=dt).clone()
            ko, vo = press.compress(att, hidden, keys, values, None, kwargs)
            assert ko is keys and vo is values                              # in place, like the reference (think_press.py:82)
            zero = (ko == 0).all(dim=2)
            n = int(s["D"] * r)
            assert (zero.sum(-1) == n).all()
            pruned = torch.nonzero(zero)[:, 2].view(s["B"], s["H"], n).cpu().numpy()
            # every other channel is untouched
            keep = ~zero[:, :, None, :].expand_as(ko)
            assert torch.equal(ko[keep], torch.from_numpy(s["keys"]).to(device=dev, dtype=dt)[keep])
            out.append((i, r, pruned))
        k0 = torch.from_numpy(s["keys"]).to(device=dev, dtype=dt)
        a, b = P.contrib.ThinKPress(0.0).compress(att, hidden, k0, values, None, kwargs)
        assert a is k0 and b is values
        with pytest.raises(AttributeError):
            P.contrib.ThinKPress(0.5).compression_ratio = 0.1
    return out


@pytest.mark.parametrize("name", NAMES)
def test_press_matches_reference_cpu(name, fake_native):
    s = _inputs.make_think_case(name)
    g = gold(name)
    for i, r, pruned in _run(s, "cpu", torch.float32):
        assert np.array_equal(pruned, g[f"pruned_f32_{i}"]), f"{name} r={r}"


def test_composed_with_token_press(fake_native):
    """The reference's test_composed_press (tests/presses/test_presses.py:33-39): Knorm then ThinK on the tiny model."""
    from transformers import DynamicCache

    import kvpress_amd as P

    model = _inputs.make_tiny_llama()
    ids = torch.randint(3, 59, (1, 64), generator=torch.Generator().manual_seed(0))
    cache = DynamicCache()
    with torch.no_grad(), P.ComposedPress([P.KnormPress(compression_ratio=0.5), P.contrib.ThinKPress(key_channel_compression_ratio=0.5, window_size=2)])(model):
        model(ids, past_key_values=cache)
    assert cache.get_seq_length() == 32
    for layer in cache.layers:
        assert ((layer.keys == 0).all(dim=2).sum(-1) == 3).all()            # int(6 * 0.5) channels of every head are zero


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_think_kernels_vs_oracle(name):
    from kvpress_amd import _native

    s = _inputs.make_think_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    q = _inputs.round_to(gold(name)["qwin_f32"], s["dtype"])
    k = torch.from_numpy(s["keys"]).to(device=DEV, dtype=dt)
    got = _native.think_channel_scores(torch.from_numpy(q).to(device=DEV, dtype=dt), k).cpu().numpy()
    np.testing.assert_allclose(got, O.think_channel_scores(q, s["keys"]), rtol=2e-5, atol=1e-30, err_msg=name)
    # strided key view (every second token)
    got2 = _native.think_channel_scores(torch.from_numpy(q).to(device=DEV, dtype=dt), k[:, :, ::2]).cpu().numpy()
    np.testing.assert_allclose(got2, O.think_channel_scores(q, s["keys"][:, :, ::2]), rtol=2e-5, atol=1e-30, err_msg=name)
    idx = _native.topk_select(torch.from_numpy(got).to(DEV), s["D"] // 3, _native.ORDER_POSITION | _native.TOPK_SMALLEST)
    kk = k.clone()
    _native.zero_channels_(kk, idx)
    _, want = O.think_prune(s["keys"], got, 0.0)
    want = np.array(s["keys"], copy=True)
    np.put_along_axis(want, np.broadcast_to(idx.cpu().numpy()[:, :, None, :].astype(np.int64), want.shape[:3] + (idx.shape[-1],)), 0, axis=-1)
    assert np.array_equal(kk.float().cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_press_matches_reference_gpu(name):
    s = _inputs.make_think_case(name)
    g = gold(name)
    for i, r, pruned in _run(s, DEV, torch.float32):
        assert np.array_equal(pruned, g[f"pruned_f32_{i}"]), f"{name} r={r}"
    if s["dtype"] != "f32":   # model dtype: the same channels up to near-ties of the 16-bit reference scores
        for i, r, pruned in _run(s, DEV, _inputs.torch_dtype(s["dtype"])):
            ref = g[f"pruned_nat_{i}"]
            same = np.mean([len(np.intersect1d(a, b)) / max(1, a.size) for a, b in zip(pruned.reshape(-1, pruned.shape[-1]), ref.reshape(-1, ref.shape[-1]))])
            assert same >= 0.9, f"{name} r={r}: overlap {same:.3f}"
