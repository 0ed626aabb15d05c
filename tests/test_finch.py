"""FinchPress against the REAL reference's outputs (tests/golden/finch_*.npz, oracle/gen_golden_finch.py).
CPU: the oracle restatement and the press's host logic over oracle-backed entry points; GPU (marked): kvp_finch_score
and the press on the HIP kernels."""
import os

import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = list(_inputs.FINCH_CASES)
DEV = "cuda:0"
RTOL = 1e-3  # BASELINE.json north_star: float scores within 1e-3 relative


def gold(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def _close(got, want, W, rtol):
    np.testing.assert_allclose(got[..., :-W], want[..., :-W], rtol=rtol, atol=1e-30)
    assert (got[..., -W:] > got[..., :-W].max()).all()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    s = _inputs.make_finch_case(name)
    g = gold(name)
    sc = O.finch_score(g["qwin_f32"], s["keys"], s["normalize"])
    _close(sc, g["scores_f32"], s["W"], 2e-4)
    for i, r in enumerate(s["ratios"]):
        assert np.array_equal(O.finch_indices(sc, r, s["chunk_length"]), g[f"pos_f32_{i}"]), f"{name} r={r}"


def _run_press(s, name, dev, dt):
    import kvpress_amd as P

    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, dev)
    keys = torch.from_numpy(s["keys"]).to(device=dev, dtype=dt)
    # a VALUE tensor that stores each token's position: channel 0 = pos // 256, channel 1 = pos % 256 (exact in bf16 / f16)
    ar = torch.arange(s["S"], device=dev)
    posv = torch.zeros((s["B"], s["H"], s["S"], s["D"]), dtype=dt, device=dev)
    posv[..., 0], posv[..., 1] = (ar // 256).to(dt), (ar % 256).to(dt)
    decode = lambda v: (v[..., 0].float() * 256 + v[..., 1].float()).round().to(torch.int64).cpu().numpy()
    kwargs = {"position_embeddings": pe}
    out = []
    with torch.no_grad():
        sc = _inputs.make_finch_press(P, s, 0.5).score(att, hidden, keys, posv, None, kwargs)
        assert sc.dtype == torch.float32 and tuple(sc.shape) == (s["B"], s["H"], s["S"])
        for i, r in enumerate(s["ratios"]):
            ko, vo = _inputs.make_finch_press(P, s, r).compress(att, hidden, keys, posv, None, kwargs)
            assert ko.is_contiguous() and ko.dtype == dt
            out.append((i, r, ko.float().cpu().numpy(), decode(vo)))
        k0, v0 = _inputs.make_finch_press(P, s, 0.0).compress(att, hidden, keys, posv, None, kwargs)
        assert k0 is keys and v0 is posv
    return sc.cpu().numpy(), out


def _check_fp32(s, name, dev, rtol):
    g = gold(name)
    sc, out = _run_press(s, name, dev, torch.float32)
    _close(sc, g["scores_f32"], s["W"], rtol)
    for i, r, ko, pos in out:
        assert np.array_equal(pos, g[f"pos_f32_{i}"]), f"{name} r={r}: kept positions"   # ours come out sorted
        if s["rerotate"]:
            np.testing.assert_allclose(ko, g[f"kout_f32_{i}"], rtol=1e-5, atol=4e-6)
        else:
            wk, _ = O.gather_kv(s["keys"], s["values"], pos)
            assert np.array_equal(ko, wk)


@pytest.mark.parametrize("name", NAMES)
def test_press_matches_reference_cpu(name, fake_native):
    _check_fp32(_inputs.make_finch_case(name), name, "cpu", 2e-4)


def test_window_from_delimiter_and_hook_lifecycle(fake_native):
    """The embedding hook (finch_press.py:124-137) and update_model_and_tokenizer (:139-151) on the tiny Llama."""
    from transformers import DynamicCache

    import kvpress_amd as P

    model, tok = _inputs.make_tiny_llama(), _inputs.make_tiny_tokenizer()
    press = P.FinchPress(compression_ratio=0.5)
    with pytest.raises(ValueError):
        with press(model):
            pass
    tok = press.update_model_and_tokenizer(model, tok)
    assert press.delimiter_token == "<|finch_sep|>" and press.delimiter_token_id == tok.convert_tokens_to_ids("<|finch_sep|>")
    assert model.get_input_embeddings().weight.shape[0] == len(tok)   # resize_token_embeddings(len(tokenizer)), :150
    ids = tok("<s>" + _inputs.tiny_context(60) + "<|finch_sep|>" + "w1 w2 w3 w4 w5", return_tensors="pt", add_special_tokens=False).input_ids
    n_ctx, n_q = 61, 5
    assert ids.shape[1] == n_ctx + 1 + n_q and int((ids == press.delimiter_token_id).sum()) == 1
    cache = DynamicCache()
    with torch.no_grad(), press(model):
        model(ids, past_key_values=cache)
    assert press.window_size == n_q
    assert cache.get_seq_length() == int((n_ctx + n_q) * 0.5)         # the delimiter never reached the cache
    assert len(model.model.embed_tokens._forward_hooks) == 0 and all(len(l.self_attn._forward_hooks) == 0 for l in model.model.layers)
    with pytest.raises(AssertionError):                                # two delimiters
        bad = torch.cat([ids, ids[:, n_ctx:n_ctx + 1], ids[:, -1:]], dim=1)
        with torch.no_grad(), press(model):
            model(bad, past_key_values=DynamicCache())


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_finch_kernel_vs_oracle(name):
    """Identical RoPE'd question queries (rounded to the case dtype) into oracle and kvp_finch_score."""
    from kvpress_amd import _native

    s = _inputs.make_finch_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    q = _inputs.round_to(gold(name)["qwin_f32"], s["dtype"])
    want = O.finch_score(q, s["keys"], s["normalize"])
    W = s["W"]
    one, zero = torch.ones((1, W, s["D"]), dtype=dt, device=DEV), torch.zeros((1, W, s["D"]), dtype=dt, device=DEV)
    got = _native.finch_score(torch.from_numpy(q).to(device=DEV, dtype=dt), one, zero,
                              torch.from_numpy(s["keys"]).to(device=DEV, dtype=dt), s["normalize"]).cpu().numpy()
    _close(got, want, W, RTOL)
    fill = np.float32(got[..., :-W].max()) + np.float32(1.0)
    assert np.all(got[..., -W:] == fill)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_press_matches_reference_gpu_fp32(name):
    _check_fp32(_inputs.make_finch_case(name), name, DEV, RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in NAMES if _inputs.FINCH_CASES[n]["dtype"] != "f32"])
def test_press_native_dtype_gpu(name):
    """bf16 / f16 model: fp32 scores from the model-dtype queries give a valid top-k of the float32 reference run (a few
    near-ties may swap), and the re-rotated keys follow the reference's per-op rounding."""
    s = _inputs.make_finch_case(name)
    g = gold(name)
    sc, out = _run_press(s, name, DEV, _inputs.torch_dtype(s["dtype"]))
    ulp = 2.0 ** (-8 if s["dtype"] == "bf16" else -11)
    for i, r, ko, pos in out:
        ref = g[f"pos_f32_{i}"]
        assert pos.shape == ref.shape and np.array_equal(pos, np.sort(pos, axis=-1))
        same = np.mean([len(np.intersect1d(a, b)) / a.size for a, b in zip(pos.reshape(-1, pos.shape[-1]), ref.reshape(-1, ref.shape[-1]))])
        assert same >= 0.97, f"{name} r={r}: overlap {same:.3f} with the float32 reference run"
        if s["rerotate"]:   # same rounding as the reference's native-dtype run wherever both kept the same token at the same rank
            refn, kn = g[f"pos_nat_{i}"], g[f"kout_nat_{i}"]
            m = pos == refn
            if m.mean() > 0.5:
                # A rotation preserves the norm of every (d, d + D/2) pair, and a one-ulp difference in cos / sin (float32 cosf is
                # accurate to ~1 ulp on either side, then rounded to the key dtype) moves an output by up to one ulp OF THAT NORM --
                # more than an ulp of the output itself where the two products cancel.  So: differences relative to the pair norm.
                a, r = ko[m], kn[m]
                half = r.shape[-1] // 2
                pn = np.sqrt(r[..., :half] ** 2 + r[..., half:] ** 2)
                d = np.abs(a - r) / np.maximum(np.concatenate([pn, pn], axis=-1), 1e-3)
                assert np.mean(a != r) < 5e-3 and d.max() <= 2.1 * ulp
