"""Pin the numpy oracle (oracle/kvpress_oracle.py) against the outputs of the REAL reference
(tests/golden/*.npz, produced by oracle/gen_golden.py from /root/reference).  CPU only."""
import os

import numpy as np
import pytest

import _inputs
from oracle import kvpress_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL = list(_inputs.CASES)


def load(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def oracle_scores(s, cos=None, sin=None):
    """Oracle chain for a case, from the same seeded inputs the reference saw."""
    if s["kind"] == "knorm":
        return O.knorm_score(s["keys"])
    if s["kind"] == "keydiff":
        return O.keydiff_score(s["keys"])
    if s["kind"] == "lagkv":
        return O.lagkv_score(s["keys"], s["values"], s["n_sink"], s["lag"], s.get("cross", False))
    if s["kind"] == "observed":
        return O.observed_attention_score(_inputs.make_attentions(s), s["H"])
    if s["kind"] == "qfilter":
        return O.qfilter_score(s["keys"], _inputs.make_qfilters(s)[_inputs.QF_LAYER])
    if s["kind"] == "cur":
        return O.cur_score(s["keys"], s["values"], s["leverage"], s.get("local", True), s.get("window", 16), s.get("sinks", 4))
    if s["kind"] == "streaming":
        return O.streaming_llm_score(s["B"], s["H"], s["S"], 0.5, s["n_sink"])  # the fixture's scores are those of ratio 0.5
    import torch

    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, torch.float32)
    cos, sin = cos.numpy(), sin.numpy()
    if s["kind"] in ("snapkv", "pyramid"):  # PyramidKV scores are SnapKV's; only the budget differs
        q = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos, sin, s["Hq"], s["D"], s["W"])
        return O.snapkv_score(q, s["keys"], s["ks"])
    if s["kind"] == "tova":
        q = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos, sin, s["Hq"], s["D"], 1)
        return O.tova_score(q, s["keys"])
    # ea: q = q_proj(hidden[:, n_sink:])  (expected_attention_press.py:70-71, utils.py:43-46)
    h = s["hidden"][:, s["n_sink"]:].astype(np.float64)
    q = (h @ s["wq"].astype(np.float64).T).reshape(s["B"], -1, s["Hq"], s["D"]).transpose(0, 2, 1, 3)
    mu, cov = O.ea_query_stats(q, s["use_covariance"])
    pos = torch.arange(s["S"], s["S"] + s["n_future"])[None]
    c, si = rot(torch.zeros(1, dtype=torch.float32), pos)
    mu, cov = O.ea_avg_rope(mu, cov, c[0].numpy(), si[0].numpy())
    return O.ea_score(s["keys"], s["values"], mu, cov, s["n_sink"], s["use_vnorm"], s["epsilon"])


@pytest.mark.parametrize("name", ALL)
def test_oracle_scores_match_reference(name):
    s = _inputs.make_case(name)
    g = load(name)
    sc = oracle_scores(s)
    ref = g["scores_f32"]
    assert sc.shape == ref.shape == (s["B"], s["H"], s["S"])
    if s["kind"] == "lagkv":
        return _inputs.assert_lag_scores_close(sc, ref, s, name)
    # float32 reference vs float64 oracle: both approximate the same math
    # (KeyDiff: a cosine in [-1, 1] that crosses zero -> absolute tolerance)
    np.testing.assert_allclose(sc, ref, rtol=2e-4, atol=2e-6 if s["kind"] in ("keydiff", "qfilter") else 1e-30)


@pytest.mark.parametrize("name", ALL)
def test_oracle_topk_matches_reference(name):
    s = _inputs.make_case(name)
    g = load(name)
    ref = g["scores_f32"]
    for i, r in enumerate(s["ratios"]):
        if s["kind"] == "pyramid":
            n = O.pyramidkv_budget(s["S"], r, s["W"], s["beta"], s["n_layers"], s["layer_idx"])
            assert n == int(g[f"nkept_{i}"]), "per-layer budget (pyramidkv_press.py:47-81)"
        else:
            n = O.n_kept(s["S"], r)
            assert n == int(g[f"nkept_{i}"]), "n_kept = int(S*(1-r)) (scorer_press.py:94)"
        if s["kind"] == "streaming":
            ref = O.streaming_llm_score(s["B"], s["H"], s["S"], r, s["n_sink"])
        idx = O.topk_select(ref, n)
        gold = g[f"idx_f32_{i}"]
        assert idx.shape == gold.shape
        # torch.topk's set is a valid top-k of the same scores, and so is ours
        ok, msg = O.topk_is_valid(ref, gold, n)
        assert ok, msg
        ok, msg = O.topk_is_valid(ref, idx, n)
        assert ok, msg
        # identical wherever the threshold value is unique (no tie at the k-th score)
        for b in range(s["B"]):
            for h in range(s["H"]):
                row = ref[b, h]
                if n == 0 or n == s["S"]:
                    continue
                t = np.sort(row)[::-1][n - 1]
                if (row == t).sum() == 1:
                    assert np.array_equal(idx[b, h], gold[b, h])


def test_tie_rule_lowest_position_wins():
    sc = np.array([[1.0, 3.0, 2.0, 2.0, 2.0, 0.5, 2.0, -0.0, 0.0]], dtype=np.float32)
    assert O.topk_select(sc, 3).tolist() == [[1, 2, 3]]
    assert O.topk_select(sc, 1).tolist() == [[1]]
    assert O.topk_select(sc, 6).tolist() == [[0, 1, 2, 3, 4, 6]]
    # -0.0 == +0.0 (torch semantics): lowest position first
    assert O.topk_select(sc, 8).tolist() == [[0, 1, 2, 3, 4, 5, 6, 7]]
    assert O.topk_select(sc, 0).shape == (1, 0)
    ok, _ = O.topk_is_valid(sc, np.array([[1, 3, 6]]), 3)
    assert ok
    ok, _ = O.topk_is_valid(sc, np.array([[1, 0, 2]]), 3)
    assert not ok


def test_n_kept_python_double_semantics():
    assert O.n_kept(131072, 0.7) == 39321
    assert O.n_kept(23, 0.4) == 13
    assert O.n_kept(10, 0.9) == 0
    assert O.n_kept(256, 0.1) == 230
    assert O.n_kept(32768, 0.5) == 16384


def test_snapkv_from_attentions_path_equals_computed():
    s = _inputs.make_case("sk_tiny")
    rs = np.random.RandomState(5)
    q = rs.standard_normal((s["B"], s["Hq"], s["W"], s["D"]))
    a = O.snapkv_score(q, s["keys"], s["ks"])
    # build a full [B,Hq,S,S] attention whose last W rows are the window attention
    attn = np.zeros((s["B"], s["Hq"], s["S"], s["S"]))
    wa = O.snapkv_window_attention(q, s["keys"])
    attn[:, :, -s["W"]:, : s["S"] - s["W"]] = wa
    b = O.snapkv_score_from_attentions(attn, s["H"], s["W"], s["ks"])
    np.testing.assert_allclose(a, b, rtol=1e-6)


# ---- the plain-PyTorch restatement that bench.py times as the CPU baseline (oracle/torch_path.py) ---------------------
TORCH_PATH_CASES = [n for n in ALL if _inputs.spec(n)["kind"] in ("knorm", "snapkv", "ea")
                    and _inputs.spec(n)["use_covariance"] and _inputs.spec(n)["use_vnorm"] and _inputs.spec(n)["epsilon"] == 0.0]


@pytest.mark.parametrize("name", TORCH_PATH_CASES)
def test_torch_path_is_the_reference_op_for_op(name):
    """oracle/torch_path.py run on the seeded inputs reproduces the REAL reference's scores bit for bit, in float32 mode and
    in the case's native dtype (same torch ops in the same order on the same CPU backend)."""
    import torch

    from oracle import torch_path as TP

    s = _inputs.make_case(name)
    g = load(name)
    for mode, dt in (("f32", torch.float32), ("nat", _inputs.torch_dtype(s["dtype"]))):
        att, rot, hidden, pe = _inputs.build_llama_attention(s, dt)
        keys, values = torch.from_numpy(s["keys"]).to(dt), torch.from_numpy(s["values"]).to(dt)
        kw = {"position_embeddings": pe}
        with torch.no_grad():
            if s["kind"] == "knorm":
                sc = TP.knorm_score(att, hidden, keys, values, kw)
            elif s["kind"] == "snapkv":
                sc = TP.snapkv_score(att, hidden, keys, values, kw, W=s["W"], ks=s["ks"])
            else:
                sc = TP.ea_score(att, hidden, keys, values, kw, n_future=s["n_future"], n_sink=s["n_sink"])
        ref = torch.from_numpy(g[f"scores_{mode}"])
        assert torch.equal(sc.float(), ref), f"{name}/{mode}: max diff {(sc.float() - ref).abs().max()}"
        if mode == "f32":
            for i, r in enumerate(s["ratios"]):
                ko, vo, idx = TP.torch_compress(lambda *a: sc, r, att, hidden, keys, values, kw)
                assert ko.shape[2] == int(g[f"nkept_{i}"])
                assert np.array_equal(np.sort(idx.numpy(), -1), g[f"idx_f32_{i}"])
