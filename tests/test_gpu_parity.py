"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs,
and vs the committed outputs of the real reference (tests/golden).  Run with `-m gpu` on MI355X.

Contract:
  * float scores: relative error <= 1e-3 against the oracle / the fp32-mode reference
    (pad positions: equal to max+1);
  * top-k: bit-exact index equality when kernel and oracle select from the SAME float32 scores;
    tie-tolerant set validity when the scores come from different float32 pipelines;
  * gather: bit-exact.
"""
import os

import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
RTOL = 1e-3  # north_star tolerance on float scores

KN = [n for n, c in _inputs.CASES.items() if c["kind"] == "knorm"]
SK = [n for n, c in _inputs.CASES.items() if c["kind"] == "snapkv"]
EA = [n for n, c in _inputs.CASES.items() if c["kind"] == "ea"]


def gold(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def to_dev(x, dtype_name):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device=DEV, dtype=_inputs.torch_dtype(dtype_name))


def assert_scores_close(got, want, rtol=RTOL, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite scores"
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
    i = np.unravel_index(np.argmax(err), err.shape)
    assert err.max() <= rtol, f"{what}: max rel err {err.max():.3e} at {i}: got {got[i]!r} want {want[i]!r}"


def native():
    from kvpress_amd import _native

    return _native


# ---------------------------------------------------------------------------------------------
# kernel level: same inputs into oracle and kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", KN)
def test_rownorm_vs_oracle(name):
    s = _inputs.make_case(name)
    k = to_dev(s["keys"], s["dtype"])
    got = native().rownorm_score(k, -1.0).cpu().numpy()
    assert_scores_close(got, O.knorm_score(s["keys"]), 1e-5, name)


def test_rownorm_strided_views():
    s = _inputs.make_case("kn_bf16_A")
    k = to_dev(s["keys"], "bf16")
    want = O.knorm_score(s["keys"])
    # sliced cache views as wrapper presses pass them: sink-stripped and chunked
    got = native().rownorm_score(k[:, :, 4:], -1.0).cpu().numpy()
    assert_scores_close(got, want[:, :, 4:], 1e-5)
    got = native().rownorm_score(k[:, 1:3, 100:1000], 1.0).cpu().numpy()
    assert_scores_close(got, -want[:, 1:3, 100:1000], 1e-5)
    # non-unit last stride -> binding makes it contiguous
    kt = k.transpose(2, 3).contiguous().transpose(2, 3)
    got = native().rownorm_score(kt, -1.0).cpu().numpy()
    assert_scores_close(got, want, 1e-5)


@pytest.mark.parametrize("name", list(_inputs.CASES))
def test_topk_bitexact_on_reference_scores(name):
    """Same float32 scores (the reference's own O32 scores) into oracle and kernel: identical indices."""
    s = _inputs.spec(name)
    g = gold(name)
    ref = g["scores_f32"]
    sc = torch.from_numpy(ref).to(DEV)
    for i, r in enumerate(s["ratios"]):
        n = O.n_kept(s["S"], r)
        got = native().topk_select(sc, n).cpu().numpy()
        want = O.topk_select(ref, n)
        assert got.dtype == np.int32 and got.shape == want.shape
        assert np.array_equal(got, want), f"{name} r={r}: indices differ from the oracle"
        ok, msg = O.topk_is_valid(ref, got, n)
        assert ok, msg


@pytest.mark.parametrize("name", ["kn_bf16_A", "sk_4096", "ea_1500_B", "st_tiny", "kd_f16_d64"])
def test_topk_order_score_matches_torch_topk(name):
    """KVP_ORDER_SCORE: descending score, ties by ascending position -- on the reference's own float32 scores the kernel
    returns the oracle's order, which is torch.topk(sorted=True)'s wherever the scores are distinct."""
    s = _inputs.spec(name)
    g = gold(name)
    ref = g["scores_f32"]
    sc = torch.from_numpy(ref).to(DEV)
    N = native()
    for r in s["ratios"]:
        n = O.n_kept(s["S"], r)
        got = N.topk_select(sc, n, N.ORDER_SCORE).cpu().numpy()
        want = O.topk_select_by_score(ref, n)
        assert got.dtype == np.int32 and np.array_equal(got, want), f"{name} r={r}"
        t = torch.from_numpy(ref).topk(n, dim=-1)
        vals = np.take_along_axis(ref, got.astype(np.int64), axis=-1)
        assert np.array_equal(vals, t.values.numpy())          # same score sequence as torch.topk(sorted=True)
        # a position whose score differs from both neighbours' is in no tie run: there the index must be torch's too
        ti = t.indices.numpy()
        pad = np.ones(vals.shape[:-1] + (1,), bool)
        dl = np.concatenate([pad, np.diff(vals, axis=-1) != 0], axis=-1)      # differs from the left neighbour
        dr = np.concatenate([np.diff(vals, axis=-1) != 0, pad], axis=-1)      # ... from the right one
        alone = dl & dr
        assert np.array_equal(got[alone], ti[alone]), f"{name} r={r}: index order differs outside tie runs"
        assert alone.any() or name == "st_tiny"           # (StreamingLLM's 0 / 1 scores are one tie run; every other case has distinct scores)
    # smallest-first variant
    n = s["S"] // 3
    got = N.topk_select(sc, n, N.ORDER_SCORE | N.TOPK_SMALLEST).cpu().numpy()
    assert np.array_equal(got, O.topk_select_by_score(-ref, n))


def test_topk_heavy_ties_and_edges():
    rs = np.random.RandomState(3)
    # bf16-like scores: ~90 distinct values, >1000 ties at the threshold (SURVEY hard part 1)
    base = _inputs.round_to(-np.sqrt(rs.chisquare(128, size=(8, 32768))).astype(np.float32), "bf16")
    cases = [(base, [1, 5, 16384, 32767, 32768]),
             (np.zeros((3, 5000), np.float32), [1, 2500, 4999]),                 # all equal
             (np.where(rs.rand(2, 4099) < 0.5, -0.0, 0.0).astype(np.float32), [7, 2050]),  # -0.0 == +0.0
             (rs.standard_normal((1, 1)).astype(np.float32), [1]),
             (rs.standard_normal((5, 2049)).astype(np.float32), [1, 1024, 2048]),
             (np.concatenate([rs.standard_normal((2, 3000)), np.full((2, 50), np.inf), np.full((2, 50), -np.inf)], 1).astype(np.float32), [10, 50, 60, 3075]),
             ]
    for sc_np, ks in cases:
        sc = torch.from_numpy(sc_np).to(DEV)
        for k in ks:
            got = native().topk_select(sc, k).cpu().numpy()
            want = O.topk_select(sc_np, k)
            assert np.array_equal(got, want), f"shape {sc_np.shape} k={k}"
    # k = 0 and strided rows
    sc = torch.from_numpy(base).to(DEV)
    assert native().topk_select(sc, 0).shape == (8, 0)
    view = sc[:, 100:20100]
    got = native().topk_select(view, 777).cpu().numpy()
    assert np.array_equal(got, O.topk_select(base[:, 100:20100], 777))


@pytest.mark.parametrize("R,S,k", [(1, 5, 3), (2, 64, 64), (3, 3000, 2048), (3, 3000, 2049), (2, 9000, 4097), (5, 40000, 13001), (8, 131072, 65536),
                                   (8, 131008, 65472), (2, 200000, 65537), (2, 200000, 91750), (1, 262144, 131072), (1, 300000, 131073), (2, 300000, 140000)])
def test_score_order_sort_shapes(R, S, k):
    """KVP_ORDER_SCORE's hand-written segmented sort (topk_order.hip) over its three regimes -- one tile per row (k <= 2048), tiles +
    sampled buckets with 2 or 4 composites per thread (k <= 65536 / <= 131072), the global merge network beyond -- on wide scores,
    flat ones (half a row in one radix bin), heavy ties (bf16-rounded values: equal scores keep ascending positions) and constant
    rows; descending score with ties by position, k smallest included: the oracle's element order exactly."""
    rs = np.random.RandomState(R * 7919 + S + k)
    N = native()
    wide = rs.standard_normal((R, S)).astype(np.float32)
    flat = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal((R, S)))).astype(np.float32)
    ties = _inputs.round_to(rs.standard_normal((R, S)).astype(np.float32), "bf16")
    const = np.full((R, S), -0.5, np.float32)
    for name, sc_np in (("wide", wide), ("flat", flat), ("ties", ties), ("const", const)):
        t = torch.from_numpy(sc_np).to(DEV)
        got = N.topk_select(t, k, N.ORDER_SCORE).cpu().numpy()
        assert np.array_equal(got, O.topk_select_by_score(sc_np, k)), f"{name} R={R} S={S} k={k}"
    got = N.topk_select(torch.from_numpy(ties).to(DEV), k, N.ORDER_SCORE | N.TOPK_SMALLEST).cpu().numpy()
    assert np.array_equal(got, O.topk_select_by_score(-ties, k)), f"smallest R={R} S={S} k={k}"
    view = torch.from_numpy(wide).to(DEV)[:, 1:S - 1] if S > 4 else None          # strided, unaligned rows
    if view is not None and k <= S - 2:
        got = N.topk_select(view, k, N.ORDER_SCORE).cpu().numpy()
        assert np.array_equal(got, O.topk_select_by_score(wide[:, 1:S - 1], k))


@pytest.mark.parametrize("S", [32769, 49153, 65536, 100003, 131072])
def test_topk_second_pass_variants(S, knobs):
    """Rows beyond 32768: the cluster select (one launch; default) and, with KVP_TK_CLUSTER=0, the (chunk, row) passes (what devices
    with fewer than 256 CUs, more than 8 rows or longer rows run): the oracle's indices every time, on wide, flat (one exponent, heavy
    ties: half the row in the threshold's first-digit bin) and constant rows, k smallest included."""
    rs = np.random.RandomState(S)
    N = native()
    wide = rs.standard_normal((3, S)).astype(np.float32)
    flat = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal((3, S)))).astype(np.float32)   # pooled-attention-like: +-5 % around one value
    ties = _inputs.round_to(-np.sqrt(rs.chisquare(64, size=(3, S))).astype(np.float32), "bf16")
    const = np.full((2, S), 0.25, np.float32)
    variants = [dict(KVP_TK_CLUSTER=None), dict(KVP_TK_CLUSTER=0)]
    for variant in variants:
        knobs(**variant)
        for sc_np in (wide, flat, ties, const):
            t = torch.from_numpy(sc_np).to(DEV)
            for k in sorted({1, S // 3, S // 2, S - 1}):
                got = N.topk_select(t, k).cpu().numpy()
                assert np.array_equal(got, O.topk_select(sc_np, k)), f"variant {variant} S={S} k={k}"
        got = N.topk_select(torch.from_numpy(flat).to(DEV), S // 2, N.ORDER_POSITION | N.TOPK_SMALLEST).cpu().numpy()
        assert np.array_equal(got, O.topk_select(-flat, S // 2)), f"variant {variant} S={S} smallest"


@pytest.mark.parametrize("R,S", [(1, 16385), (3, 20000), (8, 32769), (7, 40000), (9, 40000), (8, 131008), (2, 131073), (5, 262144), (2, 262145),
                                 (16, 131008), (17, 40000), (32, 65537), (33, 20000)])
def test_topk_cluster_select_rows_and_lengths(R, S, knobs):
    """The cluster select (topk_cluster.hip: 32 workgroups per row, keys in registers, cluster barriers between the digit steps)
    over its whole range of row lengths (every keys-per-thread instantiation, partially filled last slots, unaligned rows), with
    fewer rows than clusters and -- round 6: batches of more than one element -- with MORE rows than the 8 clusters the device holds
    at once (9, 16, 17, 32 rows: one launch, the later clusters start as the first retire), twice through the same self-cleaning
    workspace, k smallest, score order; 33 rows or 262145 scores are past its range and take the (chunk, row) passes.  Against the
    oracle AND bit-identical to KVP_TK_CLUSTER=0."""
    rs = np.random.RandomState(R * 1000003 + S)
    N = native()
    flat = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal((R, S)))).astype(np.float32)
    ties = _inputs.round_to(rs.standard_normal((R, S)).astype(np.float32), "bf16")
    for sc_np in (flat, ties):
        t = torch.from_numpy(sc_np).to(DEV)
        for k in sorted({1, S // 2, S - 1}) * 2:
            knobs(KVP_TK_CLUSTER=None)
            got = N.topk_select(t, k)
            knobs(KVP_TK_CLUSTER=0)
            legacy = N.topk_select(t, k)
            assert torch.equal(got, legacy), f"R={R} S={S} k={k}: cluster != passes"
            if R * S <= 8 * 131072 or (k == S // 2 and sc_np is flat):
                assert np.array_equal(got.cpu().numpy(), O.topk_select(sc_np, k)), f"R={R} S={S} k={k}"
    knobs(KVP_TK_CLUSTER=None)
    t = torch.from_numpy(flat).to(DEV)
    if R * S <= 8 * 131072:
        got = N.topk_select(t, S // 3, N.ORDER_POSITION | N.TOPK_SMALLEST).cpu().numpy()
        assert np.array_equal(got, O.topk_select(-flat, S // 3))
        got = N.topk_select(t, S // 3, N.ORDER_SCORE).cpu().numpy()
        assert np.array_equal(got, O.topk_select_by_score(flat, S // 3))
    view = t[:, 3:S - 2]                                   # unaligned, strided rows
    got = N.topk_select(view, (S - 5) // 2)
    knobs(KVP_TK_CLUSTER=0)
    assert torch.equal(got, N.topk_select(view, (S - 5) // 2))


@pytest.mark.parametrize("S", [1, 2, 63, 1023, 1024, 1025, 2048, 2049, 4096, 4097, 8191, 8193, 16383, 16384, 16385, 20000, 32767, 32768, 32769, 40000])
def test_topk_short_rows_one_workgroup(S):
    """Rows up to 32768 are selected by ONE workgroup per row (topk_row_kernel, every elements-per-thread variant, aligned
    and unaligned rows); longer ones by the multi-workgroup passes: same indices either way, bit-exact against the oracle."""
    rs = np.random.RandomState(S)
    N = native()
    R = 5
    wide = rs.standard_normal((R, S + 3)).astype(np.float32)
    flat = _inputs.round_to(-np.sqrt(rs.chisquare(64, size=(R, S + 3))).astype(np.float32), "bf16")   # heavy ties, one exponent
    for sc_np in (wide, flat):
        t = torch.from_numpy(sc_np).to(DEV)
        for off in (0, 1, 3):                                  # row start alignment (float4 fast path or not)
            view, ref = t[:, off:off + S], sc_np[:, off:off + S]
            for k in sorted({1, max(1, S // 7), max(1, S // 2), max(1, S - 1), S}):
                got = N.topk_select(view, k).cpu().numpy()
                assert np.array_equal(got, O.topk_select(ref, k)), f"S={S} off={off} k={k}"
            k = max(1, S // 3)
            got = N.topk_select(view, k, N.ORDER_POSITION | N.TOPK_SMALLEST).cpu().numpy()
            assert np.array_equal(got, O.topk_select(-ref, k)), f"S={S} off={off} smallest"
    if S >= 4:   # segments as rows: positions are reported relative to the outer row (+ pos_base)
        seg = S // 4
        sc_np = np.ascontiguousarray(wide[:, : 4 * seg])
        k = max(1, seg // 2)
        got = N.topk_select_segmented(torch.from_numpy(sc_np).to(DEV), seg, k, pos_base=11).cpu().numpy()
        want = np.concatenate([11 + c * seg + O.topk_select(sc_np[:, c * seg:(c + 1) * seg], k) for c in range(4)], axis=-1)
        assert np.array_equal(got, want), f"S={S} segmented"


@pytest.mark.parametrize("name", ["kn_tiny_d6", "kn_bf16_A", "kn_f16_ragged", "kn_d96_bf16", "kn_opt_geom"])
def test_gather_bitexact(name):
    s = _inputs.make_case(name)
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    sc = O.knorm_score(s["keys"])
    for r in s["ratios"]:
        idx = O.topk_select(sc, O.n_kept(s["S"], r))
        ko, vo = native().gather_kv(k, v, torch.from_numpy(idx).to(DEV))
        wk, wv = O.gather_kv(s["keys"], s["values"], idx)
        assert ko.is_contiguous() and vo.is_contiguous() and ko.dtype == k.dtype
        assert np.array_equal(ko.float().cpu().numpy(), wk) and np.array_equal(vo.float().cpu().numpy(), wv)
    # unsorted / repeated indices and sliced sources
    rs = np.random.RandomState(0)
    idx = rs.randint(0, s["S"] - 4, size=(s["B"], s["H"], 37)).astype(np.int32)
    ko, vo = native().gather_kv(k[:, :, 4:], v[:, :, 4:], torch.from_numpy(idx).to(DEV))
    wk, wv = O.gather_kv(s["keys"][:, :, 4:], s["values"][:, :, 4:], idx)
    assert np.array_equal(ko.float().cpu().numpy(), wk) and np.array_equal(vo.float().cpu().numpy(), wv)


@pytest.mark.parametrize("name", SK)
def test_snapkv_kernel_vs_oracle(name):
    """Identical RoPE'd window queries (rounded to the case dtype) into oracle and kernel."""
    s = _inputs.make_case(name)
    g = gold(name)
    q = _inputs.round_to(g["qwin_f32"], s["dtype"])
    want = O.snapkv_score(q, s["keys"], s["ks"])
    got = native().snapkv_score(to_dev(q, s["dtype"]), to_dev(s["keys"], s["dtype"]), s["ks"]).cpu().numpy()
    W = s["W"]
    assert_scores_close(got[..., :-W], want[..., :-W], RTOL, name)
    # window = max + 1 (global max over B and H)
    fill = np.float32(got[..., :-W].max()) + np.float32(1.0)
    assert np.all(got[..., -W:] == fill)
    for r in s["ratios"]:
        n = O.n_kept(s["S"], r)
        idx = native().topk_select(torch.from_numpy(got).to(DEV), n).cpu().numpy()
        assert np.array_equal(idx, O.topk_select(got, n))       # bit-exact on its own scores
        ok, msg = O.topk_is_valid(want, idx, n, rel_band=1e-4)  # and a valid top-k of the oracle's scores
        assert ok, f"{name} r={r}: {msg}"


@pytest.mark.parametrize("D", [128, 96, 64, 256])
@pytest.mark.parametrize("W,S,G,ks", [(1, 700, 4, 1), (1, 5000, 2, 1), (7, 300, 4, 5), (32, 1000, 4, 5), (33, 2100, 1, 3), (63, 64, 4, 5), (64, 65, 3, 5),
                                      (65, 700, 4, 5), (100, 4200, 4, 5), (128, 129, 2, 5), (130, 9000, 8, 7), (200, 2500, 4, 5), (257, 40000, 4, 5),
                                      (64, 3000, 5, 5), (64, 3000, 6, 5), (40, 1500, 7, 5), (64, 20000, 1, 5), (64, 20000, 2, 5),
                                      # S < Wp (the padded window is longer than the sequence): the fuzz's round-6 find
                                      (200, 253, 7, 5), (65, 66, 4, 1), (130, 150, 2, 3), (1, 2, 4, 1), (10, 40, 8, 5),
                                      # more than two group-blocks of four q-heads per kv-head (Llama-3.1-405B, Qwen3-235B: G = 16; Mistral-Large: 12)
                                      (64, 3000, 12, 5), (64, 3000, 16, 5), (40, 1500, 9, 5), (130, 5000, 13, 3)])
def test_snapkv_any_window_on_the_mfma_path(W, S, G, ks, D):
    """Round 6: the MFMA passes take ANY window size (TOVA's W = 1, FINCH's question length, user-chosen windows) as blocks of 64
    padded rows -- padding in front, normaliser +inf, the reference's causal rule in padded coordinates (snapkv_internal.h) -- for
    head sizes 256, 128, 96 and 64, G = 1 .. 16 (hand-scheduled loops for D = 128 and G % 4 == 0, the compiler-scheduled kernels otherwise), down to
    S = W + 1 (every tile masked).  Scores against the float64 oracle; pad columns; a batch of two through the same call."""
    rs = np.random.RandomState(W * 131 + S + G + D)
    N = native()
    B, H = 2, 2
    q = _inputs.round_to((rs.standard_normal((B, H * G, W, D)) * 1.5).astype(np.float32), "bf16")
    k = rs.standard_normal((B, H, S, D)).astype(np.float32)
    k[1, :, : S // 3] *= 3.0   # uneven logits: the lazy offset's raises, rows whose maximum sits in the masked tail
    k = _inputs.round_to(k, "bf16")
    want = O.snapkv_score(q, k, ks)
    got = N.snapkv_score(to_dev(q, "bf16"), to_dev(k, "bf16"), ks).cpu().numpy()
    assert_scores_close(got[..., :-W], want[..., :-W], RTOL, f"W={W} S={S} G={G} D={D}")
    fill = np.float32(got[..., :-W].max()) + np.float32(1.0)
    assert np.all(got[..., -W:] == fill)
    one = N.snapkv_score(to_dev(q[:1], "bf16"), to_dev(k[:1], "bf16"), ks).cpu().numpy()
    assert_scores_close(one[..., :-W], want[:1, :, :-W], RTOL, f"W={W} S={S} G={G} D={D} (B = 1)")


@pytest.mark.parametrize("S,Hq", [(192, 4), (4160, 4), (33000, 2), (40000, 8)])
def test_snapkv_kernel_many_chunks(S, Hq):
    """One kv-head: the MFMA passes split S into as many single-tile workgroups as fit (up to 256) - the partial
    statistics workspace and the all-masked partials of the window tiles are what this exercises."""
    rs = np.random.RandomState(S)
    q = _inputs.round_to(rs.standard_normal((1, Hq, 64, 128)).astype(np.float32), "bf16")
    k = _inputs.round_to(rs.standard_normal((1, 1, S, 128)).astype(np.float32), "bf16")
    want = O.snapkv_score(q, k, 5)
    got = native().snapkv_score(to_dev(q, "bf16"), to_dev(k, "bf16"), 5).cpu().numpy()
    assert_scores_close(got[..., :-64], want[..., :-64], RTOL, f"S={S}")


@pytest.mark.parametrize("name", ["sk_tiny", "sk_257_A", "sk_f16_d64"])
def test_snapkv_from_attentions(name):
    s = _inputs.make_case(name)
    g = gold(name)
    q = _inputs.round_to(g["qwin_f32"], s["dtype"])
    wa = O.snapkv_window_attention(q, s["keys"])                  # [B,Hq,W,S-W]
    S, W = s["S"], s["W"]
    attn = np.zeros((s["B"], s["Hq"], S, S), np.float32)
    attn[:, :, -W:, : S - W] = wa
    attn_t = torch.from_numpy(attn).to(DEV)
    want = O.snapkv_score_from_attentions(attn, s["H"], W, s["ks"])
    got = native().snapkv_score_from_attn(attn_t[..., -W:, :-W], s["H"], S, s["ks"]).cpu().numpy()
    assert_scores_close(got[..., :-W], want[..., :-W], 1e-5, name)
    assert np.all(got[..., -W:] == np.float32(got[..., :-W].max()) + np.float32(1.0))


def _ea_q(s):
    """pre-RoPE queries of the sink-stripped hidden states, rounded to the case dtype"""
    h = s["hidden"][:, s["n_sink"]:].astype(np.float64)
    q = (h @ s["wq"].astype(np.float64).T).reshape(s["B"], -1, s["Hq"], s["D"]).transpose(0, 2, 1, 3)
    return _inputs.round_to(q.astype(np.float32), s["dtype"])


@pytest.mark.parametrize("name", EA)
def test_ea_qstats_vs_oracle(name):
    s = _inputs.make_case(name)
    q = _ea_q(s)
    mu_w, cov_w = O.ea_query_stats(q, s["use_covariance"])
    # [B,S,Hq*D] storage viewed [B,Hq,S,D]: the layout q_proj produces (no copy in the binding)
    qt = to_dev(np.ascontiguousarray(q.transpose(0, 2, 1, 3)), s["dtype"]).transpose(1, 2)
    mu, cov = native().ea_qstats(qt, s["use_covariance"])
    scale = np.abs(mu_w).max()
    assert np.abs(mu.cpu().numpy() - mu_w).max() <= 1e-5 * scale + 1e-6
    if s["use_covariance"]:
        c = cov.cpu().numpy()
        d = np.sqrt(np.einsum("bhii->bhi", cov_w))
        # Sq >= 4096 with 16-bit D=128 inputs runs on the matrix cores (exact products of the 16-bit inputs, fp32 accumulation of raw
        # second moments: the error grows with (mean / sigma)^2, see ea_mfma.hip)
        mfma = s["dtype"] != "f32" and s["D"] == 128 and q.shape[2] >= 4096
        tol = (1e-3 if mfma else 1e-4) * d[..., :, None] * d[..., None, :] + 1e-9
        assert (np.abs(c - cov_w) <= tol).all(), np.abs(c - cov_w).max()
    else:
        assert cov is None


@pytest.mark.parametrize("name", EA)
def test_ea_score_kernel_vs_oracle(name):
    """Same post-RoPE statistics (float32) into oracle and kernel."""
    s = _inputs.make_case(name)
    q = _ea_q(s)
    mu, cov = O.ea_query_stats(q, s["use_covariance"])
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
    pos = torch.arange(s["S"], s["S"] + s["n_future"])[None]
    c, si = rot(torch.zeros(1), pos)
    mu, cov = O.ea_avg_rope(mu, cov, c[0].numpy(), si[0].numpy())
    mu32 = mu.astype(np.float32)
    cov32 = cov.astype(np.float32) if cov is not None else None
    want = O.ea_score(s["keys"], s["values"], mu32, cov32, s["n_sink"], s["use_vnorm"], s["epsilon"])
    got = native().ea_score(to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"]),
                            torch.from_numpy(mu32).to(DEV), torch.from_numpy(cov32).to(DEV) if cov32 is not None else None,
                            s["n_sink"], s["use_vnorm"], s["epsilon"]).cpu().numpy()
    ns = s["n_sink"]
    assert_scores_close(got[..., ns:], want[..., ns:], RTOL, name)
    if ns:
        assert np.all(got[..., :ns] == np.float32(got[..., ns:].max()) + np.float32(1.0))


KD = [n for n, c in _inputs.CASES.items() if c["kind"] == "keydiff"]
CUR = [n for n, c in _inputs.CASES.items() if c["kind"] == "cur"]


@pytest.mark.parametrize("name", CUR)
def test_cur_kernel_vs_oracle(name):
    """kvp_cur_score in the case's dtype against the float64 restatement on the same (dtype-exact) keys and values."""
    s = _inputs.make_case(name)
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    local, w, sinks = s.get("local", True), s.get("window", 16), s.get("sinks", 4)
    got = native().cur_score(k, v, s["leverage"], w if local else 0, sinks).cpu().numpy()
    assert_scores_close(got, O.cur_score(s["keys"], s["values"], s["leverage"], local, w, sinks), 2e-5, name)
    for lev in ("key", "value", "kv_avg", "kv_product"):
        got = native().cur_score(k[:, :, ::2], v[:, :, ::2], lev, 5, 2).cpu().numpy()   # strided views, odd window
        assert_scores_close(got, O.cur_score(s["keys"][:, :, ::2], s["values"][:, :, ::2], lev, True, 5, 2), 2e-5, f"{name}/{lev}")
    with pytest.raises(ValueError):
        native().cur_score(k, v, "nope", 16, 4)


@pytest.mark.parametrize("name", [n for n, c in _inputs.CASES.items() if c["kind"] == "qfilter"])
def test_rowdot_kernel_vs_oracle(name):
    """kvp_rowdot_score in the case's dtype against the float64 restatement on the same (dtype-exact) keys and filters."""
    s = _inputs.make_case(name)
    k = to_dev(s["keys"], s["dtype"])
    f = _inputs.make_qfilters(s)[_inputs.QF_LAYER]
    ft = to_dev(f, s["dtype"])
    got = native().rowdot_score(k, ft, -1.0).cpu().numpy()
    np.testing.assert_allclose(got, O.qfilter_score(s["keys"], f), rtol=2e-5, atol=2e-5, err_msg=name)
    # strided key view and a filter slice of a larger tensor
    allf = to_dev(_inputs.make_qfilters(s), s["dtype"])
    got2 = native().rowdot_score(k[:, :, ::2], allf[_inputs.QF_LAYER], 2.0).cpu().numpy()
    np.testing.assert_allclose(got2, -2.0 * O.qfilter_score(s["keys"][:, :, ::2], f), rtol=2e-5, atol=4e-5, err_msg=name)


@pytest.mark.parametrize("name", [n for n, c in _inputs.CASES.items() if c["kind"] == "lagkv" and c["S"] >= c["n_sink"] + 2 * c["lag"]])
def test_lagkv_kernel_vs_oracle(name):
    """kvp_lagkv_score in the case's dtype against the float64 restatement on the same (dtype-exact) keys and values; the
    raw (cross_scoring) scores also on strided views."""
    s = _inputs.make_case(name)
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    for cross in (False, True):
        got = native().lagkv_score(k, v, s["n_sink"], s["lag"], cross).cpu().numpy()
        _inputs.assert_lag_scores_close(got, O.lagkv_score(s["keys"], s["values"], s["n_sink"], s["lag"], cross), dict(s, cross=cross), f"{name}/{cross}")
    S2 = s["S"] - 3
    got = native().lagkv_score(k[:, :, 3:], v[:, :, 3:], s["n_sink"], s["lag"], True).cpu().numpy()
    _inputs.assert_lag_scores_close(got, O.lagkv_score(s["keys"][:, :, 3:], s["values"][:, :, 3:], s["n_sink"], s["lag"], True),
                                    dict(s, cross=True, S=S2), name + "/view")
    with pytest.raises(Exception):
        native().lagkv_score(k[:, :, : s["n_sink"] + s["lag"]], v[:, :, : s["n_sink"] + s["lag"]], s["n_sink"], s["lag"], False)


@pytest.mark.parametrize("name", [n for n, c in _inputs.CASES.items() if c["kind"] == "observed"])
def test_observed_attention_kernel_vs_oracle(name):
    """kvp_observed_attention_score in the case's dtype against the float64 restatement on the same (dtype-exact) weights,
    also on a strided view (fewer query rows than keys: a later chunk of queries)."""
    s = _inputs.make_case(name)
    a = _inputs.make_attentions(s)
    t = to_dev(a, s["dtype"])
    assert_scores_close(native().observed_attention_score(t, s["H"]).cpu().numpy(), O.observed_attention_score(a, s["H"]), 2e-5, name)
    half = s["S"] // 2
    got = native().observed_attention_score(t[:, :, half:], s["H"]).cpu().numpy()
    assert_scores_close(got, O.observed_attention_score(a[:, :, half:], s["H"]), 2e-5, name + "/rows")


@pytest.mark.parametrize("name", KD)
def test_keydiff_kernel_vs_oracle(name):
    """kvp_keydiff_score in the case's dtype against the float64 restatement on the same (dtype-exact) keys."""
    s = _inputs.make_case(name)
    k = to_dev(s["keys"], s["dtype"])
    got = native().keydiff_score(k).cpu().numpy()
    np.testing.assert_allclose(got, O.keydiff_score(s["keys"]), rtol=0, atol=2e-6, err_msg=name)
    # strided view (every second token), the way wrapper presses hand over slices
    got2 = native().keydiff_score(k[:, :, ::2]).cpu().numpy()
    np.testing.assert_allclose(got2, O.keydiff_score(s["keys"][:, :, ::2]), rtol=0, atol=2e-6, err_msg=name)


def test_keydiff_is_deterministic_and_handles_zero_rows():
    rs = np.random.RandomState(3)
    k = rs.standard_normal((1, 2, 5000, 128)).astype(np.float32)
    k[0, 0, 17] = 0.0   # an all-zero key: normalize -> 0, cosine -> 0
    kd = to_dev(_inputs.round_to(k, "bf16"), "bf16")
    a = native().keydiff_score(kd)
    b = native().keydiff_score(kd)
    assert torch.equal(a, b)
    assert a[0, 0, 17].item() == 0.0
    np.testing.assert_allclose(a.cpu().numpy(), O.keydiff_score(_inputs.round_to(k, "bf16")), rtol=0, atol=2e-6)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_streaming_walks_agree(dtype):
    """The two walk shapes of the streaming reductions (rownorm.hip / keydiff.hip / cur.hip): rows of >= 4096 tokens take one contiguous
    slot per 1024-thread workgroup, shorter ones interleaved row groups.  A row's lanes, order and rounding are the same in both, so
    the norms of the first 3000 tokens are the SAME BITS whether they are computed inside the long tensor (slot walk) or from a
    3000-token view of it (interleaved walk); every result also matches the oracle, and CUR's combine is checked for window
    lengths it stages through LDS (divisors of 256) and one it does not."""
    rs = np.random.RandomState(31)
    N = native()
    for B, H, S, D in ((1, 8, 5000, 128), (2, 2, 20001, 64), (1, 1, 4096, 128), (1, 3, 9000, 40)):
        kn = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32), dtype)
        vn = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32), dtype)
        k, v = to_dev(kn, dtype), to_dev(vn, dtype)
        rn, kd = N.rownorm_score(k, -1.0), N.keydiff_score(k)
        assert_scores_close(rn.cpu().numpy(), O.knorm_score(kn), 1e-5)
        np.testing.assert_allclose(kd.cpu().numpy(), O.keydiff_score(kn), rtol=0, atol=2e-6)
        short = N.rownorm_score(k[:, :, :3000], -1.0)                     # a strided view: the interleaved walk
        assert torch.equal(short, rn[..., :3000]), (dtype, S, D)
        np.testing.assert_allclose(N.keydiff_score(k[:, :, :3000]).cpu().numpy(), O.keydiff_score(kn[:, :, :3000]), rtol=0, atol=2e-6)
        assert_scores_close(N.cur_score(k, v, "kv_product", 16, 4).cpu().numpy(), O.cur_score(kn, vn, "kv_product", True, 16, 4), 2e-5)
        for w in (2, 4, 64, 256, 5):
            assert_scores_close(N.cur_score(k, v, "kv_avg", w, 0).cpu().numpy(), O.cur_score(kn, vn, "kv_avg", True, w, 0), 2e-5)


def test_scores_head_mean():
    rs = np.random.RandomState(4)
    x = rs.standard_normal((3, 8, 1001)).astype(np.float32)
    t = torch.from_numpy(x).to(DEV)
    out = native().scores_head_mean_(t)
    assert out.data_ptr() == t.data_ptr()
    want = np.repeat(x.astype(np.float64).mean(1, keepdims=True), 8, axis=1)
    np.testing.assert_allclose(t.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    # a strided view: heads 1::2 of a larger tensor
    big = torch.from_numpy(rs.standard_normal((2, 6, 500)).astype(np.float32)).to(DEV)
    ref = big.clone()
    native().scores_head_mean_(big[:, 1::2])
    assert torch.equal(big[:, 0::2], ref[:, 0::2])
    np.testing.assert_allclose(big[:, 1::2].cpu().numpy(), np.repeat(ref[:, 1::2].cpu().numpy().mean(1, keepdims=True), 3, 1),
                               rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", [n for n, c in _inputs.CASES.items() if c["kind"] == "tova"])
def test_tova_from_attentions(name):
    """TOVA when the layer returns attention weights (tova_press.py:45-46): last row, all but the last column."""
    import kvpress_amd as P

    s = _inputs.make_case(name)
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, torch.float32)
    q = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos.numpy(), sin.numpy(), s["Hq"], s["D"], 1)
    wa = O.snapkv_window_attention(q, s["keys"])                       # [B,Hq,1,S-1]
    full = np.zeros((s["B"], s["Hq"], 3, s["S"]), dtype=np.float32)    # a 3-row stand-in for [.., S, S]
    full[:, :, -1:, :-1] = wa
    keys = to_dev(s["keys"], "f32")
    sc = P.TOVAPress(0.5).score(None, None, keys, keys, torch.from_numpy(full).to(DEV), {})
    want = O.tova_score(q, s["keys"])
    assert_scores_close(sc.cpu().numpy()[..., :-1], want[..., :-1], RTOL, name)
    assert (sc[..., -1:] > sc[..., :-1].amax()).all()


def test_random_press_properties():
    import kvpress_amd as P

    k = torch.randn(2, 4, 300, 64, device=DEV, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    p = P.RandomPress(compression_ratio=0.5, seed=7)
    s1, s2 = p.score(None, None, k, v, None, {}), p.score(None, None, k, v, None, {})
    assert torch.equal(s1, s2) and s1.dtype == torch.float32 and tuple(s1.shape) == (2, 4, 300)
    ko, vo = p.compress(None, None, k, v, None, {})
    assert tuple(ko.shape) == (2, 4, 150, 64)
    idx = native().topk_select(s1, 150).cpu().numpy()
    wk, wv = O.gather_kv(k.float().cpu().numpy(), v.float().cpu().numpy(), idx)
    assert np.array_equal(ko.float().cpu().numpy(), wk) and np.array_equal(vo.float().cpu().numpy(), wv)


# ---------------------------------------------------------------------------------------------
# press level: the public classes against the committed outputs of the real reference
# ---------------------------------------------------------------------------------------------
def make_press(s, ratio):
    import kvpress_amd as P

    if s["kind"] == "knorm":
        return P.KnormPress(compression_ratio=ratio)
    if s["kind"] == "snapkv":
        return P.SnapKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"])
    if s["kind"] == "pyramid":
        return P.PyramidKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"], beta=s["beta"])
    if s["kind"] == "tova":
        return P.TOVAPress(compression_ratio=ratio)
    if s["kind"] == "keydiff":
        return P.KeyDiffPress(compression_ratio=ratio)
    if s["kind"] == "cur":
        return P.CURPress(compression_ratio=ratio, num_sinks=s.get("sinks", 4), leverage_type=s["leverage"],
                          use_local_approximation=s.get("local", True), local_window_size=s.get("window", 16))
    if s["kind"] == "lagkv":
        return P.contrib.LagKVPress(compression_ratio=ratio, n_sink=s["n_sink"], lag_size=s["lag"], cross_scoring=s.get("cross", False))
    if s["kind"] == "observed":
        return P.contrib.ObservedAttentionPress(compression_ratio=ratio)
    if s["kind"] == "qfilter":
        p = P.QFilterPress(compression_ratio=ratio)
        p.q_filters = torch.from_numpy(_inputs.make_qfilters(s)).to(DEV)   # the press casts per call to the key dtype
        return p
    if s["kind"] == "streaming":
        return P.StreamingLLMPress(compression_ratio=ratio, n_sink=s["n_sink"])
    return P.ExpectedAttentionPress(compression_ratio=ratio, n_future_positions=s["n_future"], n_sink=s["n_sink"],
                                    use_covariance=s["use_covariance"], use_vnorm=s["use_vnorm"], epsilon=s["epsilon"])


@pytest.mark.parametrize("name", list(_inputs.CASES))
def test_press_fp32_vs_reference(name):
    """Module and tensors in float32 (= the reference's O32 run): scores within 1e-3, compress()
    output shape and dtype as the reference, retained set a valid top-k of the reference scores."""
    s = _inputs.make_case(name)
    g = gold(name)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32, DEV)
    if s["kind"] == "qfilter":
        att.layer_idx = _inputs.QF_LAYER
    keys, values = to_dev(s["keys"], "f32"), to_dev(s["values"], "f32")
    kwargs = {"position_embeddings": pe}
    attn = to_dev(_inputs.make_attentions(s), "f32") if s["kind"] == "observed" else None
    with torch.no_grad():
        sc = make_press(s, 0.5).score(att, hidden, keys, values, attn, kwargs)
        assert sc.dtype == torch.float32 and tuple(sc.shape) == (s["B"], s["H"], s["S"])
        got = sc.cpu().numpy()
        ref = g["scores_f32"]
        pad = slice(None)
        if s["kind"] in ("snapkv", "pyramid", "tova"):
            assert_scores_close(got[..., :-s["W"]], ref[..., :-s["W"]], RTOL, name)
            assert (got[..., -s["W"]:] > got[..., :-s["W"]].max()).all()
        elif s["kind"] == "keydiff":  # a cosine in [-1, 1] crossing zero: absolute tolerance
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5, err_msg=name)
        elif s["kind"] == "lagkv":
            _inputs.assert_lag_scores_close(got, ref, s, name)
        elif s["kind"] == "observed":
            assert_scores_close(got, ref, 2e-5, name)
        elif s["kind"] == "qfilter":  # a signed dot product crossing zero
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5, err_msg=name)
        elif s["kind"] == "streaming":
            assert np.array_equal(got, ref)
        elif s["kind"] == "cur":
            assert_scores_close(got, ref, 2e-5, name)
        elif s["kind"] == "ea":
            assert_scores_close(got[..., s["n_sink"]:], ref[..., s["n_sink"]:], RTOL, name)
            if s["n_sink"]:
                assert (got[..., : s["n_sink"]] > got[..., s["n_sink"]:].max()).all()
        else:
            assert_scores_close(got, ref, 1e-5, name)
        for i, r in enumerate(s["ratios"]):
            ko, vo = make_press(s, r).compress(att, hidden, keys, values, attn, kwargs)
            n = int(g[f"nkept_{i}"])
            assert tuple(ko.shape) == tuple(vo.shape) == (s["B"], s["H"], n, s["D"])
            assert ko.is_contiguous() and vo.is_contiguous() and ko.dtype == keys.dtype
            if s["kind"] == "streaming":  # the 0/1 scores depend on the ratio; the kept set is pinned exactly
                sc = make_press(s, r).score(att, hidden, keys, values, None, kwargs)
                ref = O.streaming_llm_score(s["B"], s["H"], s["S"], r, s["n_sink"])
                assert np.array_equal(native().topk_select(sc, n).cpu().numpy(), g[f"idx_f32_{i}"])
            idx = native().topk_select(sc, n).cpu().numpy()
            ok, msg = O.topk_is_valid(ref, idx, n, rel_band=1e-4)
            assert ok, f"{name} r={r}: {msg}"
            wk, wv = O.gather_kv(s["keys"], s["values"], idx)
            assert np.array_equal(ko.cpu().numpy(), wk) and np.array_equal(vo.cpu().numpy(), wv)
        # ratio 0 returns the very same objects (scorer_press.py:86-87)
        k0, v0 = make_press(s, 0.0).compress(att, hidden, keys, values, attn, kwargs)
        assert k0 is keys and v0 is values


@pytest.mark.parametrize("name", [n for n, c in _inputs.CASES.items() if c["dtype"] != "f32"])
def test_press_native_dtype_runs_and_overlaps_reference(name):
    """bf16 / f16 module and tensors, as in production, against the reference's own native-dtype scores (SURVEY §8c iii):
    Knorm is exact up to ties; the others agree with the reference outside a few-ulp band around its threshold and overlap
    its own top-k to within 2 % of what the float32-mode reference itself achieves."""
    s = _inputs.make_case(name)
    g = gold(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, DEV)
    if s["kind"] == "qfilter":
        att.layer_idx = _inputs.QF_LAYER
    keys, values = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    kwargs = {"position_embeddings": pe}
    attn = to_dev(_inputs.make_attentions(s), s["dtype"]) if s["kind"] == "observed" else None
    with torch.no_grad():
        sc = make_press(s, 0.5).score(att, hidden, keys, values, attn, kwargs)
        ref_nat = g["scores_nat"]
        for i, r in enumerate(s["ratios"]):
            n = int(g[f"nkept_{i}"])
            ko, vo = make_press(s, r).compress(att, hidden, keys, values, attn, kwargs)
            assert tuple(ko.shape) == (s["B"], s["H"], n, s["D"]) and ko.dtype == dt
            idx = native().topk_select(sc, n).cpu().numpy()
            if s["kind"] == "knorm":
                # fp32 norms round monotonically to the reference's bf16 scores: a valid tie-broken top-k
                ok, msg = O.topk_is_valid(ref_nat, idx, n)
                assert ok, msg
            else:
                # dtype-faithful rule (SURVEY §8c iii): the reference rounds to the model dtype after every op, so its own score of a
                # position is defined to a few units in the last place only.  Outside that band around ITS threshold the
                # float32-selected set must agree with it; inside, either choice is a valid top-k of the reference's scores.
                ulps = NATIVE_ULPS.get(s["kind"])
                R = ref_nat.reshape(-1, ref_nat.shape[-1]).astype(np.float64)
                ref_idx = O.topk_select(ref_nat, n)
                inter = 1.0
                for sc_row, mine, theirs in zip(R, idx.reshape(-1, n), ref_idx.reshape(-1, n)):
                    if n == 0 or n == sc_row.size:
                        continue
                    inter = min(inter, len(np.intersect1d(mine, theirs)) / n)
                    if ulps is None:
                        continue
                    t = np.sort(sc_row)[::-1][n - 1]
                    band = ulps * (2.0 ** -8 if s["dtype"] == "bf16" else 2.0 ** -11) * abs(t)
                    kept = np.zeros(sc_row.shape, bool)
                    kept[mine] = True
                    assert not (~kept & (sc_row > t + band)).any(), f"{name} r={r}: dropped a position more than {ulps} ulps above the reference's threshold"
                    assert not (kept & (sc_row < t - band)).any(), f"{name} r={r}: kept a position more than {ulps} ulps below the reference's threshold"
                floor = NATIVE_OVERLAP.get(name, 0.0) - 0.02
                assert inter >= floor, f"{name} r={r}: overlap with the bf16 reference's own top-k {inter:.3f} < {floor:.3f}"


# Band (units in the last place of the model dtype, relative to the threshold) inside which the native-dtype reference's scores
# are not defined more precisely than its own rounding chain; measured on the fixtures as the largest distance from the
# reference's threshold at which the float32-mode reference ("O32") and the native-dtype reference disagree (snapkv <= 8.4,
# ea <= 3.5, tova <= 16.8, cur <= 3.0, observed <= 1.9), with a margin.  KeyDiff / QFilter / LagKV scores are differences of
# nearly equal numbers (cosine similarity, filter projections, ranks): their reference values are dominated by rounding, so only
# the overlap floor applies there.
NATIVE_ULPS = {"snapkv": 12, "pyramid": 12, "ea": 6, "tova": 24, "cur": 6, "observed": 4}
# smallest overlap (over the case's ratios and rows) between the float32-mode reference's top-k and the native-dtype reference's
# own, measured when the fixtures were made; the kernels select like the float32-mode reference, the test allows 0.02 below.
NATIVE_OVERLAP = {
    "sk_257_A": 0.745, "sk_257_B": 0.745, "sk_4096": 0.993, "sk_f16_d64": 0.995, "sk_ks1": 0.915, "sk_h512_bf16": 0.915,
    "sk_h1024_f16": 0.995, "ea_257_A": 0.961, "ea_1500_B": 0.993, "ea_nocov": 0.992, "ea_novnorm_eps": 0.99, "ea_eps_sink0": 0.973,
    "ea_6000_B": 0.997, "kd_bf16_A": 0.995, "kd_bf16_B": 0.995, "kd_f16_d64": 0.997, "kd_d96_bf16": 0.99, "cur_bf16_B": 0.995,
    "cur_key_nolocal": 0.995, "cur_kvavg_w7": 0.99, "tv_257": 0.98, "tv_4096_B": 0.996, "py_bf16_l7": 0.99, "qf_bf16_B": 0.995,
    "qf_f16_d64": 0.995, "lag_bf16": 0.959, "lag_cross_f16": 0.716, "oa_bf16": 0.995, "oa_f16_g1": 1.0,
}


def assert_scores_within_query_rounding(got, want, q_got, q_want, keys, W, name):
    """the library's window projection sums its partial products in another order than the GEMM library, so a few
    window queries land on the neighbouring 16-bit value.  Instead of a flat tolerance, the bound that follows from the queries at hand:
    if no logit q.k / sqrt(D) moves by more than delta, every softmax entry -- numerator and normaliser -- moves by a factor within
    e^(+-2 delta), and SnapKV's scores (means and 5-tap averages of softmax entries, all positive) inherit that factor.  Also pins how
    different the two query tensors may be at all: <= 2 % of the elements, each by <= 1 ulp of its RoPE pair's magnitude (+ the pair's
    partner: 3 ulp of the row scale, as test_qproj_rope_kernel_vs_torch)."""
    import math

    qa, qb = q_got.float(), q_want.float()
    dt_ulp = 2.0 ** -7 if q_got.dtype == torch.bfloat16 else 2.0 ** -10
    frac = (qa != qb).float().mean().item()
    assert frac <= 0.02, f"{name}: {frac:.2%} of the window queries differ"
    assert ((qa - qb).abs() <= 3.0 * dt_ulp * qb.abs().amax(dim=-1, keepdim=True)).all(), f"{name}: a window query differs by more than rounding"
    G = qa.shape[1] // keys.shape[1]
    k = keys.float().repeat_interleave(G, dim=1)
    delta = (torch.matmul(qa - qb, k.transpose(2, 3)).abs().max() / math.sqrt(qa.shape[-1])).item()
    bound = math.expm1(2.0 * delta) + 1e-5
    g, w = got.float()[..., :-W], want.float()[..., :-W]
    err = ((g - w).abs() / w.abs().clamp_min(1e-30)).max().item()
    assert err <= bound, f"{name}: scores differ by {err:.3e}, the queries' rounding explains at most {bound:.3e} (largest logit shift {delta:.3e})"
    return err, bound


@pytest.mark.parametrize("name", SK)
def test_snapkv_fused_rope_is_bit_identical_to_torch_rope(name):
    """kvp_snapkv_score_rope (RoPE inside the library, torch's per-op rounding reproduced) must give exactly
    the scores of kvp_snapkv_score fed with torch's own q*cos + rotate_half(q)*sin (snapkv_press.py:56-58)."""
    import kvpress_amd as P
    from kvpress_amd.utils import get_prerope_query_states

    s = _inputs.make_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, DEV)
    keys = to_dev(s["keys"], s["dtype"])
    W = s["W"]
    from kvpress_amd.presses.snapkv_press import _rotate_half

    with torch.no_grad():
        q_pre = get_prerope_query_states(att, hidden[:, -W:])        # ONE q_proj GEMM feeds both paths
        cos, sin = pe[0][:, -W:], pe[1][:, -W:]
        q_win = (q_pre * cos.unsqueeze(1)) + (_rotate_half(q_pre) * sin.unsqueeze(1))  # torch's RoPE
        a = native().snapkv_score(q_win, keys, s["ks"])
        b = native().snapkv_score_rope(q_pre, cos, sin, keys, s["ks"])
        c = P.SnapKVPress(0.5, window_size=W, kernel_size=s["ks"]).score(att, hidden, keys, None, None, {"position_embeddings": pe})
    if s["dtype"] == "f32":
        # float32 has no rounding step to reproduce; torch's own evaluation order may differ in the last bit
        assert torch.allclose(a, b, rtol=2e-6, atol=0), f"max abs diff {(a - b).abs().max().item():.3e}"
    else:  # f16 and bf16 (the production dtype): bit-identical
        assert torch.equal(a, b), f"max abs diff {(a - b).abs().max().item():.3e}"
    # the press projects the window itself: with the model's q_proj (a library GEMM that need not be run-to-run bit-stable)
    # or, for a plain bf16 / f16 nn.Linear with W = 64 and D = 128, in the library's own kernel, whose fp32 summation order
    # differs from the GEMM library's (a few queries round to the neighbouring 16-bit value)
    if native().qproj_rope_supported(att, hidden, W):
        with torch.no_grad():
            q_lib = native().snapkv_qproj_rope(hidden[:, -W:], att.q_proj.weight, cos, sin)
        assert_scores_within_query_rounding(c, b, q_lib, q_win, keys, W, name)
    else:
        assert torch.allclose(b, c, rtol=1e-5, atol=0)


def test_ea_logits_triangular_form_vs_oracle():
    """ea_logits_mfma_tri_kernel (k^T C k on the doubled upper triangle of C, 40 instead of 64 MFMAs per tile and wave) against the
    oracle: symmetric covariances as the press produces them, an ASYMMETRIC matrix (U_jc = C_jc + C_cj is exact for any C), ragged
    lengths, one tile and many, both 16-bit dtypes; and the mean-only form (use_covariance=False: the full-form kernel without its
    covariance chains)."""
    N = native()
    rs = np.random.RandomState(12)
    for dtype in ("bf16", "f16"):
        for B, Hq, Hkv, S, n_sink, sym in ((1, 8, 2, 4500, 4, True), (2, 4, 4, 130, 0, True), (1, 4, 1, 9000, 3, False), (1, 32, 8, 12345, 4, True)):
            kn = _inputs.round_to((rs.standard_normal((B, Hkv, S, 128)) * 0.7).astype(np.float32), dtype)
            vn = _inputs.round_to(rs.standard_normal((B, Hkv, S, 128)).astype(np.float32), dtype)
            mu = (rs.standard_normal((B, Hq, 128)) * 0.4).astype(np.float32)
            a = (rs.standard_normal((B, Hq, 128, 128)) * 0.04).astype(np.float32)
            cov = (a @ a.transpose(0, 1, 3, 2)) if sym else (a @ a.transpose(0, 1, 3, 2) + 0.01 * rs.standard_normal((B, Hq, 128, 128)).astype(np.float32))
            k, v = to_dev(kn, dtype), to_dev(vn, dtype)
            want = O.ea_score(kn, vn, mu, cov, n_sink, True, 0.0)
            got = N.ea_score(k, v, torch.from_numpy(mu).to(DEV), torch.from_numpy(cov).to(DEV), n_sink, True, 0.0).cpu().numpy()
            rel = np.abs(got[..., n_sink:] - want[..., n_sink:]) / np.abs(want[..., n_sink:])
            assert rel.max() <= 1e-3, (dtype, S, sym, rel.max())
            want0 = O.ea_score(kn, vn, mu, None, n_sink, True, 0.0)
            got0 = N.ea_score(k, v, torch.from_numpy(mu).to(DEV), None, n_sink, True, 0.0).cpu().numpy()
            rel0 = np.abs(got0[..., n_sink:] - want0[..., n_sink:]) / np.abs(want0[..., n_sink:])
            assert rel0.max() <= 1e-3, (dtype, S, "mean only", rel0.max())


@pytest.mark.parametrize("D", [64, 96, 256])
def test_ea_logits_small_heads_vs_oracle(D):
    """Round 6: head sizes 64 (Llama-3.2-1B, Qwen2-0.5B) and 96 (Phi-3-mini) on the matrix cores (ea_logits_mfma_small_kernel: every wave
    takes one 32-key sub-tile with all two / three strips of the doubled upper triangle) instead of the scalar generic kernel -- and 256
    (Gemma), which the generic kernel REFUSED until round 6 (its 66 KiB tile is above a launch's default LDS limit): symmetric
    and ASYMMETRIC covariances, GQA groups 1 .. 7, ragged lengths from one partial tile to several chunks, sinks, both 16-bit dtypes, a
    [B, S, H, D] K buffer seen as [B, H, S, D], the mean-only form, a batch of two."""
    N = native()
    rs = np.random.RandomState(600 + D)
    for dtype in ("bf16", "f16"):
        for B, Hq, Hkv, S, n_sink, sym, bshd in ((1, 8, 2, 4500, 4, True, False), (2, 4, 4, 70, 0, True, True), (1, 7, 1, 9000, 3, False, False),
                                                 (1, 32, 8, 12345, 4, True, True), (2, 6, 2, 333, 1, False, False), (1, 3, 3, 197, 7, True, False)):
            kn = _inputs.round_to((rs.standard_normal((B, Hkv, S, D)) * 0.8).astype(np.float32), dtype)
            vn = _inputs.round_to(rs.standard_normal((B, Hkv, S, D)).astype(np.float32), dtype)
            mu = (rs.standard_normal((B, Hq, D)) * 0.4).astype(np.float32)
            a = (rs.standard_normal((B, Hq, D, D)) * 0.06).astype(np.float32)
            cov = a @ a.transpose(0, 1, 3, 2)
            if not sym:
                cov = cov + 0.01 * rs.standard_normal((B, Hq, D, D)).astype(np.float32)
            k = to_dev(np.ascontiguousarray(kn.transpose(0, 2, 1, 3)), dtype).transpose(1, 2) if bshd else to_dev(kn, dtype)
            v = to_dev(vn, dtype)
            want = O.ea_score(kn, vn, mu, cov, n_sink, True, 0.0)
            got = N.ea_score(k, v, torch.from_numpy(mu).to(DEV), torch.from_numpy(cov).to(DEV), n_sink, True, 0.0).cpu().numpy()
            rel = np.abs(got[..., n_sink:] - want[..., n_sink:]) / np.abs(want[..., n_sink:])
            assert rel.max() <= 1e-4, (dtype, D, S, sym, rel.max())   # (measured ~1e-6: the contract is 1e-3)
            if n_sink:
                assert np.all(got[..., :n_sink] == np.float32(got[..., n_sink:].max()) + np.float32(1.0))
            want0 = O.ea_score(kn, vn, mu, None, n_sink, False, 0.01)
            got0 = N.ea_score(k, v, torch.from_numpy(mu).to(DEV), None, n_sink, False, 0.01).cpu().numpy()
            rel0 = np.abs(got0[..., n_sink:] - want0[..., n_sink:]) / np.abs(want0[..., n_sink:])
            assert rel0.max() <= 1e-4, (dtype, D, S, "mean only", rel0.max())


@pytest.mark.parametrize("D", [64, 96, 256])
def test_ea_qstats_narrow_heads_on_the_matrix_cores(D):
    """Round 6: the statistics of 64- and 96-dimensional heads on the 128-wide syrk.  D = 64 with the queries as q_proj leaves them ([B, S, Hq * 64]:
    heads 64 elements apart, an even number of them) runs as PAIRS of heads (the diagonal blocks of a pair's second moments are the two heads'
    own); D = 96, an odd head count and a contiguous [B, Hq, S, D] tensor run as heads of 128 dimensions whose upper ones are zero in LDS
    (the lanes of the missing chunks request nothing); D = 256 (Gemma) as the SIX pairs of 64-dimension quarters of every head, scattered into
    the 256 x 256 covariance by the combine.  Large means, dominant channels, a ragged tail, both dtypes."""
    rs = np.random.RandomState(D)
    N = native()
    for dtype, (B, Hq, Sq) in (("bf16", (1, 6, 10000)), ("f16", (2, 2, 4500)), ("bf16", (1, 3, 5000))):
        q = rs.standard_normal((B, Hq, Sq, D)).astype(np.float32) * np.exp(0.5 * rs.standard_normal((1, Hq, 1, D))).astype(np.float32)
        q += rs.standard_normal((1, Hq, 1, D)).astype(np.float32) * 2.0
        q[:, :, :, ::13] *= 5.0
        q = _inputs.round_to(q, dtype)
        mu_w, cov_w = O.ea_query_stats(q, True)
        d = np.sqrt(np.einsum("bhii->bhi", cov_w))
        layouts = {"q_proj": to_dev(np.ascontiguousarray(q.transpose(0, 2, 1, 3)), dtype).transpose(1, 2), "contiguous": to_dev(q, dtype)}
        for lname, qt in layouts.items():
            mu, cov = N.ea_qstats(qt, True)
            assert np.abs(mu.cpu().numpy() - mu_w).max() <= 1e-5 * np.abs(mu_w).max() + 1e-6, (dtype, lname)
            err = np.abs(cov.cpu().numpy() - cov_w) / (d[..., :, None] * d[..., None, :])
            assert err.max() <= 1e-3 and err.mean() <= 1e-4, (dtype, lname, Hq, err.max(), err.mean())
            mu2, cov2 = N.ea_qstats(qt, False)
            assert cov2 is None and np.abs(mu2.cpu().numpy() - mu_w).max() <= 1e-5 * np.abs(mu_w).max() + 1e-6


def test_ea_fused_finalize_equals_three_kernels(knobs):
    """kvp_ea_score's one-pass ||v|| + row normalisers + finalize (ea_vnorm_finalize_kernel: 256-byte rows, >= 4096 scored keys) against
    the three kernels it replaces (KVP_EA_FUSED_FINALIZE=0: the generic path other shapes take): the same bits, with and without sinks, GQA
    groups of 1 / 2 / 4 / 8 / 16 (one pass; 8 and 16 since round 6, in rounds of 2 / 1 keys per 16 lanes) and 3 (the three kernels), ragged lengths, a strided V view."""
    N = native()
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    for dt in (torch.bfloat16, torch.float16):
        for B, Hq, Hkv, S, n_sink in ((1, 8, 2, 6000, 4), (2, 4, 4, 4100, 0), (1, 16, 2, 9001, 7), (1, 32, 8, 20000, 4), (1, 4, 2, 5003, 1), (1, 32, 2, 5000, 4), (2, 16, 1, 4200, 0),
                                      (1, 6, 2, 4500, 4)):
            k = (torch.randn((B, Hkv, S, 128), generator=g, device=DEV) * 0.5).to(dt)
            vfull = torch.randn((B, S, Hkv, 128), generator=g, device=DEV).to(dt)
            v = vfull.transpose(1, 2)                                   # [B, Hkv, S, 128] view of a [B, S, Hkv, 128] buffer
            mu = torch.randn((B, Hq, 128), generator=g, device=DEV) * 0.3
            a = torch.randn((B, Hq, 128, 128), generator=g, device=DEV) * 0.05
            cov = a @ a.transpose(-1, -2)
            knobs(KVP_EA_FUSED_FINALIZE=0)
            ref = N.ea_score(k, v, mu, cov, n_sink, True, 0.02)
            knobs(KVP_EA_FUSED_FINALIZE=None)
            got = N.ea_score(k, v, mu, cov, n_sink, True, 0.02)
            assert torch.equal(got, ref), (dt, B, Hq, Hkv, S, n_sink, float((got - ref).abs().max()))


def test_ea_qstats_mfma_multichunk():
    """bf16, D=128, 10 000 rows (3 chunks + ragged tail), non-zero mean and a few dominant channels:
    exercises the syrk on the matrix cores (transposed LDS reads) and the pairwise combine."""
    rs = np.random.RandomState(7)
    B, Hq, Sq, D = 1, 3, 10000, 128
    q = rs.standard_normal((B, Hq, Sq, D)).astype(np.float32) * np.exp(0.5 * rs.standard_normal((1, Hq, 1, D))).astype(np.float32)
    q += rs.standard_normal((1, Hq, 1, D)).astype(np.float32) * 3.0     # means up to ~10 sigma
    q[:, :, :, ::17] *= 6.0
    q = _inputs.round_to(q, "bf16")
    mu_w, cov_w = O.ea_query_stats(q, True)
    qt = to_dev(np.ascontiguousarray(q.transpose(0, 2, 1, 3)), "bf16").transpose(1, 2)  # q_proj layout [B,S,Hq*D]
    mu, cov = native().ea_qstats(qt, True)
    assert np.abs(mu.cpu().numpy() - mu_w).max() <= 1e-5 * np.abs(mu_w).max() + 1e-6
    d = np.sqrt(np.einsum("bhii->bhi", cov_w))
    err = np.abs(cov.cpu().numpy() - cov_w) / (d[..., :, None] * d[..., None, :])
    # exact products of the bf16 inputs, raw moments accumulated in fp32 per chunk: means up to ~10 sigma cost ~1e-4
    assert err.max() <= 1e-3 and err.mean() <= 1e-4, (err.max(), err.mean())
    mu2, cov2 = native().ea_qstats(qt, False)
    assert cov2 is None and torch.equal(mu, mu2)


def test_ea_full_chain_mfma_vs_oracle():
    """bf16, S=6000 (>= 4096: query statistics AND logits on the matrix cores), structured data with large
    query means: the whole kernel chain against the float64 oracle fed with the same bf16 queries."""
    s = _inputs.make_case("ea_6000_B")
    q = _ea_q(s)                                                       # pre-RoPE queries, rounded to bf16
    mu_w, cov_w = O.ea_query_stats(q, True)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
    pos = torch.arange(s["S"], s["S"] + s["n_future"])[None]
    c, si = rot(torch.zeros(1), pos)
    mu_w, cov_w = O.ea_avg_rope(mu_w, cov_w, c[0].numpy(), si[0].numpy())
    want = O.ea_score(s["keys"], s["values"], mu_w, cov_w, s["n_sink"], True, 0.0)
    # kernel chain
    import kvpress_amd as P
    qt = to_dev(np.ascontiguousarray(q.transpose(0, 2, 1, 3)), "bf16").transpose(1, 2)
    mu, cov = native().ea_qstats(qt, True)
    att_d, rot_d, _, _ = _inputs.build_llama_attention(s, torch.bfloat16, DEV)
    press = P.ExpectedAttentionPress(0.5)
    mu, cov = press.apply_avg_rope(att_d, mu, cov, s["S"])
    got = native().ea_score(to_dev(s["keys"], "bf16"), to_dev(s["values"], "bf16"), mu, cov, s["n_sink"], True, 0.0).cpu().numpy()
    ns = s["n_sink"]
    assert_scores_close(got[..., ns:], want[..., ns:], RTOL, "ea_6000_B chain")


# ---------------------------------------------------------------------------------------------
# fused compress entry points == the modular sequence, bit for bit
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", KN)
def test_fused_knorm_compress_equals_modular(name):
    s = _inputs.make_case(name)
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    N = native()
    sc = N.rownorm_score(k, -1.0)
    for n in sorted({0, 1, s["S"] // 3, s["S"] // 2, s["S"] - 1, s["S"]}):
        ko, vo = N.knorm_compress(k, v, n)
        wk, wv = N.gather_kv(k, v, N.topk_select(sc, n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"{name} n={n}"
    # strided views (every second token) and a second data set through the same cached workspace
    k2, v2 = k[:, :, ::2], v[:, :, ::2]
    n = k2.shape[2] // 2
    ko, vo = N.knorm_compress(k2, v2, n)
    wk, wv = N.gather_kv(k2, v2, N.topk_select(N.rownorm_score(k2, -1.0), n))
    assert torch.equal(ko, wk) and torch.equal(vo, wv)


@pytest.mark.parametrize("variant", ["cluster_knorm", "passes"])
@pytest.mark.parametrize("S", [16384, 16385, 20000, 32768, 32769, 70001])
def test_fused_knorm_select_variants_equal_modular(S, variant, knobs):
    """Knorm's fused compress across the select's row-length regimes: own digits <= 16384; beyond, norms + select in ONE cluster
    launch (default) or -- what a device that cannot hold the cluster grid runs, KVP_TK_CLUSTER=0 -- the norm kernel with its fused
    first-digit histogram + the (chunk, row) passes; heavy ties included (bf16 norms): the modular sequence's bytes, twice through the
    same self-cleaning workspace."""
    if variant != "cluster_knorm" and S <= 16384:
        pytest.skip("the variants differ only for rows beyond 16384 scores")
    knobs(**{"cluster_knorm": {}, "passes": dict(KVP_TK_CLUSTER=0)}[variant])
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(S)
    k = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    k[:, :, ::7] = k[:, :, 3:4]                       # many exactly equal norms
    sc = N.rownorm_score(k, -1.0)
    for n in sorted({1, S // 2, (3 * S) // 4, S - 1}) * 2:
        ko, vo = N.knorm_compress(k, v, n)
        wk, wv = N.gather_kv(k, v, N.topk_select(sc, n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"S={S} n={n}"
    if S > 16384:   # two batch elements x four heads (row = b * H + h), float16, a strided view of the cache
        k2 = torch.randn((2, 4, S + 5, 128), generator=g, device=DEV).to(torch.float16)[:, :, 5:]
        v2 = torch.randn((2, 4, S + 5, 128), generator=g, device=DEV).to(torch.float16)[:, :, 5:]
        ko, vo = N.knorm_compress(k2, v2, S // 2)
        wk, wv = N.gather_kv(k2, v2, N.topk_select(N.rownorm_score(k2, -1.0), S // 2))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"S={S} (B=2, f16, view)"


@pytest.mark.parametrize("name", SK + [n for n, c in _inputs.CASES.items() if c["kind"] == "tova"])
def test_fused_snapkv_compress_equals_modular(name):
    s = _inputs.make_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, dt, DEV)
    from kvpress_amd.utils import get_prerope_query_states

    W, S = s["W"], s["S"]
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    N = native()
    with torch.no_grad():
        q_pre = get_prerope_query_states(att, hidden[:, -W:])
    c, si = cos[:, -W:], sin[:, -W:]
    sc = N.snapkv_score_rope(q_pre, c, si, k, s["ks"])
    # below / at / above the window size, generic sizes, everything kept
    for n in sorted({1, max(1, W - 1), W, min(S, W + 1), S // 2, (2 * S) // 3, S - 1, S}):
        ko, vo = N.snapkv_compress_rope(q_pre, c, si, k, v, s["ks"], n)
        wk, wv = N.gather_kv(k, v, N.topk_select(sc, n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"{name} n={n}"


@pytest.mark.parametrize("variant", ["default", "passes"])
@pytest.mark.parametrize("S", [70, 1087, 1089, 1500, 2112, 2113, 3000, 4160, 4161, 9000, 16448, 16449, 20001, 32832, 32833, 40000, 70003])
def test_fused_snapkv_select_variants_equal_modular(S, variant, knobs):
    """The fused compress picks its select by row length (pool + select in one launch up to 4096 columns, one-launch select
    up to 16384; beyond: pooling inside the cluster select's loader (default) or -- a device that cannot hold the cluster grid,
    KVP_TK_CLUSTER=0 -- the pooling kernel with its fused first-digit histogram + the (chunk, row) passes): always the modular
    sequence's bytes."""
    if variant != "default" and S - 64 <= 16384:
        pytest.skip("the variants differ only for rows beyond 16384 columns")
    knobs(**{"default": {}, "passes": dict(KVP_TK_CLUSTER=0)}[variant])
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(S)
    k = torch.randn((2, 2, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((2, 2, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    q = torch.randn((2, 8, 64, 128), generator=g, device=DEV).to(torch.bfloat16)
    ang = torch.rand((1, 64, 128), generator=g, device=DEV)
    c, si = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    sc = N.snapkv_score_rope(q, c, si, k, 5)
    for n in sorted({64, 65, S // 2, (3 * S) // 4, S - 1}) * 2:
        ko, vo = N.snapkv_compress_rope(q, c, si, k, v, 5, n)
        wk, wv = N.gather_kv(k, v, N.topk_select(sc, n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"S={S} n={n}"


@pytest.mark.parametrize("S", [70, 1500, 4160, 4161, 9000, 16448, 16449, 40000, 131072])
def test_fused_compress_in_score_order_equals_modular(S):
    """flags | KVP_ORDER_SCORE on the fused compress calls: K' / V' rows in descending score order (the reference's layout: window tokens
    first, ties by position).  The sort takes its keys from wherever the fused call left them -- the pooled scores, or (long rows, and
    short ones pooled inside the select) the un-pooled column sums, pooled again per kept position with the loader's arithmetic -- so
    the rows must be the bytes of the modular sequence score -> kvp_topk_select(KVP_ORDER_SCORE) -> gather, for every select regime,
    both kernel sizes' paths, Knorm included."""
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(S + 1)
    k = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    q = torch.randn((1, 32, 64, 128), generator=g, device=DEV).to(torch.bfloat16)
    ang = torch.rand((1, 64, 128), generator=g, device=DEV)
    c, si = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    for ks in (5, 3):
        sc = N.snapkv_score_rope(q, c, si, k, ks)
        for n in sorted({64, 65, S // 2, S - 1}):
            if n > S:
                continue
            ko, vo = N.snapkv_compress_rope(q, c, si, k, v, ks, n, N.ORDER_SCORE)
            idx = N.topk_select(sc, n, N.ORDER_SCORE)
            wk, wv = N.gather_kv(k, v, idx)
            assert torch.equal(ko, wk) and torch.equal(vo, wv), f"snapkv S={S} ks={ks} n={n}"
            if n >= 64:   # the window tokens lead, in position order
                assert torch.equal(idx[..., :64], torch.arange(S - 64, S, device=DEV, dtype=torch.int32).expand(1, 8, 64))
    sn = N.rownorm_score(k, -1.0)
    for n in sorted({1, S // 2, S - 1}):
        ko, vo = N.knorm_compress(k, v, n, N.ORDER_SCORE)
        wk, wv = N.gather_kv(k, v, N.topk_select(sn, n, N.ORDER_SCORE))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"knorm S={S} n={n}"


@pytest.mark.parametrize("S", [32833, 70003, 131072])
@pytest.mark.parametrize("data", ["flat", "tied", "constant"])
def test_fused_snapkv_hist1_cluster_select(S, data):
    """the cluster select's HIST1 form (first digit accumulated by the kernel that wrote the scores, three rounds with counter
    barriers) is what the fused SnapKV compress takes for kernel_size != 5 on rows beyond 16384 columns (the kernel_size-5 path pools
    inside the select's own loader).  Driven here on flat keys, keys with many exact duplicates (tied column sums) and CONSTANT keys
    (every score of a row equal: the whole selection is decided by the tie rule), twice through the same self-cleaning workspace and
    for both orders: the bytes of the modular sequence score -> kvp_topk_select -> gather."""
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(S + len(data))
    k = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    q = torch.randn((1, 32, 64, 128), generator=g, device=DEV).to(torch.bfloat16)
    if data == "tied":
        k[:, :, ::3] = k[:, :, 5:6]
    elif data == "constant":
        k[:] = k[:, :, 7:8]
    else:
        k.mul_(0.05)
    ang = torch.rand((1, 64, 128), generator=g, device=DEV)
    c, si = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    for ks in (3, 7):
        sc = N.snapkv_score_rope(q, c, si, k, ks)
        if data == "constant":   # every column sum of a head is the same value up to the last bits (the steady-state loop and the walk's
            body = sc[0, :, 8:S - 64 - 8]   # last tiles sum in different orders): the selection is decided by near-ties and the tie rule
            assert float((body.max(-1).values / body.min(-1).values).max()) < 1 + 1e-5
        for n in sorted({65, S // 2, S - 1}) * 2:
            ko, vo = N.snapkv_compress_rope(q, c, si, k, v, ks, n)
            wk, wv = N.gather_kv(k, v, N.topk_select(sc, n))
            assert torch.equal(ko, wk) and torch.equal(vo, wv), f"S={S} {data} ks={ks} n={n}"
        ko, vo = N.snapkv_compress_rope(q, c, si, k, v, ks, S // 2, N.ORDER_SCORE)
        wk, wv = N.gather_kv(k, v, N.topk_select(sc, S // 2, N.ORDER_SCORE))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), f"S={S} {data} ks={ks} score order"
    N.async_error_check()


def test_snapkv_scores_are_run_to_run_identical_with_balanced_pass2(knobs):
    """Round 6: pass 2 takes its tile ranges from pass 1's measured workgroup times (snapkv_internal.h: snapkv_p2_shares_plan), i.e. from a
    quantity that differs from run to run.  Its column sums are computed per tile by one workgroup in a fixed order, so the SCORES must
    not: 20 consecutive runs at the BASELINE shape are bit-identical, equal to the static interleaved walk (KVP_SK_BALANCE=0), and the
    same holds for a ragged length (where the walk's last tiles take the plain path) and for a batch of two."""
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(606)
    for B, S, Hkv in ((1, 131072, 8), (1, 131072 - 1000 + 37, 8), (2, 70003, 8), (1, 60000, 2)):   # (the last: G = 16, four group-blocks per kv-head)
        k = torch.randn((B, Hkv, S, 128), generator=g, device=DEV).to(torch.bfloat16)
        q = (torch.randn((B, 32, 64, 128), generator=g, device=DEV) * 1.2).to(torch.bfloat16)
        knobs(KVP_SK_BALANCE=None)
        ref = N.snapkv_score(q, k, 5)
        for i in range(20 if S == 131072 else 6):
            if i % 3 == 0:   # something else in between: the clocks (and with them pass 1's times) move
                torch.mm(torch.randn((2048, 2048), device=DEV), torch.randn((2048, 2048), device=DEV))
            assert torch.equal(N.snapkv_score(q, k, 5), ref), f"B={B} S={S}: run {i} differs"
        knobs(KVP_SK_BALANCE=0)
        static = N.snapkv_score(q, k, 5)
        if S % 128 == 0:
            assert torch.equal(static, ref), f"B={B} S={S}: balanced != static walk"
        else:   # a ragged tail hands other tiles to the plain path (different summation order inside a tile): last-bit differences only
            assert float(((static - ref).abs() / ref.abs().clamp_min(1e-30)).max()) < 2e-6
    knobs(KVP_SK_BALANCE=None)


def test_snapkv_single_row_rotary_table_broadcasts():
    """A decoding step hands over the rotary table of the current position only; the reference's ``cos[:, -W:]`` then
    broadcasts that row over the whole window.  Same here (stride 0), identical to an explicitly repeated table."""
    N = native()
    g = torch.Generator(device=DEV); g.manual_seed(5)
    k = torch.randn((1, 2, 300, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 2, 300, 128), generator=g, device=DEV).to(torch.bfloat16)
    q = torch.randn((1, 8, 64, 128), generator=g, device=DEV).to(torch.bfloat16)
    ang = torch.rand((1, 1, 128), generator=g, device=DEV)
    c1, s1 = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    cw, sw = c1.repeat(1, 64, 1), s1.repeat(1, 64, 1)
    assert torch.equal(N.snapkv_score_rope(q, c1, s1, k, 5), N.snapkv_score_rope(q, cw, sw, k, 5))
    a, b = N.snapkv_compress_rope(q, c1, s1, k, v, 5, 150), N.snapkv_compress_rope(q, cw, sw, k, v, 5, 150)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_fused_compress_without_clean_flag():
    """flags = 0: the library zeroes the histogram region itself, whatever the workspace holds."""
    import ctypes

    N = native()
    L = N.lib()
    k = torch.randn(2, 4, 3000, 128, device=DEV, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    n = 1234
    nws = L.kvp_knorm_compress_workspace_bytes(2, 4, 3000, n)
    ws = torch.full((nws,), 0xAB, dtype=torch.uint8, device=DEV)  # garbage
    ko, vo = torch.empty(2, 4, n, 128, device=DEV, dtype=torch.bfloat16), torch.empty(2, 4, n, 128, device=DEV, dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for _ in range(2):  # the second call reuses the (now clean) workspace, still with flags = 0
        rc = L.kvp_knorm_compress(P(k), k.stride(0), k.stride(1), k.stride(2), P(v), v.stride(0), v.stride(1), v.stride(2), 2, 2, 4, 3000,
                                  128, n, P(ko), P(vo), P(ws), nws, 0, st)
        assert rc == 0, L.kvp_last_error()
        wk, wv = N.gather_kv(k, v, N.topk_select(N.rownorm_score(k, -1.0), n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv)
    # too small a workspace is an error, not a crash
    rc = L.kvp_knorm_compress(P(k), k.stride(0), k.stride(1), k.stride(2), P(v), v.stride(0), v.stride(1), v.stride(2), 2, 2, 4, 3000, 128,
                              n, P(ko), P(vo), P(ws), 1024, 0, st)
    assert rc == -4 or rc != 0


# ---------------------------------------------------------------------------------------------
# window q_proj + RoPE inside the library
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sk_h512_bf16", "sk_h1024_f16"])
@pytest.mark.parametrize("dtname", ["bf16", "f16"])
def test_qproj_rope_kernel_vs_torch(name, dtname):
    """kvp_snapkv_qproj_rope against the model's own q_proj followed by torch's RoPE in the same dtype: the fp32
    accumulation order differs from the GEMM library's, so a projected value may round to the neighbouring 16-bit number
    (rarely); everything after the projection is bit-identical arithmetic."""
    from kvpress_amd.utils import get_prerope_query_states

    s = _inputs.make_case(name)
    dt = _inputs.torch_dtype(dtname)
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, dt, DEV)
    W = s["W"]
    N = native()
    assert N.qproj_rope_eligible(att, hidden, W)
    assert N.qproj_rope_supported(att, hidden, W) == (hidden.shape[0] <= 2)  # on by default since round 4; up to two batch elements (round 6)
    with torch.no_grad():
        got = N.snapkv_qproj_rope(hidden[:, -W:], att.q_proj.weight, cos[:, -W:], sin[:, -W:])
        q = get_prerope_query_states(att, hidden[:, -W:])
        c, si = cos[:, -W:].unsqueeze(1), sin[:, -W:].unsqueeze(1)
        half = q.shape[-1] // 2
        want = (q * c) + (torch.cat((-q[..., half:], q[..., :half]), dim=-1) * si)
    assert got.shape == want.shape and got.dtype == dt and got.is_contiguous()
    g, w_ = got.float(), want.float()
    ulp = 2.0 ** (-7 if dtname == "bf16" else -10)
    frac = (g != w_).float().mean().item()
    assert frac < 0.02, f"{frac:.3%} of the window queries differ"
    scale = w_.abs().amax(dim=-1, keepdim=True)  # a 1-ulp flip of q moves both outputs of its RoPE pair
    assert ((g - w_).abs() <= 3.0 * ulp * scale).all()
    # exact against an fp64 restatement rounded once: |error| <= 1 ulp of the projected value's magnitude
    q64 = (hidden[:, -W:].double() @ att.q_proj.weight.double().T).view(s["B"], W, s["Hq"], s["D"]).transpose(1, 2)
    want64 = q64 * c.double() + torch.cat((-q64[..., half:], q64[..., :half]), dim=-1) * si.double()
    assert ((g.double() - want64).abs() <= 4.0 * ulp * want64.abs().amax(dim=-1, keepdim=True) + 1e-6).all()


def test_qproj_rope_not_taken_for_hooked_or_subclassed_projections():
    """the library projection reads q_proj.weight instead of calling q_proj, so a projection with forward (pre-)hooks, an
    accelerate `_hf_hook`, or a weight that is a tensor SUBCLASS (quantised / sharded wrappers keep the nn.Linear type) must keep the
    model's own call -- and the press still works through it."""
    import kvpress_amd as P

    s = _inputs.make_case("sk_h512_bf16")
    dt = _inputs.torch_dtype("bf16")
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, DEV)
    N = native()
    W = s["W"]
    assert N.qproj_rope_eligible(att, hidden, W)
    h = att.q_proj.register_forward_hook(lambda m, i, o: o)
    assert not N.qproj_rope_eligible(att, hidden, W)
    h.remove()
    assert N.qproj_rope_eligible(att, hidden, W)
    h = att.q_proj.register_forward_pre_hook(lambda m, i: None)
    assert not N.qproj_rope_eligible(att, hidden, W)
    h.remove()
    att.q_proj._hf_hook = object()
    assert not N.qproj_rope_eligible(att, hidden, W)
    del att.q_proj._hf_hook
    assert N.qproj_rope_eligible(att, hidden, W)

    class Wrapped(torch.Tensor):
        pass

    plain = att.q_proj._parameters["weight"]
    att.q_proj._parameters["weight"] = plain.detach().as_subclass(Wrapped)
    try:
        assert not N.qproj_rope_eligible(att, hidden, W)
        keys = to_dev(s["keys"], "bf16")
        with torch.no_grad():
            sc = P.SnapKVPress(0.5, window_size=W, kernel_size=s["ks"]).score(att, hidden, keys, None, None, {"position_embeddings": pe})
        assert torch.isfinite(sc[..., :-W]).all()
    finally:
        att.q_proj._parameters["weight"] = plain


@pytest.mark.parametrize("K,Hq,dtname", [(4096, 32, "bf16"), (8192, 4, "bf16"), (4096, 4, "f16"), (1280, 2, "bf16"), (256, 2, "bf16")])
def test_qproj_rope_kernel_llama_sizes(K, Hq, dtname):
    """qproj.hip at the hidden sizes of Llama-3.1-8B (4096) and 70B (8192), at an odd tile count and at a single tile, batch 2, a
    strided hidden window, against the float64 product rounded once (<= 1 ulp of the projected value's magnitude through the RoPE
    pair)."""
    N = native()
    dt = _inputs.torch_dtype(dtname)
    g = torch.Generator().manual_seed(K + Hq)
    hidden = torch.randn(2, 96, K, generator=g).to(DEV, dt)
    wq = (torch.randn(Hq * 128, K, generator=g) * 0.02).to(DEV, dt)
    ang = torch.rand(2, 64, 64, generator=g) * 6.28
    cos, sin = torch.cat([ang.cos()] * 2, -1).to(DEV, dt), torch.cat([ang.sin()] * 2, -1).to(DEV, dt)
    hw = hidden[:, -64:]
    got = N.snapkv_qproj_rope(hw, wq, cos, sin)
    q64 = (hw.double() @ wq.double().T).view(2, 64, Hq, 128).transpose(1, 2)
    c, si = cos.double().unsqueeze(1), sin.double().unsqueeze(1)
    want64 = q64 * c + torch.cat((-q64[..., 64:], q64[..., :64]), dim=-1) * si
    ulp = 2.0 ** (-7 if dtname == "bf16" else -10)
    assert got.shape == (2, Hq, 64, 128) and got.dtype == dt
    assert ((got.double() - want64).abs() <= 4.0 * ulp * want64.abs().amax(dim=-1, keepdim=True) + 1e-6).all()


@pytest.mark.parametrize("name", ["sk_h512_bf16", "sk_h1024_f16"])
def test_hidden_path_scores_and_compress(name):
    """score / compress from the hidden states == the same from the library's own q_rot through the rotated-query entry."""
    s = _inputs.make_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, dt, DEV)
    W, S = s["W"], s["S"]
    k, v = to_dev(s["keys"], s["dtype"]), to_dev(s["values"], s["dtype"])
    N = native()
    hw, c, si = hidden[:, -W:], cos[:, -W:], sin[:, -W:]
    with torch.no_grad():
        q_rot = N.snapkv_qproj_rope(hw, att.q_proj.weight, c, si)
        want = N.snapkv_score(q_rot, k, s["ks"])
        got = N.snapkv_score_hidden(hw, att.q_proj.weight, c, si, k, s["ks"])
        assert torch.equal(got, want)
        for n in (W - 1, W, S // 2, S):
            ko, vo = N.snapkv_compress_hidden(hw, att.q_proj.weight, c, si, k, v, s["ks"], n)
            wk, wv = N.gather_kv(k, v, N.topk_select(want, n))
            assert torch.equal(ko, wk) and torch.equal(vo, wv), n
    # and the scores agree with the float64 oracle fed with the same (bf16) window queries within the north-star tolerance
    ref = O.snapkv_score(q_rot.float().cpu().numpy(), s["keys"], s["ks"])
    assert_scores_close(got.cpu().numpy()[..., :-W], ref[..., :-W], RTOL, name)
    # the press takes this path unless the switch is off (on by default since round 4)
    import kvpress_amd as P

    press = P.SnapKVPress(0.5, window_size=W, kernel_size=s["ks"])
    kw = {"position_embeddings": (cos, sin)}
    with torch.no_grad():
        saved = N.USE_LIBRARY_QPROJ
        try:
            N.USE_LIBRARY_QPROJ = True
            on = press.score(att, hidden, k, v, None, kw)
            ko_on, _ = press.compress(att, hidden, k, v, None, kw)
            N.USE_LIBRARY_QPROJ = False
            off = press.score(att, hidden, k, v, None, kw)
        finally:
            N.USE_LIBRARY_QPROJ = saved
    assert tuple(ko_on.shape) == (s["B"], s["H"], S // 2, s["D"])
    if s["B"] <= 2:   # up to two batch elements: the press projects in the library (round 6: profiles/r06_qproj_batch_lab.txt)
        assert torch.equal(on, got)
    else:             # a larger batch: the model's own GEMM reads the weight once for all its rows (qproj_rope_supported)
        assert torch.equal(on, off)
    # model GEMM vs library projection: the scores differ by no more than the rounding of the few differing queries explains
    with torch.no_grad():
        q_gemm = press.compute_window_queries(att, hidden, W, (cos, sin))
    if s["B"] <= 2:
        assert_scores_within_query_rounding(got, off, q_rot, q_gemm, k, W, name)
    if s["B"] == 2:   # three elements take the model's GEMM: the same scores as with the switch off
        h3, k3, v3 = torch.cat([hidden, hidden[:1]]), torch.cat([k, k[:1]]), torch.cat([v, v[:1]])
        with torch.no_grad():
            assert not N.qproj_rope_supported(att, h3, W)
            on3 = press.score(att, h3, k3, v3, None, kw)
            saved = N.USE_LIBRARY_QPROJ
            try:
                N.USE_LIBRARY_QPROJ = False
                off3 = press.score(att, h3, k3, v3, None, kw)
            finally:
                N.USE_LIBRARY_QPROJ = saved
        assert torch.equal(on3, off3)


# ---------------------------------------------------------------------------------------------
# degenerate shapes the reference's wrappers produce (ratio set to 1 through the attribute, one-token caches ...)
# ---------------------------------------------------------------------------------------------
def test_edge_cases_empty_and_tiny():
    import kvpress_amd as P

    N = native()
    k = torch.randn(2, 3, 17, 64, device=DEV, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    press = P.KnormPress(0.5)
    press.compression_ratio = 1.0                      # wrappers bypass the constructor's assert (per_layer_compression_press.py:56-61)
    ko, vo = press.compress(None, None, k, v, None, {})
    assert tuple(ko.shape) == tuple(vo.shape) == (2, 3, 0, 64) and ko.dtype == k.dtype
    # one-token cache: int(1 * 0.5) == 0 kept; int(1 * (1 - 0.4)) == 0 as well
    k1, v1 = k[:, :, :1], v[:, :, :1]
    ko, vo = P.KnormPress(0.4).compress(None, None, k1, v1, None, {})
    assert tuple(ko.shape) == (2, 3, 0, 64)
    # everything kept (a ratio so small that 1 - r == 1.0 in double, hence int(S * (1 - r)) == S): the very same rows, in order
    ko, vo = P.KnormPress(1e-17).compress(None, None, k, v, None, {})
    assert torch.equal(ko, k) and torch.equal(vo, v) and ko.data_ptr() != k.data_ptr()
    # modular entry points with empty selections
    sc = N.rownorm_score(k, -1.0)
    assert tuple(N.topk_select(sc, 0).shape) == (2, 3, 0)
    ko, vo = N.gather_kv(k, v, N.topk_select(sc, 0))
    assert tuple(ko.shape) == (2, 3, 0, 64)
    # a batch of one head and an odd head_dim (scalar kernels), non-contiguous views
    k3 = torch.randn(1, 1, 33, 10, device=DEV, dtype=torch.float32)[:, :, ::3]
    v3 = torch.randn(1, 1, 33, 10, device=DEV, dtype=torch.float32)[:, :, ::3]
    ko, vo = P.KnormPress(0.5).compress(None, None, k3, v3, None, {})
    idx = O.topk_select(O.knorm_score(k3.cpu().numpy()), 5)
    wk, wv = O.gather_kv(k3.cpu().numpy(), v3.cpu().numpy(), idx)
    assert np.array_equal(ko.cpu().numpy(), wk) and np.array_equal(vo.cpu().numpy(), wv)
    # CPU tensors are refused loudly: there is no fallback
    with pytest.raises(N.KvpressHipError):
        N.rownorm_score(k.cpu(), -1.0)


# ---------------------------------------------------------------------------------------------
# random shapes: ragged lengths, tile / chunk boundaries, every group size the matrix-core path takes
# ---------------------------------------------------------------------------------------------
def test_fuzz_shapes_snapkv_and_knorm():
    import kvpress_amd as P

    rs = np.random.RandomState(2024)
    N = native()
    lengths = [65, 66, 127, 128, 129, 191, 192, 193, 255, 256, 257, 1023, 1024, 1025, 1151, 1152, 2047, 2049]
    for it in range(36):
        B = int(rs.choice([1, 2]))
        H = int(rs.choice([1, 2, 3]))
        G = int(rs.choice([1, 2, 4, 8]))
        S = int(lengths[it % len(lengths)] if it < 2 * len(lengths) else rs.randint(65, 3000))
        dtname = "bf16" if it % 3 else "f16"
        ratio = float(rs.choice([0.1, 0.37, 0.5, 0.77, 0.93]))
        ks = int(rs.choice([1, 3, 5, 7]))
        Hq, W, D = H * G, 64, 128
        keys = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32) * rs.choice([0.3, 1.0, 3.0]), dtname)
        values = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32), dtname)
        q_win = _inputs.round_to(rs.standard_normal((B, Hq, W, D)).astype(np.float32), dtname)
        k, v, q = to_dev(keys, dtname), to_dev(values, dtname), to_dev(q_win, dtname)
        tag = f"it={it} B={B} H={H} G={G} S={S} {dtname} r={ratio} ks={ks}"
        sc = N.snapkv_score(q, k, ks)
        want = O.snapkv_score(q_win, keys, ks)
        assert_scores_close(sc.cpu().numpy()[..., :-W], want[..., :-W], RTOL, tag)
        n = int(S * (1 - ratio))
        idx = N.topk_select(sc, n).cpu().numpy()
        assert np.array_equal(idx, O.topk_select(sc.cpu().numpy(), n)), tag           # same scores -> same indices
        ok, msg = O.topk_is_valid(want, idx, n, rel_band=1e-3)
        assert ok, f"{tag}: {msg}"
        # fused paths == modular paths
        ko, vo = N.knorm_compress(k, v, n)
        wk, wv = N.gather_kv(k, v, N.topk_select(N.rownorm_score(k, -1.0), n))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), tag
        ones = torch.ones((1, W, D), device=DEV, dtype=k.dtype)
        zeros = torch.zeros((1, W, D), device=DEV, dtype=k.dtype)
        ko, vo = N.snapkv_compress_rope(q, ones, zeros, k, v, ks, n)                    # identity rotation: q_rot == q
        wk, wv = N.gather_kv(k, v, torch.from_numpy(idx).to(DEV))
        assert torch.equal(ko, wk) and torch.equal(vo, wv), tag


def test_fuzz_shapes_expected_attention_score():
    """kvp_ea_score over ragged lengths / group sizes / sink counts with synthetic statistics (symmetric PSD covariance)."""
    rs = np.random.RandomState(77)
    N = native()
    for it in range(16):
        B = int(rs.choice([1, 2]))
        H = int(rs.choice([1, 2]))
        G = int(rs.choice([1, 2, 4, 8]))
        S = int(rs.choice([70, 127, 128, 129, 1000, 2047, 2049, 4097])) if it < 8 else int(rs.randint(70, 5000))
        n_sink = int(rs.choice([0, 1, 4]))
        dtname = "bf16" if it % 2 else "f16"
        use_vnorm, use_cov = bool(it % 3), bool(it % 5)
        eps = float(rs.choice([0.0, 0.01]))
        Hq, D = H * G, 128
        keys = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32), dtname)
        values = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32) * 2.0, dtname)
        mu = (rs.standard_normal((B, Hq, D)) * 0.5).astype(np.float32)
        A = rs.standard_normal((B, Hq, D, 16)).astype(np.float32) * 0.3
        cov = (A @ A.transpose(0, 1, 3, 2) + 0.05 * np.eye(D, dtype=np.float32)).astype(np.float32) if use_cov else None
        want = O.ea_score(keys, values, mu, cov, n_sink, use_vnorm, eps)
        got = N.ea_score(to_dev(keys, dtname), to_dev(values, dtname), torch.from_numpy(mu).to(DEV),
                         torch.from_numpy(cov).to(DEV) if cov is not None else None, n_sink, use_vnorm, eps).cpu().numpy()
        tag = f"it={it} B={B} H={H} G={G} S={S} {dtname} sink={n_sink} vnorm={use_vnorm} cov={use_cov} eps={eps}"
        assert_scores_close(got[..., n_sink:], want[..., n_sink:], RTOL, tag)
        if n_sink:
            assert np.all(got[..., :n_sink] == np.float32(got[..., n_sink:].max()) + np.float32(1.0)), tag


def test_snapkv_extreme_logit_spread():
    """Keys aligned with a window query (logit ~ 2.5e4, far beyond the fp32 exponent range relative to their neighbours) in
    different sub-tiles of several tiles: the running maximum of pass 1 must follow them (a variant that refreshed the
    maximum only once per tile was measured at -2 us and dropped; this is the input that variant needed a fallback for)."""
    rs = np.random.RandomState(5)
    B, H, G, S, W, D = 1, 2, 4, 3000, 64, 128
    keys = rs.standard_normal((B, H, S, D)).astype(np.float32)
    q_win = rs.standard_normal((B, H * G, W, D)).astype(np.float32)
    for pos, hq, w in ((100, 0, 3), (40 + 128 * 5, 5, 60), (2900, 7, 10)):   # sub-tile 3, 1 and 2 of their tiles
        keys[0, hq // G, pos] = 200.0 * q_win[0, hq, w]
    keys, q_win = _inputs.round_to(keys, "bf16"), _inputs.round_to(q_win, "bf16")
    got = native().snapkv_score(to_dev(q_win, "bf16"), to_dev(keys, "bf16"), 5).cpu().numpy()
    want = O.snapkv_score(q_win, keys, 5)
    assert np.isfinite(got).all()
    assert_scores_close(got[..., :-W], want[..., :-W], RTOL, "extreme logit spread")


def test_snapkv_offset_raises_mid_walk(knobs):
    """Pass 1 keeps ONE offset per row and raises it only when a sub-tile's sum overflows 2^64 (tools/gen_stage_asm.py, "lazy offset").
    Long tile walks (KVP_SK_SLOTS=8: ~80 tiles per workgroup instead of one) with keys aligned with a window query at chosen places:
    a jump of ~100 log2 units late in a walk (raise in the steady-state loop), one of ~33 (below the threshold: NO raise, the term is
    2^33 times its neighbours), one beyond the float32 exponent range (inf -> raise), a huge key in the very first sub-tile (every
    later term underflows against it) and one in the last tile of a walk (raise in the drain) -- the oracle's scores throughout."""
    rs = np.random.RandomState(11)
    B, H, G, S, W, D = 1, 2, 4, 40000, 64, 128
    keys = rs.standard_normal((B, H, S, D)).astype(np.float32)
    q_win = rs.standard_normal((B, H * G, W, D)).astype(np.float32)
    #           position        q-head  row  alpha (logit = alpha * |q|^2 ~ alpha * 128; log2 units: ~16.3 * alpha)
    for pos, hq, w, alpha in ((3, 0, 1, 50.0), (128 * 31 + 70, 1, 7, 6.0), (128 * 150 + 5, 2, 33, 2.0), (128 * 200 + 127, 5, 60, 200.0),
                              (128 * 77 + 40, 6, 12, 6.0), (S - 64 - 1 - 128 * 3, 7, 63, 6.0), (128 * 290 + 64, 4, 0, 9.0)):
        keys[0, hq // G, pos] = alpha * q_win[0, hq, w]
    keys, q_win = _inputs.round_to(keys, "bf16"), _inputs.round_to(q_win, "bf16")
    want = O.snapkv_score(q_win, keys, 5)
    for slots in (8, 64, None):
        knobs(KVP_SK_SLOTS=slots)
        got = native().snapkv_score(to_dev(q_win, "bf16"), to_dev(keys, "bf16"), 5).cpu().numpy()
        assert np.isfinite(got).all()
        assert_scores_close(got[..., :-W], want[..., :-W], RTOL, f"offset raises, KVP_SK_SLOTS={slots}")


@pytest.mark.parametrize("G", [5, 8])
def test_snapkv_mfma_group_blocks_are_deterministic(G):
    """G > 4 splits a kv-head's query heads over two workgroups in pass 2; their column sums are merged in a fixed order (no
    float atomics): two runs give bit-identical scores, and they match the oracle."""
    g = torch.Generator().manual_seed(77)
    S = 5000
    keys = torch.randn((2, 2, S, 128), generator=g).to(torch.bfloat16).to(DEV)
    q = (torch.randn((2, 2 * G, 64, 128), generator=g) * 1.2).to(torch.bfloat16).to(DEV)
    a = native().snapkv_score(q, keys, 5)
    for _ in range(3):
        assert torch.equal(a, native().snapkv_score(q, keys, 5))
    ref = O.snapkv_score(q.float().cpu().numpy(), keys.float().cpu().numpy(), 5)
    got = a.cpu().numpy()
    np.testing.assert_allclose(got[..., :-64], ref[..., :-64], rtol=1e-3)


@pytest.mark.parametrize("S,W", [(8256, 64), (8260, 64), (12288 + 64, 64), (20000, 64), (40000, 64), (65600, 64), (8228, 30), (9000, 2)])
def test_snapkv_pool_vector_and_scalar_paths(S, W, knobs):
    """Long rows with kernel_size 5 are pooled four scores per thread (8-byte loads, one 16-byte store; aligned rows only), other
    rows one score per thread -- the same additions in the same order.  Both shapes occur in this list (S - W = 2 mod 4 leaves a
    partial last group of four; windows other than 64 take the generic attention kernels): the scores match the oracle, and the
    fused compress -- through the cluster select's pooling loader (default) and through the pooling kernel with its fused
    histogram + the (chunk, row) passes (KVP_TK_CLUSTER=0) -- returns the modular sequence's bytes."""
    g = torch.Generator().manual_seed(S)
    keys = torch.randn((1, 2, S, 128), generator=g).to(torch.bfloat16).to(DEV)
    vals = torch.randn((1, 2, S, 128), generator=g).to(torch.bfloat16).to(DEV)
    q = (torch.randn((1, 8, W, 128), generator=g) * 1.2).to(torch.bfloat16).to(DEV)
    cos = torch.ones((1, W, 128), dtype=torch.bfloat16, device=DEV)
    sin = torch.zeros((1, W, 128), dtype=torch.bfloat16, device=DEV)
    N = native()
    sc = N.snapkv_score(q, keys, 5)
    ref = O.snapkv_score(q.float().cpu().numpy(), keys.float().cpu().numpy(), 5)
    np.testing.assert_allclose(sc.cpu().numpy()[..., :-W], ref[..., :-W], rtol=1e-3)
    wk, wv = N.gather_kv(keys, vals, N.topk_select(sc, S // 2))
    for cluster in (None, 0):
        knobs(KVP_TK_CLUSTER=cluster)
        ko, vo = N.snapkv_compress_rope(q, cos, sin, keys, vals, 5, S // 2)
        assert torch.equal(ko, wk) and torch.equal(vo, wv), (S, W, cluster)
