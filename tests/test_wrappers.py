"""ChunkPress / KeyRerotationPress against the REAL reference's outputs (tests/golden/wrap_*.npz, made by
oracle/gen_golden_wrappers.py).  CPU: oracle restatement and the host logic of the wrappers over oracle-backed entry
points; GPU (marked): the same classes on the HIP kernels, plus kernel-level checks of the two new entry points."""
import os

import numpy as np
import pytest
import torch

import _inputs
from oracle import kvpress_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ADAKV = [n for n, c in _inputs.WRAP_CASES.items() if c["wrapper"] == "adakv"]
BLOCK = [n for n, c in _inputs.WRAP_CASES.items() if c["wrapper"] == "block"]
CHUNK = [n for n, c in _inputs.WRAP_CASES.items() if c["wrapper"] == "chunk"]
CHUNKKV = [n for n, c in _inputs.WRAP_CASES.items() if c["wrapper"] == "chunkkv"]
REROT = [n for n, c in _inputs.WRAP_CASES.items() if c["wrapper"] == "rerot"]
DEV = "cuda:0"



def _pair_rel(got, ref):
    """|got - ref| relative to the norm of the (d, d + D/2) pair: a rotation preserves that norm, and a one-ulp difference in
    cos / sin moves an output by up to one ulp OF THE PAIR NORM (more than an ulp of the output where the two products cancel)."""
    half = ref.shape[-1] // 2
    pn = np.sqrt(ref[..., :half] ** 2 + ref[..., half:] ** 2)
    return np.abs(got - ref) / np.maximum(np.concatenate([pn, pn], axis=-1), 1e-3)

def _elem_rel_no_cancellation(got, ref):
    """Per-ELEMENT relative error where the rotation's two products do not cancel (|out| >= half the pair norm): there the pair bound
    of 2.1 ulp of the norm is at most 4.2 ulp of the element itself (keep a per-element bound where one is meaningful)."""
    half = ref.shape[-1] // 2
    pn = np.sqrt(ref[..., :half] ** 2 + ref[..., half:] ** 2)
    pn = np.concatenate([pn, pn], axis=-1)
    m = (np.abs(ref) >= 0.5 * pn) & (pn > 1e-3)
    return float((np.abs(got - ref)[m] / np.abs(ref)[m]).max()) if m.any() else 0.0


def gold(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def inner_press(s, ratio):
    import kvpress_amd as P

    return {"knorm": lambda: P.KnormPress(ratio), "keydiff": lambda: P.KeyDiffPress(ratio),
            "snapkv": lambda: P.SnapKVPress(ratio, window_size=s["W"], kernel_size=s["ks"]),
            "streaming": lambda: P.StreamingLLMPress(ratio, n_sink=s["n_sink"])}[s["kind"]]()


def wrapped(s, ratio):
    import kvpress_amd as P

    if s["wrapper"] == "adakv":
        return P.AdaKVPress(inner_press(s, ratio), alpha_safeguard=s["alpha"])
    if s["wrapper"] == "block":
        return P.contrib.BlockPress(inner_press(s, ratio), block_size=s["block_size"])
    if s["wrapper"] == "chunkkv":
        return P.contrib.ChunkKVPress(inner_press(s, ratio), chunk_length=s["chunk_length"])
    return P.ChunkPress(inner_press(s, ratio), chunk_length=s["chunk_length"]) if s["wrapper"] == "chunk" else P.KeyRerotationPress(inner_press(s, ratio))


def oracle_chunk_scores(s, cos=None, sin=None):
    """score_chunk(i, j) of the inner press restated by the oracle on chunk [i, j)."""
    if s["kind"] == "knorm":
        return lambda i, j: O.knorm_score(s["keys"][:, :, i:j])
    if s["kind"] == "keydiff":
        return lambda i, j: O.keydiff_score(s["keys"][:, :, i:j])
    W = s["W"]

    def snap(i, j):  # the wrapped SnapKV sees the chunk's hidden states but the FULL sequence's last-W rotary tables
        h = s["hidden"][:, i:j]
        q = O.snapkv_window_queries(h, s["wq"], None, cos, sin, s["Hq"], s["D"], W)
        return O.snapkv_score(q, s["keys"][:, :, i:j], s["ks"])
    return snap


@pytest.mark.parametrize("name", CHUNK)
def test_oracle_chunk_indices_match_reference(name):
    s = _inputs.make_wrap_case(name)
    g = gold(name)
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, torch.float32)
    fn = oracle_chunk_scores(s, cos.numpy(), sin.numpy())
    for i, r in enumerate(s["ratios"]):
        idx = np.sort(O.chunk_press_indices(fn, s["S"], s["chunk_length"], r), axis=-1)
        assert np.array_equal(idx, g[f"pos_{i}"]), f"{name} r={r}"


@pytest.mark.parametrize("name", CHUNKKV)
def test_oracle_chunkkv_indices_match_reference(name):
    s = _inputs.make_wrap_case(name)
    g = gold(name)
    sc = _oracle_scores_full(s)
    for i, r in enumerate(s["ratios"]):
        if s["S"] < s["chunk_length"]:   # no complete chunk: the wrapped press's own selection (in torch.topk's order there)
            assert np.array_equal(O.topk_select(sc, O.n_kept(s["S"], r)), np.sort(g[f"pos_{i}"], axis=-1)), f"{name} r={r}"
        else:
            want = np.broadcast_to(O.chunkkv_indices(sc, s["chunk_length"], r), g[f"pos_{i}"].shape)
            assert np.array_equal(want, g[f"pos_{i}"]), f"{name} r={r}"


def _check_chunkkv(s, name, dev):
    g, out = _run_wrapper(s, name, dev, torch.float32)
    for i, r, ko, pos in out:
        assert np.array_equal(pos, np.sort(g[f"pos_{i}"], axis=-1)), f"{name} r={r}: kept positions"
        wk, _ = O.gather_kv(s["keys"], s["values"], pos)
        assert np.array_equal(ko, wk)


@pytest.mark.parametrize("name", CHUNKKV)
def test_chunkkv_matches_reference_cpu(name, fake_native):
    _check_chunkkv(_inputs.make_wrap_case(name), name, "cpu")


@pytest.mark.parametrize("name", REROT)
def test_oracle_rerotation_matches_reference(name):
    s = _inputs.make_wrap_case(name)
    g = gold(name)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
    inv = rot.inv_freq.numpy()
    for i, r in enumerate(s["ratios"]):
        for mode, dt in (("f32", "f32"), ("nat", s["dtype"])):
            pos = g[f"pos_{i}"] if mode == "f32" else g[f"pos_nat_{i}"]
            kk, _ = O.gather_kv(s["keys"], s["values"], pos)
            got = O.rerotate_keys(kk, pos, inv, dt)
            ref = g[f"kout_{mode}_{i}"]
            if dt == "f32":
                np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6)
            else:  # per-op rounding in the key dtype: identical up to a 1-ulp flip where fp32 cos/sin round differently
                ulp = 2.0 ** (-8 if dt == "bf16" else -11)
                assert np.mean(got != ref) < 2e-3 and _pair_rel(got, ref).max() <= 2.1 * ulp
                assert _elem_rel_no_cancellation(got, ref) <= 4.2 * ulp


def _run_wrapper(s, name, dev, dt):
    g = gold(name)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, dev)
    att.rotary_emb = rot
    keys = torch.from_numpy(s["keys"]).to(device=dev, dtype=dt)
    posv = torch.arange(s["S"], dtype=torch.float32, device=dev)[None, None, :, None].expand(s["B"], s["H"], s["S"], s["D"]).contiguous()
    kwargs = {"position_embeddings": pe}
    out = []
    with torch.no_grad():
        for i, r in enumerate(s["ratios"]):
            ko, vo = wrapped(s, r).compress(att, hidden, keys, posv, None, kwargs)
            assert ko.is_contiguous() and ko.dtype == dt and tuple(ko.shape[:2]) == (s["B"], s["H"])
            out.append((i, r, ko.float().cpu().numpy(), vo[..., 0].round().to(torch.int64).cpu().numpy()))
        k0, v0 = wrapped(s, 0.0).compress(att, hidden, keys, posv, None, kwargs)
        assert k0 is keys and v0 is posv
    return g, out


@pytest.mark.parametrize("name", CHUNK + REROT)
def test_wrappers_match_reference_cpu(name, fake_native):
    s = _inputs.make_wrap_case(name)
    g, out = _run_wrapper(s, name, "cpu", torch.float32)
    for i, r, ko, pos in out:
        assert np.array_equal(pos, g[f"pos_{i}"]), f"{name} r={r}: kept positions"   # ours come out sorted
        if s["wrapper"] == "rerot":
            np.testing.assert_allclose(ko, g[f"kout_f32_{i}"], rtol=1e-5, atol=2e-6)
        else:
            wk, _ = O.gather_kv(s["keys"], s["values"], pos)
            assert np.array_equal(ko, wk)


def _oracle_scores_full(s):
    if s["kind"] == "knorm":
        return O.knorm_score(s["keys"])
    att, rot, hidden, (cos, sin) = _inputs.build_llama_attention(s, torch.float32)
    q = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos.numpy(), sin.numpy(), s["Hq"], s["D"], s["W"])
    return O.snapkv_score(q, s["keys"], s["ks"])


@pytest.mark.parametrize("name", ADAKV)
def test_oracle_adakv_matches_reference(name):
    s = _inputs.make_wrap_case(name)
    g = gold(name)
    sc = _oracle_scores_full(s)
    for i, r in enumerate(s["ratios"]):
        assert np.array_equal(O.adakv_pruned(sc, r, s["alpha"]), g[f"masked_{i}"]), f"{name} r={r}"


def _run_adakv(s, name, dev):
    g = gold(name)
    att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32, dev)
    att.config._attn_implementation = "sdpa"
    keys = torch.from_numpy(s["keys"]).to(dev)
    values = torch.from_numpy(s["values"]).to(dev)
    with torch.no_grad():
        for i, r in enumerate(s["ratios"]):
            ko, vo = wrapped(s, r).compress(att, hidden, keys, values, None, {"position_embeddings": pe})
            assert ko is keys and vo is values                       # nothing is removed from the cache
            bi, hi, si = att.masked_key_indices
            n_pruned = s["H"] * (s["S"] - int(s["S"] * (1 - r)))
            assert bi.shape == hi.shape == si.shape == (s["B"] * n_pruned,) and bi.dtype == torch.int64
            assert torch.equal(bi.cpu(), torch.arange(s["B"]).repeat_interleave(n_pruned))
            flat = (hi * s["S"] + si).reshape(s["B"], -1).cpu().numpy()
            assert np.array_equal(flat, g[f"masked_{i}"]), f"{name} r={r}"  # ours come out sorted


@pytest.mark.parametrize("name", ADAKV)
def test_adakv_matches_reference_cpu(name, fake_native):
    _run_adakv(_inputs.make_wrap_case(name), name, "cpu")


def test_attention_patch_masks_keys():
    """The patched attention function gives masked keys (numerically) zero weight while decoding and resets the mask on
    the next prefill -- checked against attention over the physically pruned keys."""
    from kvpress_amd.attention_patch import attention_patch, search_hyperplane

    torch.manual_seed(0)
    X = torch.randn(6, 12, 16) + 0.3
    Y = search_hyperplane(X)
    assert (torch.bmm(X, Y.unsqueeze(-1)) < -1e3).all()

    def plain(module, q, k, v, mask, dropout, **kw):
        w = torch.softmax(q @ k.repeat_interleave(q.shape[1] // k.shape[1], 1).transpose(-1, -2) / q.shape[-1] ** 0.5, -1)
        return w @ v.repeat_interleave(q.shape[1] // k.shape[1], 1), w

    patched = attention_patch(plain)
    assert attention_patch(patched) is patched                     # idempotent

    class M:
        masked_key_indices = None
    m = M()
    B, Hq, Hkv, S, D = 2, 4, 2, 10, 8
    q, k, v = torch.randn(B, Hq, 1, D), torch.randn(B, Hkv, S, D), torch.randn(B, Hkv, S, D)
    b_idx = torch.tensor([0, 0, 1]); h_idx = torch.tensor([0, 1, 1]); s_idx = torch.tensor([3, 5, 0])
    m.masked_key_indices = (b_idx, h_idx, s_idx)
    out, w = patched(m, q, k.clone(), v, None, 0.0)
    wg = w.view(B, Hkv, Hq // Hkv, 1, S)
    assert float(wg[0, 0, :, 0, 3].max()) == 0.0 and float(wg[0, 1, :, 0, 5].max()) == 0.0 and float(wg[1, 1, :, 0, 0].max()) == 0.0
    # same output as attention over the caches with those keys physically removed
    for b, h, s_ in ((0, 0, 3), (1, 1, 0)):
        keep = [i for i in range(S) if i != s_]
        for gq in range(Hq // Hkv):
            hq = h * (Hq // Hkv) + gq
            ww = torch.softmax(q[b, hq] @ k[b, h, keep].T / D ** 0.5, -1)
            assert torch.allclose(out[b, hq], ww @ v[b, h, keep], atol=1e-6)
    # a prefill-shaped call (q_len == k_len) clears the mask
    patched(m, torch.randn(B, Hq, S, D), k, v, None, 0.0)
    assert m.masked_key_indices is None


def _check_block(s, name, dev):
    g, out = _run_wrapper(s, name, dev, torch.float32)
    for i, r, ko, pos in out:
        ref = g[f"pos_{i}"]
        assert pos.shape == ref.shape
        if s["kind"] == "snapkv":
            # position-aware scorer inside an iteration: a flipped near-tie changes the candidate order of the next round;
            # the survivors must still agree almost everywhere
            same = np.mean([len(np.intersect1d(a, b)) / a.size for a, b in zip(pos.reshape(-1, pos.shape[-1]), ref.reshape(-1, ref.shape[-1]))])
            assert same >= 0.97, f"{name} r={r}: overlap {same:.3f}"
        else:
            assert np.array_equal(pos, ref), f"{name} r={r}: survivors and their (descending-score) order"
        wk, _ = O.gather_kv(s["keys"], s["values"], pos)
        assert np.array_equal(ko, wk)


@pytest.mark.parametrize("name", BLOCK)
def test_block_press_matches_reference_cpu(name, fake_native):
    _check_block(_inputs.make_wrap_case(name), name, "cpu")


def test_block_press_is_streaming_top_k(fake_native):
    """The reference's own test (tests/presses/test_block_press.py:30-63): with a scorer that depends on the token alone,
    block-wise selection keeps exactly the global top-k, whatever the block size."""
    from dataclasses import dataclass

    from transformers import DynamicCache

    import kvpress_amd as P

    @dataclass
    class HiddenStatesPress(P.ScorerPress):
        def score(self, module, hidden_states, keys, values, attentions, kwargs):
            return hidden_states.mean(-1).unsqueeze(1).expand_as(keys.norm(dim=-1)).float().contiguous()

    model = _inputs.make_tiny_llama()
    ids = torch.randint(3, 59, (1, 256), generator=torch.Generator().manual_seed(0))
    sums = []
    for press in [P.contrib.BlockPress(press=HiddenStatesPress(0.5), block_size=b) for b in (2, 4, 8, 128, 256)] + [HiddenStatesPress(0.5)]:
        cache = DynamicCache()
        with torch.no_grad(), press(model):
            model(ids, past_key_values=cache)
        assert cache.get_seq_length() == 128
        sums.append((torch.cat([l.keys for l in cache.layers]).sum().item(), torch.cat([l.values for l in cache.layers]).sum().item()))
    t = torch.tensor(sums)
    assert torch.allclose(t, t[-1])


def test_chunk_press_asserts(fake_native):
    import kvpress_amd as P

    with pytest.raises(AssertionError):
        P.ChunkPress(press=P.ChunkPress(P.KnormPress(0.5)), chunk_length=8)
    cp = P.ChunkPress(P.KnormPress(0.5), chunk_length=8)
    cp.compression_ratio = 0.25
    assert cp.press.compression_ratio == 0.25
    k = torch.zeros(1, 1, 16, 4)
    with pytest.raises(AssertionError):
        cp.compress(None, torch.zeros(1, 16, 8), k, k, torch.zeros(1), {})


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", CHUNK + REROT)
def test_wrappers_match_reference_gpu_fp32(name):
    s = _inputs.make_wrap_case(name)
    g, out = _run_wrapper(s, name, DEV, torch.float32)
    for i, r, ko, pos in out:
        assert np.array_equal(pos, g[f"pos_{i}"]), f"{name} r={r}: kept positions"
        if s["wrapper"] == "rerot":
            np.testing.assert_allclose(ko, g[f"kout_f32_{i}"], rtol=1e-5, atol=4e-6)
        else:
            wk, _ = O.gather_kv(s["keys"], s["values"], pos)
            assert np.array_equal(ko, wk)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CHUNKKV)
def test_chunkkv_matches_reference_gpu(name):
    _check_chunkkv(_inputs.make_wrap_case(name), name, DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BLOCK)
def test_block_press_matches_reference_gpu(name):
    _check_block(_inputs.make_wrap_case(name), name, DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ADAKV)
def test_adakv_matches_reference_gpu(name):
    _run_adakv(_inputs.make_wrap_case(name), name, DEV)


@pytest.mark.gpu
def test_topk_smallest_and_fill_at():
    from kvpress_amd import _native

    rs = np.random.RandomState(11)
    sc = rs.standard_normal((3, 5000)).astype(np.float32)
    sc[:, ::9] = -0.5                                            # ties
    t = torch.from_numpy(sc).to(DEV)
    for k in (1, 17, 2500, 4999, 5000):
        got = _native.topk_select(t, k, _native.ORDER_POSITION | _native.TOPK_SMALLEST).cpu().numpy()
        assert np.array_equal(got, O.topk_select(-sc, k)), k
    idx = _native.topk_select(t, 100)
    _native.scores_fill_at_(t, idx, 7.5)
    want = sc.copy()
    np.put_along_axis(want, idx.cpu().numpy().astype(np.int64), 7.5, axis=-1)
    assert np.array_equal(t.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in REROT if _inputs.WRAP_CASES[n]["dtype"] != "f32"])
def test_rerotation_native_dtype_gpu(name):
    """bf16 / f16 keys: the kernel's per-op rounding against the reference's, on the reference's own kept positions."""
    from kvpress_amd import _native

    s = _inputs.make_wrap_case(name)
    g = gold(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, DEV)
    keys = torch.from_numpy(s["keys"]).to(device=DEV, dtype=dt)
    ulp = 2.0 ** (-8 if s["dtype"] == "bf16" else -11)
    for i, r in enumerate(s["ratios"]):
        pos = torch.from_numpy(g[f"pos_nat_{i}"].astype(np.int32)).to(DEV)
        ko, _ = _native.gather_kv(keys, keys, pos)
        got = _native.rerotate_keys_(ko, pos, rot.inv_freq).float().cpu().numpy()
        ref = g[f"kout_nat_{i}"]
        assert np.mean(got != ref) < 2e-3, f"{name}: {np.mean(got != ref):.2e} of the elements differ"
        assert _pair_rel(got, ref).max() <= 2.1 * ulp and _elem_rel_no_cancellation(got, ref) <= 4.2 * ulp
        # and bit-identical to the oracle's emulation of the same rounding wherever cosf/sinf agree with numpy
        want = O.rerotate_keys(s["keys"][np.arange(s["B"])[:, None, None], np.arange(s["H"])[None, :, None], g[f"pos_nat_{i}"]],
                               g[f"pos_nat_{i}"], rot.inv_freq.cpu().numpy(), s["dtype"])
        assert np.mean(got != want) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_gather_rerotate_one_pass_equals_two_kernels(dtype):
    """kvp_gather_kv_rerotate (KeyRerotationPress / FinchPress) == kvp_gather_kv then kvp_rerotate_keys, bit for bit: the one-pass kernel
    (2-byte dtypes, D % 16 == 0), its cached / streaming variants, views with strides, and the shapes that fall back to the two kernels."""
    from kvpress_amd import _native

    dt = _inputs.torch_dtype(dtype)
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    for B, H, S, D, n in ((1, 8, 4096, 128, 2048), (2, 3, 1000, 64, 333), (1, 2, 777, 128, 777), (1, 4, 513, 48, 100), (1, 1, 50, 6, 7),
                          (1, 8, 70000, 128, 1)):
        k = torch.randn((B, H, S, D), generator=g, device=DEV).to(dt)
        v = torch.randn((B, H, S, D), generator=g, device=DEV).to(dt)
        pos = torch.stack([torch.randperm(S, generator=g, device=DEV)[:n].sort().values for _ in range(B * H)]).view(B, H, n).to(torch.int32)
        inv = (10000.0 ** (-torch.arange(0, D, 2, device=DEV, dtype=torch.float32) / D))
        ko, vo = _native.gather_kv(k, v, pos)
        _native.rerotate_keys_(ko, pos, inv)
        k1, v1 = _native.gather_kv_rerotate(k, v, pos, inv)   # one pass (16-bit dtypes, D % 16 == 0) == the two kernels above
        assert torch.equal(k1, ko) and torch.equal(v1, vo), (dtype, B, H, S, D, n)
        if D % 16 == 0:   # a [B, S, H, D] cache layout seen through transpose(1, 2): rows 16-byte aligned, not contiguous
            kt = k.transpose(1, 2).contiguous().transpose(1, 2)
            vt = v.transpose(1, 2).contiguous().transpose(1, 2)
            k2, v2 = _native.gather_kv_rerotate(kt, vt, pos, inv)
            assert torch.equal(k2, ko) and torch.equal(v2, vo), (dtype, B, H, S, D, n, "strided")


@pytest.mark.gpu
def test_topk_segmented_vs_oracle():
    from kvpress_amd import _native

    rs = np.random.RandomState(9)
    for R, nseg, L, k, base in ((4, 7, 100, 33, 0), (2, 3, 2048, 1024, 5), (8, 128, 1024, 512, 0), (3, 1, 77, 77, 1000), (1, 5, 64, 1, 0),
                                (2, 5, 20000, 7000, 3), (3, 2, 40001, 20000, 0)):   # (long segments: the cluster select with segment offsets)
        sc = rs.standard_normal((R, nseg * L)).astype(np.float32)
        sc[:, ::7] = 0.25  # ties inside every chunk
        got = _native.topk_select_segmented(torch.from_numpy(sc).to(DEV), L, k, pos_base=base).cpu().numpy()
        want = np.concatenate([base + c * L + O.topk_select(sc[:, c * L:(c + 1) * L], k) for c in range(nseg)], axis=-1)
        assert got.dtype == np.int32 and np.array_equal(got, want), (R, nseg, L, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wrap_chunk_knorm_bf16", "wrap_chunk_snapkv"])
def test_chunk_press_batched_scoring_equals_per_chunk_loop(name):
    """The chunks-as-batch scoring of ChunkPress gives exactly what scoring every chunk on its own gives."""
    from kvpress_amd import _native

    s = _inputs.make_wrap_case(name)
    dt = _inputs.torch_dtype(s["dtype"])
    att, rot, hidden, pe = _inputs.build_llama_attention(s, dt, DEV)
    keys = torch.from_numpy(s["keys"]).to(device=DEV, dtype=dt)
    values = torch.from_numpy(s["values"]).to(device=DEV, dtype=dt)
    kwargs = {"position_embeddings": pe}
    r, L = s["ratios"][0], s["chunk_length"]
    with torch.no_grad():
        ko, vo = wrapped(s, r).compress(att, hidden, keys, values, None, kwargs)
        p = inner_press(s, r)
        parts = []
        for i in range(0, s["S"], L):
            sc = p.score(att, hidden[:, i:i + L], keys[:, :, i:i + L], values[:, :, i:i + L], None, kwargs)
            n = max(1, int(sc.shape[-1] * (1 - r)))
            parts.append(i + _native.topk_select(sc, n))
        wk, wv = _native.gather_kv(keys, values, torch.cat(parts, dim=-1))
    assert torch.equal(ko, wk) and torch.equal(vo, wv)
