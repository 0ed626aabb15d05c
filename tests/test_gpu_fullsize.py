"""Parity at BASELINE.json's full sizes (configs 1-4) against the REAL reference and through size-independent properties:

  * reference: tests/golden/full_*.npz hold the outputs of NVIDIA/kvpress itself (oracle/gen_golden_fullsize.py, float32-mode
              "O32" and bf16 "Obf" runs on the CPU-seeded inputs of tests/_fullsize.py): kernel scores within 1e-3 of the
              reference's float32 scores, retained set identical up to the tolerance band at the threshold (SURVEY §8c i-ii),
              and the dtype-faithful check against the reference's own bf16 scores with a bf16-ulp band (§8c iii);
  * scores  : additionally |kernel - torch fp32 restatement on the GPU| <= 1e-3 relative on every position;
  * top-k   : indices ascending/unique/in range; partition property (every kept score >= every
              dropped score); the multiset of kept score values equals torch.topk's on the same
              scores; idempotence (selecting all kept again returns them all);
  * gather  : K'/V' bit-identical to torch.gather with the kernel's indices; inputs untouched.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H_Q, H_KV, D, HIDDEN = 32, 8, 128, 4096


def _native():
    from kvpress_amd import _native

    return _native


def check_topk_and_gather(scores, keys, values, n, ko, vo):
    nat = _native()
    B, H, S = scores.shape
    idx = nat.topk_select(scores, n).long()
    assert idx.shape == (B, H, n)
    assert (idx[..., 1:] > idx[..., :-1]).all(), "indices must be strictly ascending"
    assert idx.min() >= 0 and idx.max() < S
    kept = torch.zeros((B, H, S), dtype=torch.bool, device=scores.device)
    kept.scatter_(2, idx, True)
    assert int(kept.sum()) == B * H * n
    s_kept = scores.gather(2, idx)
    dropped_max = scores.masked_fill(kept, float("-inf")).amax(-1)
    assert (s_kept.amin(-1) >= dropped_max).all(), "a dropped score exceeds a kept one"
    ref_vals = scores.topk(n, dim=-1).values.sort(-1).values
    assert torch.equal(s_kept.sort(-1).values, ref_vals), "kept score multiset differs from torch.topk"
    # tie rule: among scores equal to the threshold the lowest positions are kept
    t = s_kept.amin(-1, keepdim=True)
    eq = scores == t
    eq_kept = (eq & kept).sum(-1)
    first_eq = (eq.cumsum(-1) <= eq_kept.unsqueeze(-1)) & eq
    assert torch.equal(first_eq, eq & kept), "ties at the threshold are not resolved lowest-position-first"
    # idempotence
    again = nat.topk_select(s_kept, n).long()
    assert torch.equal(again, torch.arange(n, device=scores.device).expand(B, H, n))
    # gather
    e = idx.unsqueeze(-1).expand(-1, -1, -1, keys.shape[-1])
    assert torch.equal(ko, keys.gather(2, e)) and torch.equal(vo, values.gather(2, e))
    assert ko.is_contiguous() and vo.is_contiguous()


def llama_module():
    import bench

    return bench.build_module(torch.device(DEV))


def full_case(name):
    """CPU-seeded full-size inputs (tests/_fullsize.py) on the GPU + the reference fixture."""
    import os

    import _fullsize as F

    spec = F.FULL_CASES[name]
    keys, values = (t.to(DEV) for t in F.make_kv(spec))
    hidden = F.make_hidden(spec).to(DEV)
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
    return spec, keys, values, hidden, fx


# overlap of the float32-selected set with the bf16 reference's own torch.topk choice, measured when the fixtures were made
# (oracle/gen_golden_fullsize.py prints it); the tests allow 1 % below.  All disagreements sit within 2 bf16 ulps of the threshold.
NATIVE_OVERLAP = {"full_knorm32k": 0.9961, "full_snapkv128k": 0.9420, "full_snapkv128k_B": 0.9837, "full_ea128k": 0.9975}
NATIVE_ULPS = 3


def test_config2_knorm_32k():
    import _fullsize as F
    import kvpress_amd as P

    spec, keys, values, hidden, fx = full_case("full_knorm32k")
    S = spec["S"]
    kc, vc = keys.clone(), values.clone()
    press = P.KnormPress(0.5)
    sc = press.score(None, None, keys, values, None, {})
    ref = -keys.float().norm(dim=-1)
    assert ((sc - ref).abs() <= 1e-5 * ref.abs()).all()
    ko, vo = press.compress(type("M", (), {"head_dim": D})(), None, keys, values, None, {})
    assert ko.shape == (1, H_KV, 16384, D) and ko.dtype == torch.bfloat16
    check_topk_and_gather(sc, keys, values, 16384, ko, vo)
    assert torch.equal(keys, kc) and torch.equal(values, vc), "inputs must not be modified"
    idx = _native().topk_select(sc, 16384)
    worst, differ = F.check_against_reference(fx, sc, idx)
    overlap, _ = F.check_against_native(fx, idx, S, NATIVE_ULPS, NATIVE_OVERLAP["full_knorm32k"] - 0.01)
    print(f"knorm32k vs reference: max rel err {worst:.2e}, {differ} set differences (in band), overlap with bf16 reference {overlap:.4f}")


@pytest.mark.parametrize("name", ["full_keydiff128k", "full_cur128k"])
def test_f2_scorers_128k_vs_reference(name):
    """SURVEY §8(f-2) scorers at the BASELINE size against outputs of the REAL reference (oracle/gen_golden_fullsize.py: KeyDiffPress /
    CURPress in float32 and in bf16 on structured keys and values): scores within 1e-3 (KeyDiff's cosines cross zero: + 2e-6 absolute),
    the retained set identical outside the tolerance band, and a floor on the overlap with what the bf16 reference itself keeps (its
    bf16 scores near a threshold of -0.003 are only defined to ~170 ulps for KeyDiff, 4 for CUR: measured when the fixtures were made)."""
    import _fullsize as F
    import kvpress_amd as P

    spec, keys, values, hidden, fx = full_case(name)
    S, n = spec["S"], spec["S"] // 2
    kc, vc = keys.clone(), values.clone()
    press = P.KeyDiffPress(0.5) if spec["kind"] == "keydiff" else P.CURPress(0.5)
    sc = press.score(None, None, keys, values, None, {})
    assert sc.dtype == torch.float32 and tuple(sc.shape) == (1, H_KV, S)
    ko, vo = press.compress(type("M", (), {"head_dim": D})(), None, keys, values, None, {})
    check_topk_and_gather(sc, keys, values, n, ko, vo)
    assert torch.equal(keys, kc) and torch.equal(values, vc), "inputs must not be modified"
    idx = _native().topk_select(sc, n)
    kd = spec["kind"] == "keydiff"
    worst, differ = F.check_against_reference(fx, sc, idx, atol=2e-6 if kd else 0.0, dense_atol=2e-5 if kd else None)
    ulps, floor = (256, 0.9989 - 0.01) if spec["kind"] == "keydiff" else (6, 0.9990 - 0.01)
    overlap, _ = F.check_against_native(fx, idx, S, ulps, floor)
    print(f"{name} vs reference: max rel err {worst:.2e}, {differ} set differences (in band), overlap with bf16 reference {overlap:.4f}")


def torch_snapkv_reference(q_win, keys, kernel_size):
    """fp32 restatement of snapkv_press.py:60-105 with torch ops on the GPU (q_win already RoPE'd)."""
    B, Hq, W, Dh = q_win.shape
    H, S = keys.shape[1], keys.shape[2]
    G = Hq // H
    q = q_win.float()
    k = keys.float().repeat_interleave(G, dim=1)
    attn = torch.matmul(q, k.transpose(2, 3)) / (Dh ** 0.5)
    mask = torch.triu(torch.full_like(attn, float("-inf")), diagonal=S - W + 1)
    attn = torch.softmax(attn + mask, dim=-1)[..., :-W]
    sc = attn.mean(dim=-2)
    sc = torch.nn.functional.avg_pool1d(sc, kernel_size=kernel_size, padding=kernel_size // 2, stride=1)
    sc = sc.view(B, H, G, S - W).mean(2)
    return torch.nn.functional.pad(sc, (0, W), value=sc.max().item() + 1)


@pytest.mark.parametrize("name", ["full_snapkv128k", "full_snapkv128k_B"])
def test_config3_snapkv_128k(name):
    import _fullsize as F
    import kvpress_amd as P

    spec, keys, values, hidden, fx = full_case(name)
    S = spec["S"]
    att, rot = llama_module()
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=DEV)[None])
        press = P.SnapKVPress(0.5)
        kwargs = {"position_embeddings": pe}
        sc = press.score(att, hidden, keys, values, None, kwargs)
        q_win = press.compute_window_queries(att, hidden, 64, pe)
        ref = torch_snapkv_reference(q_win, keys, 5)
        rel = ((sc - ref).abs() / ref.abs().clamp_min(1e-30))[..., :-64]
        assert rel.max() <= 1e-3, f"max rel err {rel.max().item():.3e}"
        assert (sc[..., -64:] == sc[..., :-64].max() + 1).all()
        n = int(S * 0.5)
        ko, vo = press.compress(att, hidden, keys, values, None, kwargs)
        assert ko.shape == (1, H_KV, n, D)
        check_topk_and_gather(sc, keys, values, n, ko, vo)
        idx = _native().topk_select(sc, n)
        assert (idx[..., -64:] == torch.arange(S - 64, S, device=DEV, dtype=torch.int32)).all(), "window must be kept"
        # the REAL reference at this size (float32 mode and bf16).  `sc` came through the DEFAULT path: for one batch element the window
        # projection runs in the library (qproj.hip) -- on the structured case (full_snapkv128k_B: per-channel key
        # scales, heavy sink rows) a rounding flip of a query matters most.  Both projections must meet the reference's 1e-3.
        assert _native().qproj_rope_supported(att, hidden, 64) and _native().USE_LIBRARY_QPROJ
        saved = _native().USE_LIBRARY_QPROJ
        try:
            _native().USE_LIBRARY_QPROJ = False
            sc_gemm = press.score(att, hidden, keys, values, None, kwargs)
        finally:
            _native().USE_LIBRARY_QPROJ = saved
        w_gemm, d_gemm = F.check_against_reference(fx, sc_gemm, _native().topk_select(sc_gemm, n))
        print(f"{name} (model GEMM projection) vs reference: max rel err {w_gemm:.2e}, {d_gemm} set differences (in band)")
        worst, differ = F.check_against_reference(fx, sc, idx)
        overlap, _ = F.check_against_native(fx, idx, S, NATIVE_ULPS, NATIVE_OVERLAP[name] - 0.01)
        print(f"{name} vs reference: max rel err {worst:.2e}, {differ} set differences (in band), overlap with bf16 reference {overlap:.4f}")
        # the fused one-call compress keeps exactly the selected rows
        e = idx.long().unsqueeze(-1).expand(-1, -1, -1, D)
        assert torch.equal(ko, keys.gather(2, e)) and torch.equal(vo, values.gather(2, e))


def test_config5_shards_keep_what_the_batched_reference_keeps():
    """BASELINE config 5 (batch sharded over the GPUs) pinned by the reference: tests/golden/full_snapkv128k_B2.npz is ONE
    float32 run of NVIDIA/kvpress over a batch of two elements whose score maxima differ by ~3000x, with ITS pad constant (the maximum
    over the whole batch + 1, snapkv_press.py:103).  (a) every element on its own, as a shard of `bench.py --gpus N` sees it (pad = its own
    maximum + 1), and (b) the whole batch in one call (pad = the batch maximum + 1, the reference's value): scores within 1e-3 of the
    reference's, retained sets the reference's up to the tolerance band, and (a) == (b) up to float32 summation order."""
    import os

    import _fullsize as F
    import kvpress_amd as P

    name = "full_snapkv128k_B2"
    spec = F.BATCH_CASES[name]
    fxa = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
    S, Bn = spec["S"], len(spec["elements"])
    n = int(S * 0.5)
    kv = [F.make_kv(F.element_spec(spec, b)) for b in range(Bn)]
    keys, values = torch.cat([k for k, _ in kv]).to(DEV), torch.cat([v for _, v in kv]).to(DEV)
    hidden = torch.cat([F.make_hidden(F.element_spec(spec, b)) for b in range(Bn)]).to(DEV)
    att, rot = llama_module()
    press = P.SnapKVPress(0.5)
    with torch.no_grad():
        pe = rot(hidden[:1], torch.arange(S, device=DEV)[None])
        kwargs = {"position_embeddings": pe}
        scB = press.score(att, hidden, keys, values, None, kwargs)
        idxB = _native().topk_select(scB, n)
        koB, voB = press.compress(att, hidden, keys, values, None, kwargs)
        pad = float(fxa["pad_value"])
        assert abs(float(scB[..., -64:].max()) - pad) <= 1e-3 * pad and (scB[..., -64:] == scB[..., -64:].max()).all(), "batched call: the reference's pad constant"
        for b in range(Bn):
            fx = {k[:-len(f"__b{b}")]: fxa[k] for k in fxa.files if k.endswith(f"__b{b}")}
            sc = press.score(att, hidden[b:b + 1], keys[b:b + 1], values[b:b + 1], None, kwargs)
            idx = _native().topk_select(sc, n)
            worst, differ = F.check_against_reference(fx, sc, idx)                       # (a) the shard
            worstB, differB = F.check_against_reference(fx, scB[b:b + 1], idxB[b:b + 1])  # (b) the batch
            own = sc[..., :-64].max() + 1
            assert (sc[..., -64:] == own).all() and float(own) <= pad * (1 + 1e-3)
            # shard and batch split a head's keys over a different number of workgroups (one launch covers B * H_kv heads), so their
            # float32 partial sums are merged in a different order: the same scores to ~1e-5 and the same set up to near-ties at
            # the threshold -- each of which the reference check above has already confined to its tolerance band
            d = ((sc[..., :-64] - scB[b:b + 1, :, :-64]).abs() / scB[b:b + 1, :, :-64].abs().clamp_min(1e-30)).max()
            assert float(d) <= 1e-4, f"shard vs batch scores differ by {float(d):.2e}"   # (each is within 1e-3 of the reference)
            kept, keptB = torch.zeros((H_KV, S), dtype=torch.bool, device=DEV), torch.zeros((H_KV, S), dtype=torch.bool, device=DEV)
            kept.scatter_(1, idx[0].long(), True)
            keptB.scatter_(1, idxB[b].long(), True)
            ndiff = int((kept != keptB).sum())
            assert ndiff <= differ + differB + 64, f"shard and batch keep different sets beyond their in-band differences with the reference: {ndiff}"
            ko, vo = press.compress(att, hidden[b:b + 1], keys[b:b + 1], values[b:b + 1], None, kwargs)
            e = idx.long().unsqueeze(-1).expand(-1, -1, -1, D)
            assert torch.equal(ko, keys[b:b + 1].gather(2, e)) and torch.equal(vo, values[b:b + 1].gather(2, e))
            print(f"{name}[{b}] vs the batched reference run: max rel err {worst:.2e} (shard) / {worstB:.2e} (batch), {differ} / {differB} set differences "
                  f"with the reference (in band), shard vs batch: scores {float(d):.1e}, {ndiff} positions")


def test_config1_opt125m_plumbing():
    """BASELINE config 1: OPT-125m geometry (12 heads, D=64, fp32, 2k tokens), KnormPress(0.5) under the
    hook; [1,12,2048,64] -> [1,12,1024,64].  (The reference's __call__ cannot attach to OPT.)"""
    from transformers import DynamicCache, OPTConfig, OPTForCausalLM

    import kvpress_amd as P
    from oracle import kvpress_oracle as O

    torch.manual_seed(0)
    model = OPTForCausalLM(OPTConfig()).eval().to(DEV)
    ids = torch.randint(0, 1000, (1, 2048), generator=torch.Generator().manual_seed(0)).to(DEV)
    full = DynamicCache()
    with torch.no_grad():
        model(ids, past_key_values=full)
    k_full, v_full = full.layers[0].keys, full.layers[0].values
    assert k_full.shape == (1, 12, 2048, 64) and k_full.dtype == torch.float32
    cache = DynamicCache()
    with torch.no_grad(), P.KnormPress(0.5)(model):
        model(ids, past_key_values=cache)
    for layer in cache.layers:
        assert layer.keys.shape == (1, 12, 1024, 64)
    sc = O.knorm_score(k_full.cpu().numpy())
    got_sc = _native().rownorm_score(k_full, -1.0).cpu().numpy()
    np.testing.assert_allclose(got_sc, sc, rtol=1e-5)
    ko, vo, idx = O.compress(got_sc, k_full.cpu().numpy(), v_full.cpu().numpy(), 0.5)
    assert np.array_equal(cache.layers[0].keys.cpu().numpy(), ko)
    assert np.array_equal(cache.layers[0].values.cpu().numpy(), vo)


def test_llama_hook_end_to_end_gpu():
    """Hook path on the GPU with a small random Llama (bf16): every layer compressed to int(S*(1-r)),
    decoding afterwards appends without re-compressing."""
    from transformers import DynamicCache, LlamaConfig, LlamaForCausalLM

    import kvpress_amd as P

    cfg = LlamaConfig(hidden_size=512, num_attention_heads=4, num_key_value_heads=1, head_dim=128, num_hidden_layers=2,
                      intermediate_size=256, vocab_size=128, max_position_embeddings=4096)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).eval().to(DEV, torch.bfloat16)
    ids = torch.randint(0, 128, (2, 700), device=DEV)
    for press, n in [(P.SnapKVPress(0.5), 350), (P.KnormPress(0.3), 489), (P.ExpectedAttentionPress(0.7), 210)]:
        cache = DynamicCache()
        with torch.no_grad(), press(model):
            model(ids, past_key_values=cache)
            assert all(l.keys.shape == (2, 1, n, 128) for l in cache.layers)
            model(ids[:, :1], past_key_values=cache)
            assert all(l.keys.shape == (2, 1, n + 1, 128) for l in cache.layers)


def test_config4_expected_attention_128k():
    """BASELINE config 4: ExpectedAttentionPress(0.7), 128k, Llama-3.1-8B geometry, against a float64
    restatement of expected_attention_press.py:62-165 with torch ops on the GPU."""
    import kvpress_amd as P

    import _fullsize as F

    spec, keys, values, hidden, fx = full_case("full_ea128k")
    S, n_sink, nfut = spec["S"], 4, 512
    att, rot = llama_module()
    press = P.ExpectedAttentionPress(0.7)
    with torch.no_grad():
        sc = press.score(att, hidden, keys, values, None, {})
        # ---- float64 reference -------------------------------------------------------------------
        q = att.q_proj(hidden[:, n_sink:]).view(1, S - n_sink, H_Q, D).transpose(1, 2).double()
        mu = q.mean(dim=2)
        c = q - mu.unsqueeze(2)
        cov = torch.matmul(c.transpose(2, 3), c) / (S - n_sink)
        del c, q
        pos = torch.arange(S, S + nfut, device=DEV)[None]
        cos, sin = rot(torch.zeros(1, device=DEV, dtype=torch.float32), pos)
        cos, sin = cos[0].double(), sin[0].double()
        Pm = torch.zeros((D, D), device=DEV, dtype=torch.float64)
        Pm[D // 2:, : D // 2] = torch.eye(D // 2, device=DEV, dtype=torch.float64)
        Pm[: D // 2, D // 2:] = -torch.eye(D // 2, device=DEV, dtype=torch.float64)
        R = (cos.unsqueeze(1) * torch.eye(D, device=DEV, dtype=torch.float64) + sin.unsqueeze(1) * Pm).mean(0)
        mu = mu @ R.T
        cov = R @ cov @ R.T
        k = keys[:, :, n_sink:].double()
        ref = torch.empty((1, H_KV, S - n_sink), device=DEV, dtype=torch.float64)
        for h in range(H_KV):
            kh = k[0, h]                                                       # [S', D]
            acc = 0
            for gq in range(H_Q // H_KV):
                hq = h * (H_Q // H_KV) + gq
                lg = kh @ mu[0, hq] / D ** 0.5 + ((kh @ cov[0, hq]) * kh).sum(-1) / D / 2
                acc = acc + torch.softmax(lg, dim=-1)
            ref[0, h] = acc / (H_Q // H_KV) * values[0, h, n_sink:].double().norm(dim=-1)
        got = sc[..., n_sink:].double()
        rel = (got - ref).abs() / ref.abs().clamp_min(1e-300)
        assert rel.max() <= 1e-3, f"max rel err {rel.max().item():.3e}"
        assert (sc[..., :n_sink] == sc[..., n_sink:].max() + 1).all()
        n = int(S * (1 - 0.7))
        assert n == 39321
        ko, vo = press.compress(att, hidden, keys, values, None, {})
        assert ko.shape == (1, H_KV, n, D)
        check_topk_and_gather(sc, keys, values, n, ko, vo)
        idx = _native().topk_select(sc, n)
        worst, differ = F.check_against_reference(fx, sc, idx)
        overlap, _ = F.check_against_native(fx, idx, S, NATIVE_ULPS, NATIVE_OVERLAP["full_ea128k"] - 0.01)
        print(f"ea128k vs reference: max rel err {worst:.2e}, {differ} set differences (in band), overlap with bf16 reference {overlap:.4f}")


def test_config4_on_the_bench_tensors():
    """VERDICT r4 #4b: bench.py times ExpectedAttention on SURVEY §8(d)'s flat set A (seed 104); tests/golden/full_ea128k_A.npz holds
    the REAL reference's outputs for exactly those tensors, so `extra.ea128k.parity` of the driver's line is a real check.  Here:
    the same fixture through the press (scores within 1e-3, set parity outside the band) and through bench.fixture_parity."""
    import _fullsize as F
    import bench
    import kvpress_amd as P

    spec, keys, values, hidden, fx = full_case("full_ea128k_A")
    S, n = spec["S"], 39321
    att, rot = llama_module()
    press = P.ExpectedAttentionPress(0.7)
    with torch.no_grad():
        sc = press.score(att, hidden, keys, values, None, {})
        idx = _native().topk_select(sc, n)
        worst, differ = F.check_against_reference(fx, sc, idx)
        overlap, _ = F.check_against_native(fx, idx, S, NATIVE_ULPS, 0.9771 - 0.01)
        print(f"ea128k (set A) vs reference: max rel err {worst:.2e}, {differ} set differences (in band), overlap with bf16 reference {overlap:.4f}")
        assert bench.FIXTURES["ea128k"] == "full_ea128k_A" and bench.SEEDS["ea128k"] == spec["seed"]
        k2, v2, h2, _ = bench.bench_inputs("ea128k", 0, torch.device(DEV))
        assert torch.equal(k2, keys) and torch.equal(v2, values) and torch.equal(h2, hidden), "bench.py must time the fixture's tensors"
        par = bench.fixture_parity("ea128k", press, att, hidden, keys, values, {}, n)
    assert par and par["ok"], par


def test_ea_qstats_128k_large_mean_adversarial():
    """query statistics at 128k tokens with |mean| ~ 10 sigma in a few dominant channels (the regime where raw second
    moments cancel): mean and covariance against float64, and the FINAL scores through kvp_ea_score within 1e-3 of the float64
    chain fed with the float64 statistics."""
    S, Sk = 131072, 16384
    g = torch.Generator(device=DEV)
    g.manual_seed(66)
    sig = torch.exp(0.5 * torch.randn((1, 1, H_Q * D), generator=g, device=DEV))
    mean = torch.randn((1, 1, H_Q * D), generator=g, device=DEV) * sig
    dom = torch.arange(H_Q * D, device=DEV) % 17 == 3                      # a few dominant channels per head
    sig[..., dom] *= 6.0
    mean[..., dom] = 10.0 * sig[..., dom] * torch.sign(mean[..., dom])     # |mean| = 10 sigma there
    q = (torch.randn((1, S, H_Q * D), generator=g, device=DEV) * sig + mean).to(torch.bfloat16)
    qt = q.view(1, S, H_Q, D).transpose(1, 2)                              # the layout q_proj produces
    mu, cov = _native().ea_qstats(qt, True)
    mu_r = torch.empty((1, H_Q, D), dtype=torch.float64, device=DEV)
    cov_r = torch.empty((1, H_Q, D, D), dtype=torch.float64, device=DEV)
    for h in range(H_Q):
        x = qt[0, h].double()
        mu_r[0, h] = x.mean(0)
        xc = x - mu_r[0, h]
        cov_r[0, h] = xc.T @ xc / S
    assert ((mu.double() - mu_r).abs() <= 1e-5 * mu_r.abs().amax() + 1e-6).all()
    d = cov_r.diagonal(dim1=-2, dim2=-1).sqrt()
    err = (cov.double() - cov_r).abs() / (d.unsqueeze(-1) * d.unsqueeze(-2))
    assert err.max() <= 1e-3, f"covariance error {err.max().item():.2e} sigma_i sigma_j"
    # final scores: float64 chain (expected_attention_press.py:148-160 without the averaged RoPE) with the float64 statistics
    keys = (torch.randn((1, H_KV, Sk, D), generator=g, device=DEV) * 0.3).to(torch.bfloat16)
    values = torch.randn((1, H_KV, Sk, D), generator=g, device=DEV).to(torch.bfloat16)
    sc = _native().ea_score(keys, values, mu, cov, 4, True, 0.0)
    G = H_Q // H_KV
    ref = torch.empty((1, H_KV, Sk - 4), dtype=torch.float64, device=DEV)
    for h in range(H_KV):
        kh = keys[0, h, 4:].double()
        acc = 0
        for gq in range(G):
            hq = h * G + gq
            lg = kh @ mu_r[0, hq] / D ** 0.5 + ((kh @ cov_r[0, hq]) * kh).sum(-1) / D / 2
            acc = acc + torch.softmax(lg, dim=-1)
        ref[0, h] = acc / G * values[0, h, 4:].double().norm(dim=-1)
    rel = (sc[..., 4:].double() - ref).abs() / ref.abs().clamp_min(1e-300)
    assert rel.max() <= 1e-3, f"final scores differ by {rel.max().item():.2e}"
    print(f"ea qstats 128k adversarial: cov err {err.max().item():.2e} sigma_i sigma_j, final score err {rel.max().item():.2e}")


@pytest.mark.parametrize("Dn,layout", [(64, "q_proj"), (64, "contiguous"), (96, "q_proj"), (256, "q_proj")])
def test_ea_narrow_heads_at_size(Dn, layout):
    """Round 6: ExpectedAttention for head sizes 64 / 96 at a size where the statistics stream Q with non-temporal loads (> 192 MiB of rows) and
    the logits run several chunks per head: statistics against float64 (pairs of neighbouring heads for D = 64 in the projection's layout,
    zero-padded heads of 128 otherwise) and the final scores through kvp_ea_score within 1e-3 of the float64 chain."""
    S, Sk, Hq, Hkv = 40960, 20000, 32, 8
    g = torch.Generator(device=DEV)
    g.manual_seed(600 + Dn)
    sig = torch.exp(0.4 * torch.randn((1, 1, Hq * Dn), generator=g, device=DEV))
    mean = torch.randn((1, 1, Hq * Dn), generator=g, device=DEV) * sig * 2.0
    q = (torch.randn((1, S, Hq * Dn), generator=g, device=DEV) * sig + mean).to(torch.bfloat16)
    qt = q.view(1, S, Hq, Dn).transpose(1, 2)
    if layout == "contiguous":
        qt = qt.contiguous()
    mu, cov = _native().ea_qstats(qt, True)
    mu_r = torch.empty((1, Hq, Dn), dtype=torch.float64, device=DEV)
    cov_r = torch.empty((1, Hq, Dn, Dn), dtype=torch.float64, device=DEV)
    for h in range(Hq):
        x = qt[0, h].double()
        mu_r[0, h] = x.mean(0)
        xc = x - mu_r[0, h]
        cov_r[0, h] = xc.T @ xc / S
    assert ((mu.double() - mu_r).abs() <= 1e-5 * mu_r.abs().amax() + 1e-6).all()
    d = cov_r.diagonal(dim1=-2, dim2=-1).sqrt()
    err = (cov.double() - cov_r).abs() / (d.unsqueeze(-1) * d.unsqueeze(-2))
    assert err.max() <= 1e-3, f"covariance error {err.max().item():.2e} sigma_i sigma_j"
    keys = (torch.randn((1, Hkv, Sk, Dn), generator=g, device=DEV) * 0.3).to(torch.bfloat16)
    values = torch.randn((1, Hkv, Sk, Dn), generator=g, device=DEV).to(torch.bfloat16)
    sc = _native().ea_score(keys, values, mu, cov, 4, True, 0.0)
    G = Hq // Hkv
    ref = torch.empty((1, Hkv, Sk - 4), dtype=torch.float64, device=DEV)
    for h in range(Hkv):
        kh = keys[0, h, 4:].double()
        acc = 0
        for gq in range(G):
            hq = h * G + gq
            lg = kh @ mu_r[0, hq] / Dn ** 0.5 + ((kh @ cov_r[0, hq]) * kh).sum(-1) / Dn / 2
            acc = acc + torch.softmax(lg, dim=-1)
        ref[0, h] = acc / G * values[0, h, 4:].double().norm(dim=-1)
    rel = (sc[..., 4:].double() - ref).abs() / ref.abs().clamp_min(1e-300)
    assert rel.max() <= 1e-3, f"final scores differ by {rel.max().item():.2e}"
    print(f"ea D={Dn} {layout}: cov err {err.max().item():.2e} sigma_i sigma_j, final score err {rel.max().item():.2e}")


@pytest.mark.parametrize("Dn", [64, 96, 256])
def test_expected_attention_press_on_narrow_head_models(Dn):
    """ExpectedAttentionPress.score on an attention module with 64- / 96-dimensional heads (Llama-3.2-1B / Phi-3-mini class; round 6: their statistics
    and quadratic form run on the matrix cores) against a float64 restatement of expected_attention_press.py:62-165 fed with the module's own bf16
    queries: the projection's layout, the averaged RoPE on a D x D covariance, the GQA mean, the sink pad."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaAttention, LlamaRotaryEmbedding

    import kvpress_amd as P

    Hq, Hkv, S, n_sink, nfut = 8, 2, 9000, 4, 512
    cfg = LlamaConfig(hidden_size=512, num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=Dn, num_hidden_layers=1, intermediate_size=256,
                      vocab_size=128, max_position_embeddings=16384, rope_theta=10000.0)
    torch.manual_seed(Dn)
    att = LlamaAttention(cfg, layer_idx=0).eval()
    with torch.no_grad():
        att.q_proj.weight.mul_(3.0)   # queries of order 1 (the default initializer_range gives logits ~0: nothing to tell apart)
    att = att.to(DEV, torch.bfloat16)
    rot = LlamaRotaryEmbedding(cfg).to(DEV)
    att.rotary_emb = rot
    g = torch.Generator(device=DEV)
    g.manual_seed(Dn + 1)
    hidden = (torch.randn((1, S, 512), generator=g, device=DEV) + 0.3).to(torch.bfloat16)
    keys = (torch.randn((1, Hkv, S, Dn), generator=g, device=DEV) * 0.8).to(torch.bfloat16)
    values = torch.randn((1, Hkv, S, Dn), generator=g, device=DEV).to(torch.bfloat16)
    press = P.ExpectedAttentionPress(0.5)
    with torch.no_grad():
        sc = press.score(att, hidden, keys, values, None, {})
        q = att.q_proj(hidden[:, n_sink:]).view(1, S - n_sink, Hq, Dn).transpose(1, 2).double()
        mu = q.mean(dim=2)
        c = q - mu.unsqueeze(2)
        cov = torch.matmul(c.transpose(2, 3), c) / (S - n_sink)
        pos = torch.arange(S, S + nfut, device=DEV)[None]
        cos, sin = rot(torch.zeros(1, device=DEV, dtype=torch.float32), pos)
        cos, sin = cos[0].double(), sin[0].double()
        Pm = torch.zeros((Dn, Dn), device=DEV, dtype=torch.float64)
        Pm[Dn // 2:, : Dn // 2] = torch.eye(Dn // 2, device=DEV, dtype=torch.float64)
        Pm[: Dn // 2, Dn // 2:] = -torch.eye(Dn // 2, device=DEV, dtype=torch.float64)
        R = (cos.unsqueeze(1) * torch.eye(Dn, device=DEV, dtype=torch.float64) + sin.unsqueeze(1) * Pm).mean(0)
        mu = mu @ R.T
        cov = R @ cov @ R.T
        ref = torch.empty((1, Hkv, S - n_sink), device=DEV, dtype=torch.float64)
        for h in range(Hkv):
            kh = keys[0, h, n_sink:].double()
            acc = 0
            for gq in range(Hq // Hkv):
                hq = h * (Hq // Hkv) + gq
                lg = kh @ mu[0, hq] / Dn ** 0.5 + ((kh @ cov[0, hq]) * kh).sum(-1) / Dn / 2
                acc = acc + torch.softmax(lg, dim=-1)
            ref[0, h] = acc / (Hq // Hkv) * values[0, h, n_sink:].double().norm(dim=-1)
    got = sc[..., n_sink:].double()
    rel = (got - ref).abs() / ref.abs().clamp_min(1e-300)
    assert rel.max() <= 1e-3, f"D={Dn}: scores differ by {rel.max().item():.2e}"
    assert (sc[..., :n_sink] > sc[..., n_sink:].max()).all()
    spread = (ref.max() / ref.min()).item()
    assert spread > 3.0, f"the case does not tell keys apart (max / min score {spread:.2f})"
    print(f"EA press D={Dn}: max rel err {rel.max().item():.2e}, score spread {spread:.1f}")


def test_f_rows_at_128k_properties():
    """The §8(f) kernels at the BASELINE size (8 x 131072 x 128 bf16: the slot walks, the streaming loads and the one-pass gather +
    re-rotation are what runs there), through size-independent properties: the row norms against torch in float64, the default walk ==
    the interleaved walk bit for bit (row norms, CUR) / within 2e-6 (KeyDiff: the anchor is summed in another order), KeyDiff against
    its float64 definition on a subsample of positions, the one-pass gather + re-rotation == the two kernels it replaces, a rotation
    preserves every (d, d + D/2) pair's norm, and positions already in place (delta 0) come out untouched."""
    nat = _native()
    S = 131072
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    k = torch.randn((1, H_KV, S, D), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, H_KV, S, D), generator=g, device=DEV).to(torch.bfloat16)
    rn = nat.rownorm_score(k, -1.0)
    assert torch.allclose(rn.double(), -k.double().norm(dim=-1), rtol=1e-6, atol=0)
    cur = nat.cur_score(k, v, "kv_product", 16, 4)
    kd = nat.keydiff_score(k)
    # the slot walk of the long rows and the interleaved walk of a 4000-token view: the same bits per row
    assert torch.equal(nat.rownorm_score(k[:, :, :4000], -1.0), rn[..., :4000])
    kn = torch.nn.functional.normalize(k.double(), dim=-1)
    anchor = kn.mean(dim=2, keepdim=True)
    sub = torch.arange(0, S, 97, device=DEV)
    want = -torch.nn.functional.cosine_similarity(k[:, :, sub].double(), anchor, dim=-1)
    assert (kd[:, :, sub].double() - want).abs().max() <= 2e-6
    assert abs(float(cur[:, :, 4:].sum(-1).mean()) - 1.0) < 1e-3 and bool((cur[:, :, :4] == 1.0).all())   # normalised rows, sinks = 1
    # gather + re-rotation: keep every second position, the first 1000 in place (delta 0)
    n = S // 2
    pos = torch.cat([torch.arange(1000, device=DEV), 1000 + 2 * torch.arange(n - 1000, device=DEV) + 1]).to(torch.int32)
    pos = pos.clamp_(max=S - 1).expand(1, H_KV, n).contiguous()
    inv = 500000.0 ** (-torch.arange(0, D, 2, device=DEV, dtype=torch.float32) / D)
    k1, v1 = nat.gather_kv_rerotate(k, v, pos, inv)
    k2, v2 = nat.gather_kv(k, v, pos)
    assert torch.equal(v1, v2) and torch.equal(k1[:, :, :1000], k2[:, :, :1000])
    nat.rerotate_keys_(k2, pos, inv)
    assert torch.equal(k1, k2)
    src = k[:, :, pos[0, 0].long()].float()
    pair = lambda x: (x[..., : D // 2] ** 2 + x[..., D // 2:] ** 2).sqrt()
    assert ((pair(k1.float()) - pair(src)).abs() <= 2.0 ** -6 * pair(src) + 1e-6).all()


def test_bench_measures_its_traffic_live():
    """bench.py's roofline.traffic: with rocprofv3 on the box the number comes from two --pmc passes spawned by
    the run itself (FETCH_SIZE, WRITE_SIZE; kernel trace only), not from a committed file.  Knorm 32k: the dominant kernel is the cluster
    select with the norm stream inside, algorithmic bytes = K once (67 MB); the counters must land within [0.9, 1.5] x of that."""
    import json
    import os
    import shutil
    import subprocess
    import sys

    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KVP_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "knorm32k", "--steps", "20", "--warmup", "5", "--prewarm-ms", "5",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    roof = line["roofline"]
    if not roof["traffic_source"].startswith("live:"):   # the profiler did not run here (permissions, another profiler attached, ...): the
        pytest.skip(f"no live PMC pass on this box: {roof['traffic_source']}")   # fallback path is covered by tests/test_dist_gloo.py
    assert 0.9 <= roof["traffic"] / roof["algorithmic_bytes_per_launch"] <= 1.5, (roof["traffic"], roof["algorithmic_bytes_per_launch"], roof["kernel"])


def test_bench_two_ranks_real_kernels_on_one_gpu():
    """The N > 1 path of bench.py with the REAL kernels: `python bench.py --gpus 2` launches two ranks (one process per
    "GPU"); KVP_BENCH_SHARE_GPU=1 lets both use GPU 0 of this one-GPU box (gloo for the timing reduction: RCCL refuses two ranks on one
    device).  Real sharding (one batch element per rank), barrier-bracketed timing, MAX over ranks, one JSON line with n_gpus = 2; the
    two ranks share one GPU, so the aggregate rate is about the N = 1 rate."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVP_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "knorm32k", "--steps", "5",
                        "--warmup", "2", "--prewarm-ms", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 1
    assert 0.02 < line["ms_per_step"] < 5.0, line["ms_per_step"]
    assert abs(line["value"] - 2 * 32768 / (32 * line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    assert line["roofline"] is not None and line["roofline"]["path_frac"] > 0


def test_bench_two_ranks_rccl():
    """The RCCL leg itself: `python bench.py --gpus 2 --backend nccl` -- one process per GPU, init_process_group("nccl",
    device_id=...), barriers and the MAX all-reduce over RCCL.  Needs two GPUs: skipped on the one-GPU boxes, executed by the first
    multi-GPU box that runs this suite."""
    import json
    import os
    import subprocess
    import sys

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KVP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "nccl", "--workload", "knorm32k", "--steps", "3",
                        "--warmup", "1", "--prewarm-ms", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 1
    assert abs(line["value"] - 2 * 32768 / (32 * line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
