"""Numpy models of the decompositions the round-6 kernels rest on (CPU, milliseconds): each kernel-side trick restated in ten lines and checked
against the plain formula of the reference it replaces.  The GPU tests check the kernels; these pin the ARGUMENTS their comments make.

  * ExpectedAttention statistics for narrow / wide heads on the 128-wide syrk (csrc/ea_mfma.hip, QstatArgs::nch / quarters; reference:
    expected_attention_press.py:74-80): pairs of neighbouring 64-dim heads, zero-padded 96-dim heads, the six pairs of quarters of a 256-dim head;
  * the quadratic form on the doubled upper triangle, its hi + lo split and the strip / k-step bookkeeping (expected_attention_press.py:151);
  * SnapKV's causal rule in padded window coordinates, including sequences shorter than the padded window (snapkv_press.py:63-65);
  * pass 2's tile ranges from pass 1's times: monotone, complete, neighbours agree (snapkv_mfma.hip: snapkv_p2_asm's prologue).
"""
import numpy as np

QA = [0, 2, 0, 1, 0, 1]   # EQ_QA / EQ_QB of csrc/ea_mfma.hip: the six pairs of 64-dimension quarters
QB = [1, 3, 2, 3, 3, 2]


def _cov(x):
    c = x - x.mean(0)
    return c.T @ c / x.shape[0]


def test_pairs_of_64_dim_heads_on_a_128_wide_syrk():
    rs = np.random.RandomState(0)
    x = rs.standard_normal((500, 2, 64)) * np.array([1.0, 3.0])[None, :, None] + 0.7   # [tokens, two neighbouring heads, 64]
    wide = _cov(x.reshape(500, 128))                                                    # what the kernel computes for the pair
    for h in range(2):
        assert np.allclose(wide[64 * h:64 * h + 64, 64 * h:64 * h + 64], _cov(x[:, h]), rtol=0, atol=1e-12)   # the diagonal blocks ARE the heads' own
    assert np.allclose(x.reshape(500, 128).mean(0), np.concatenate([x[:, 0].mean(0), x[:, 1].mean(0)]))


def test_zero_padded_head_of_96_dimensions():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((300, 96)) + 0.3
    xp = np.concatenate([x, np.zeros((300, 32))], axis=1)   # the LDS tile: upper dimensions zeroed once, never written
    assert np.array_equal(_cov(xp)[:96, :96], _cov(x)) and not _cov(xp)[96:].any() and not _cov(xp)[:, 96:].any()


def test_six_pairs_of_quarters_cover_a_256_dim_head():
    rs = np.random.RandomState(2)
    x = rs.standard_normal((400, 256)) * np.exp(0.3 * rs.standard_normal(256)) + rs.standard_normal(256)
    want, mu_want = _cov(x), x.mean(0)
    got, mu = np.full((256, 256), np.nan), np.full(256, np.nan)
    pairs_seen = set()
    for p in range(6):
        a, b = QA[p], QB[p]
        pairs_seen.add(frozenset((a, b)))
        v = np.concatenate([x[:, 64 * a:64 * a + 64], x[:, 64 * b:64 * b + 64]], axis=1)   # a virtual head of 128 dimensions
        cv, mv = _cov(v), v.mean(0)
        if p < 2:
            mu[128 * p:128 * p + 128] = mv                                                   # pairs 0, 1 hold dimensions 0 .. 127, 128 .. 255 in order
        for qi, Qi in ((0, a), (1, b)):
            for qj, Qj in ((0, a), (1, b)):
                if Qi != Qj or p < 2:                                                        # the combine's owner rule
                    blk = got[64 * Qi:64 * Qi + 64, 64 * Qj:64 * Qj + 64]
                    assert np.isnan(blk).all() or Qi == Qj, "a cross block is written once"
                    got[64 * Qi:64 * Qi + 64, 64 * Qj:64 * Qj + 64] = cv[64 * qi:64 * qi + 64, 64 * qj:64 * qj + 64]
    assert len(pairs_seen) == 6 and not np.isnan(got).any() and not np.isnan(mu).any()
    assert np.allclose(got, want, rtol=0, atol=1e-12) and np.allclose(mu, mu_want)


def test_quadratic_form_on_the_doubled_upper_triangle_is_exact_for_any_matrix():
    rs = np.random.RandomState(3)
    for D in (64, 96, 128, 256):
        C = rs.standard_normal((D, D))                      # NOT symmetric
        U = np.triu(C, 1) + np.tril(C, -1).T + np.diag(np.diag(C))
        k = rs.standard_normal((50, D))
        assert np.allclose(np.einsum("si,ij,sj->s", k, C, k), np.einsum("si,ij,sj->s", k, U, k), rtol=1e-12)
        # strip s (rows 32 s .. 32 s + 31 of U) only has the 16-wide k-steps 2 s .. D / 16 - 1
        nk = D // 16
        for s in range(D // 32):
            assert not U[32 * s:32 * s + 32, :32 * s].any()
        products = [nk - 2 * s for s in range(D // 32)]
        assert sum(products) == {64: 6, 96: 12, 128: 20, 256: 72}[D]
        if D == 256:    # wave w holds strips w and 7 - w: the same work for every wave
            assert {products[w] + products[7 - w] for w in range(4)} == {18}
        if D == 128:    # waves 0 / 1: strips (0, 3), waves 2 / 3: strips (1, 2)
            assert products[0] + products[3] == products[1] + products[2] == 10


def _bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def test_hi_lo_split_of_the_covariance_carries_seventeen_bits():
    rs = np.random.RandomState(4)
    a = rs.standard_normal((128, 128)) * 0.1
    C = (a @ a.T).astype(np.float32)
    hi = _bf16(C)
    lo = _bf16(C - hi)
    k = _bf16(rs.standard_normal((200, 128)))
    want = np.einsum("si,ij,sj->s", k.astype(np.float64), C.astype(np.float64), k.astype(np.float64))
    one = np.einsum("si,ij,sj->s", k.astype(np.float64), hi.astype(np.float64), k.astype(np.float64))
    two = one + np.einsum("si,ij,sj->s", k.astype(np.float64), lo.astype(np.float64), k.astype(np.float64))
    assert np.max(np.abs(two - want) / np.abs(want)) < 2.0 ** -15          # hi + lo: ~2^-17 per entry
    assert np.max(np.abs(one - want) / np.abs(want)) > 10 * np.max(np.abs(two - want) / np.abs(want))   # one bf16 matrix is what the 1e-3 contract cannot afford (LAB R6.5)


def test_causal_rule_in_padded_window_coordinates():
    """reference: attn_weights masked by triu(ones(W, W), diagonal=1) on the last W columns (snapkv_press.py:63-65), i.e. window row r may
    attend key k iff k <= S - W + r.  Kernel: blocks of 64 PADDED rows, padded row p = real row p - (Wp - W), limit mlim + p with
    mlim = S - Wp as a SIGNED number (S < Wp happens: the bug the shape fuzzer found)."""
    for W, S in ((64, 1000), (1, 2), (7, 300), (65, 66), (100, 4200), (200, 253), (130, 150), (10, 40), (257, 40000)):
        Wp = (W + 63) // 64 * 64
        r = np.arange(W)[:, None]
        k = np.arange(S)[None, :]
        ref_visible = k <= S - W + r
        mlim = np.int32(S) - np.int32(Wp)                     # may be negative
        p = np.arange(Wp)[:, None]
        pad_visible = (k.astype(np.int32) <= mlim + p.astype(np.int32))
        assert np.array_equal(pad_visible[Wp - W:], ref_visible), (W, S)
        # a 128-key tile starting at key0 needs no per-element mask iff its last key is visible to padded row 0: key0 + 127 <= mlim.  With S < Wp no tile
        # qualifies (mlim < 0); the first version compared against S - Wp in UNSIGNED arithmetic -- 2^32 - (Wp - S) -- and called every tile unmasked
        # (the per-element form `k > S - Wp + p` wraps back to the right value; the tile classification does not)
        for key0 in range(0, S, 128):
            unmasked = bool((pad_visible[0, key0:min(key0 + 128, S)]).all() and key0 + 127 < S)
            assert (key0 + 127 <= int(mlim)) == unmasked, (W, S, key0)
            if S < Wp:
                assert key0 + 127 <= (S - Wp) % (1 << 32) and not unmasked, (W, S, key0)


def _ranges(times, ntiles):
    """snapkv_p2_asm's prologue: times clamped to [tmin, 2 tmin], speed = 1 / time, inclusive scan, end = round(ntiles * prefix / total)"""
    t = np.maximum(np.asarray(times, np.float32), 1)
    t = np.minimum(t, 2 * t.min())
    incl = np.cumsum((1.0 / t).astype(np.float32), dtype=np.float32)
    end = np.minimum(np.rint(np.float32(ntiles) * (incl / incl[-1])).astype(np.int64), ntiles)
    end[-1] = ntiles
    beg = np.concatenate([[0], end[:-1]])
    return beg, np.maximum(end - beg, 0)


def test_pass2_tile_ranges_from_pass1_times_partition_the_tiles():
    rs = np.random.RandomState(5)
    for nch, ntiles in ((32, 1024), (8, 130), (64, 512), (2, 16), (32, 1023)):
        for _ in range(50):
            times = rs.randint(5000, 8000, size=nch)
            if rs.randint(3) == 0:
                times[rs.randint(nch)] *= 50          # a workgroup held up for a reason of its own
            beg, cnt = _ranges(times, ntiles)
            assert beg[0] == 0 and (beg[1:] == (beg + cnt)[:-1]).all() and beg[-1] + cnt[-1] == ntiles   # contiguous, complete, neighbours agree
            assert (cnt >= 0).all() and cnt.min() >= (ntiles // nch) // 2 - 1                              # the clamp: nobody starves
        beg, cnt = _ranges(np.full(nch, 6000), ntiles)
        assert cnt.max() - cnt.min() <= 1                                                                  # equal times -> equal shares
