"""Host-side model of the two-hop cluster select (kvpress_amd/csrc/topk_cluster.hip, round 5), step for step in numpy: sample -> bracket ->
window digit -> per-slot candidate records -> local rounds -> ordered compaction.  CPU only: it pins the ALGORITHM (any monotone
binning of the keys keeps the radix select exact; the tie rule survives the candidate records) against a plain sort, for the row
kinds the GPU tests run through the kernel (tests/_fault_child.py `paths`), and documents when the form declines ("miss": the
threshold lies outside the sample's bracket; "overflow": a slot has more candidates than its record holds)."""
import numpy as np
import pytest

TC_NS, TC_WB, TC_REC, SLOTS, THREADS = 128, 256, 128, 32, 1024   # topk_cluster.hip / topk_internal.h


def f2key(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def two_hop(scores, k):
    S = scores.size
    per = -(-S // (SLOTS * THREADS))
    per = 1 if per <= 1 else 2 if per <= 2 else 4 if per <= 4 else 8
    L = THREADS * per
    keys = np.maximum(f2key(scores), 1).astype(np.uint64)
    stride = S // TC_NS
    smp = keys[np.minimum(np.arange(TC_NS) * stride + stride // 2, S - 1)]
    order = np.lexsort((np.arange(TC_NS), -smp.astype(np.int64)))   # descending, ties by sample index
    pq = np.float32(k) / np.float32(S)
    rstar = min(int(pq * np.float32(TC_NS)), TC_NS - 1)
    delta = int(np.float32(4.5) * np.sqrt(np.float32(TC_NS) * pq * (np.float32(1) - pq))) + 3
    Hk, Lk = int(smp[order[max(rstar - delta, 0)]]), int(smp[order[min(rstar + delta, TC_NS - 1)]])
    s = 0
    while ((Hk >> s) - (Lk >> s)) > TC_WB - 3:
        s += 1
    base = Lk >> s
    v = keys >> np.uint64(s)
    b = np.where(v < base, 0, np.minimum(v - base + 1, TC_WB - 1)).astype(int)
    hist = np.bincount(b, minlength=TC_WB)
    cum, d1, k1 = 0, None, None
    for d in range(TC_WB - 1, -1, -1):
        if cum < k <= cum + hist[d]:
            d1, k1 = d, k - cum
            break
        cum += hist[d]
    if not 1 <= d1 <= TC_WB - 2:
        return None, "miss"
    recs = []
    for slot in range(SLOTS):
        lo, hi = slot * L, min((slot + 1) * L, S)
        cand = keys[lo:hi][b[lo:hi] == d1]
        if cand.size > TC_REC - 2:
            return None, "overflow"
        recs.append((cand, int((b[lo:hi] > d1).sum())))
    allc = np.concatenate([c for c, _ in recs])
    T, krem, hi_ = (base + d1 - 1) << s, k1, s
    while hi_ > 0:   # local rounds of <= 12 bits below the bin's bits
        wd = min(hi_, 12)
        lo_ = hi_ - wd
        m = allc[(allc >> np.uint64(hi_)) == (T >> hi_)]
        h = np.bincount(((m >> np.uint64(lo_)) & np.uint64((1 << wd) - 1)).astype(int), minlength=1 << wd)
        cum = 0
        for d in range((1 << wd) - 1, -1, -1):
            if cum < krem <= cum + h[d]:
                T |= d << lo_
                krem -= cum
                break
            cum += h[d]
        hi_ = lo_
    quota, out, gt_before, eq_before = krem, [], 0, 0
    for slot in range(SLOTS):   # every slot needs only the records of the slots before it
        lo, hi = slot * L, min((slot + 1) * L, S)
        kk = keys[lo:hi]
        eq_rank = eq_before + np.cumsum(kk == T) - (kk == T)
        keep = (kk > T) | ((kk == T) & (eq_rank < quota))
        out.append(lo + np.nonzero(keep)[0])
        cand, ngt = recs[slot]
        assert ngt + int((cand > T).sum()) == int((kk > T).sum()) and int((cand == T).sum()) == int((kk == T).sum())
        gt_before += ngt + int((cand > T).sum())
        eq_before += int((cand == T).sum())
    return np.concatenate(out), "ok"


def ref(scores, k):
    keys = np.maximum(f2key(scores), 1).astype(np.int64)
    return np.sort(np.lexsort((np.arange(scores.size), -keys))[:k])


def _rows(S, rs):
    flat = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal(S))).astype(np.float32)
    yield "flat", flat, "ok"
    yield "knorm", -np.sqrt((rs.standard_normal((S, 32)) ** 2).sum(1)).astype(np.float32), "ok"
    yield "consecutive keys", (np.float32(1.0).view(np.uint32) + rs.randint(0, 200, size=S).astype(np.uint32)).view(np.float32), "ok"
    yield "all exponents", (np.exp(12 * rs.standard_normal(S)) * rs.choice([-1.0, 1.0], size=S)).astype(np.float32), "ok"
    x = flat.copy()
    x[::97] = np.inf
    x[5::89] = -np.inf
    yield "infinities", x, "ok"
    yield "20 distinct values", rs.randint(0, 20, size=S).astype(np.float32), "overflow"
    yield "sorted", np.sort(flat), "overflow"
    yield "constant", np.full(S, 0.25, np.float32), "overflow"


@pytest.mark.parametrize("S", [20000, 65536, 131008])
def test_two_hop_model_equals_sort(S):
    rs = np.random.RandomState(S)
    for name, x, want_why in _rows(S, rs):
        for frac in (0.5, 0.1, 0.9):
            k = max(1, int(S * frac))
            got, why = two_hop(x, k)
            if want_why != "ok":   # (a short row's slots may still hold such a row's candidates: only the long ones must decline)
                assert why != "ok" or (S < 131008 and name != "constant"), f"{name} S={S} k={k}: expected the form to decline"
            elif frac == 0.5:
                assert why == "ok", f"{name} S={S} k={k}: {why}"
            if got is not None:
                assert np.array_equal(got, ref(x, k)), f"{name} S={S} k={k}"


def test_two_hop_model_ties_at_the_threshold_and_extremes():
    rs = np.random.RandomState(5)
    S, k = 131008, 65504
    x = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal(S))).astype(np.float32)
    order = np.argsort(-x, kind="stable")
    x[rs.choice(order[k + 100:], 48, replace=False)] = x[order[k - 1]]   # 49 keys equal to the threshold, spread over the slots
    got, why = two_hop(x, k)
    assert why == "ok" and np.array_equal(got, ref(x, k))
    for kk in (1, S - 1):   # the sample cannot bracket an extreme reliably: either form is fine, a declined one says so
        got, why = two_hop(x, kk)
        assert why in ("ok", "miss")
        if got is not None:
            assert np.array_equal(got, ref(x, kk))
