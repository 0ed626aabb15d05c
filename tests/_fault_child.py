"""Child process of tests/test_gpu_cluster_failure.py: runs with KVPRESS_HIP_LIB = the fault-injection twin of the library
(kvpress_amd/lib/libkvpress_hip_faultinject.so: topk_cluster.hip compiled with -DKVP_TC_FAULT_INJECTION, kvpress_amd/build.py), in which
KVP_TC_TEST_DELAY_SLOT makes one workgroup of cluster 0 arrive 2 x timeout late.  The product library has no such hook.

    python tests/_fault_child.py barrier | fused | stale | paths   -> prints CHILD_PASS on success, raises otherwise
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from kvpress_amd import _native as n  # noqa: E402
from oracle import kvpress_oracle as O  # noqa: E402

DEV = "cuda:0"


def knobs(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    n.tuning_reload()


def barrier():
    """One workgroup of cluster 0 arrives 2 x timeout late: the others give up.  Row 0's indices are -1 (other rows: untouched clusters,
    correct), the gather turns them into NaN rows, the next library call raises ONCE (32 workgroups reported the same launch), the
    cached clean workspaces are gone, and then everything works again."""
    g = torch.Generator().manual_seed(1)
    sc = torch.randn(8, 131008, generator=g, dtype=torch.float32)
    want = O.topk_select(sc.numpy(), 65472)
    d = sc.to(DEV)
    k = torch.randn(1, 8, 131008, 128, device=DEV, dtype=torch.bfloat16)
    v = torch.randn(1, 8, 131008, 128, device=DEV, dtype=torch.bfloat16)
    assert np.array_equal(n.topk_select(d, 65472).cpu().numpy(), want)   # sanity, no fault
    n.gather_kv(k, v, torch.from_numpy(want).to(DEV).view(1, 8, -1))
    torch.cuda.synchronize()
    n.async_error_check()

    knobs(KVP_TC_TIMEOUT_US=20000, KVP_TC_TEST_DELAY_SLOT=5)
    got = n.topk_select(d, 65472)            # returns KVP_OK: the failure happens on the device, later
    ko, vo = n.gather_kv(k, v, got.view(1, 8, -1))
    torch.cuda.synchronize()
    gg = got.cpu().numpy()
    assert (gg[0] == -1).all(), "row of the cluster that timed out must be poisoned"
    assert np.array_equal(gg[1:], want[1:]), "clusters that did not time out are unaffected"
    assert torch.isnan(ko[0, 0].float()).all() and torch.isnan(vo[0, 0].float()).all(), "poisoned indices must gather NaN rows"
    assert torch.equal(ko[0, 1], k[0, 1][torch.from_numpy(want[1]).long().to(DEV)])
    knobs(KVP_TC_TIMEOUT_US=None, KVP_TC_TEST_DELAY_SLOT=None)
    try:
        n.topk_select(d, 65472)
        raise AssertionError("the call after a failed select must raise")
    except n.KvpressHipError as e:
        assert "cluster select" in str(e), str(e)
    assert not n._TOPK_WS, "a reported failure must drop every cached 'clean' workspace"
    for _ in range(3):                        # ONE report per failed launch: no second KVP_EASYNC from a late store of the same launch
        assert np.array_equal(n.topk_select(d, 65472).cpu().numpy(), want)
        torch.cuda.synchronize()
        n.async_error_check()


def fused():
    """The same through the fused Knorm compress (the cluster kernel computes the norms itself): poisoned rows come out as NaN."""
    g = torch.Generator().manual_seed(3)
    k = torch.randn(1, 8, 32768, 128, generator=g).to(DEV, torch.bfloat16)
    v = torch.randn(1, 8, 32768, 128, generator=g).to(DEV, torch.bfloat16)
    ko_ref, vo_ref = n.knorm_compress(k, v, 16384)
    torch.cuda.synchronize()
    knobs(KVP_TC_TIMEOUT_US=20000, KVP_TC_TEST_DELAY_SLOT=0)
    ko, vo = n.knorm_compress(k, v, 16384)
    torch.cuda.synchronize()
    assert torch.isnan(ko[0, 0].float()).all() and torch.isnan(vo[0, 0].float()).all()
    assert torch.equal(ko[0, 1:], ko_ref[0, 1:]) and torch.equal(vo[0, 1:], vo_ref[0, 1:])
    knobs(KVP_TC_TIMEOUT_US=None, KVP_TC_TEST_DELAY_SLOT=None)
    try:
        n.knorm_compress(k, v, 16384)
        raise AssertionError("the call after a failed compress must raise")
    except n.KvpressHipError as e:
        assert "cluster select" in str(e), str(e)
    ko2, vo2 = n.knorm_compress(k, v, 16384)
    torch.cuda.synchronize()
    n.async_error_check()
    assert torch.equal(ko2, ko_ref) and torch.equal(vo2, vo_ref)


def stale():
    """A workspace handed in as 'clean' after a failure WITHOUT a zero-fill (what a C-ABI caller might do; the Python binding drops
    its cached workspaces instead) still carries the cluster's flag: the rows are poisoned again and the failure is reported again."""
    import ctypes

    g = torch.Generator().manual_seed(5)
    sc = torch.randn(8, 131008, generator=g, dtype=torch.float32).to(DEV)
    L = n.lib()
    nws = L.kvp_topk_workspace_bytes(8, 131008, 65472)
    ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
    idx = torch.empty((8, 65472), dtype=torch.int32, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    CLEAN = n.TOPK_WS_CLEAN

    def call():
        return L.kvp_topk_select(P(sc), 8, 131008, 131008, 65472, n.ORDER_POSITION | CLEAN, P(idx), P(ws), nws, st)

    assert call() == 0
    torch.cuda.synchronize()
    assert (idx >= 0).all()
    knobs(KVP_TC_TIMEOUT_US=20000, KVP_TC_TEST_DELAY_SLOT=3)
    assert call() == 0
    torch.cuda.synchronize()
    assert (idx[0] == -1).all()
    knobs(KVP_TC_TIMEOUT_US=20000, KVP_TC_TEST_DELAY_SLOT=None)   # (no fault from here on; the short timeout only bounds the stale row's polling)
    assert call() != 0, "the report of the failed launch"
    assert call() == 0                       # the same, still dirty workspace, declared clean: launches ...
    torch.cuda.synchronize()
    assert (idx[0] == -1).all(), "... poisons the stale cluster's row again ..."
    assert call() != 0, "... and reports it again (KVP_EASYNC)"
    ws.zero_()
    knobs(KVP_TC_TIMEOUT_US=None)
    assert call() == 0
    torch.cuda.synchronize()
    assert (idx >= 0).all()
    assert L.kvp_async_error_check() == 0


def paths():
    """Which form of the cluster select finishes which kind of row (the test twin leaves a marker per (cluster, slot) in the workspace):
    the two-hop form (sample-steered first digit, candidates, local finish) on rows whose threshold the sample brackets -- with ties at
    the threshold across slots, with single-key bins (no local round), with keys spread over every exponent (several local rounds) --
    and the three-round form where the sample must mislead (k at an extreme, rows of few distinct values, sorted rows).  Indices
    against the oracle in every case."""
    import ctypes

    rs = np.random.RandomState(11)
    R, S = 8, 131008
    L = n.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    al = lambda x: (x + 255) // 256 * 256
    bar_off = (2 * al(R * 4096 * 4) + al(R * 256 * 4)) // 4 + 32 * 32 + 32   # topk_ws_layout: hist1, hist2, hist3, bar (+ TC_MAXC = 32 lines + 1)

    def run(x, k, flags=0):
        sc = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
        nws = L.kvp_topk_workspace_bytes(R, x.shape[1], k)
        ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
        idx = torch.empty((R, k), dtype=torch.int32, device=DEV)
        for _ in range(2):   # twice through the same self-cleaning workspace
            rc = L.kvp_topk_select(P(sc), R, x.shape[1], x.shape[1], k, n.ORDER_POSITION | n.TOPK_WS_CLEAN | flags, P(idx), P(ws), nws, st)
            assert rc == 0, L.kvp_last_error()
            torch.cuda.synchronize()
        m = ws.view(torch.int32).cpu().numpy()[bar_off:bar_off + 32 * 32].reshape(32, 32)[:, :R]
        assert L.kvp_async_error_check() == 0
        return idx.cpu().numpy(), m

    def check(name, x, k, want_form, smallest=False):
        got, m = run(x, k, n.TOPK_SMALLEST if smallest else 0)
        want = O.topk_select(-x if smallest else x, k)
        assert np.array_equal(got, want), f"{name}: wrong indices"
        forms = sorted(set(m.reshape(-1).tolist()))
        # want_form 2 / 1: every row must finish in that form; 0: rows may differ (a row whose extreme happens to lie inside the
        # sample's bracket finishes in the two-hop form), but each row's 32 slots agree
        assert forms == [want_form] or (want_form == 0 and set(forms) <= {1, 2}), f"{name}: form markers {forms}, expected {want_form} (2 = two-hop, 1 = three rounds)"
        assert all(len(set(m[:, r].tolist())) == 1 for r in range(R)), f"{name}: the slots of a row disagree about the form: {m.T.tolist()}"
        print(f"paths {name}: forms {forms} ok", flush=True)

    flat = (2.0 ** -17 * (1 + 0.05 * rs.standard_normal((R, S)))).astype(np.float32)
    for frac in (0.5, 0.1, 0.3, 0.7, 0.9):
        check(f"flat k/S={frac}", flat, int(S * frac), 2)
    check("flat smallest", flat, S // 3, 2, smallest=True)
    # ties AT the threshold, spread over the slots: 48 positions whose score is below the threshold get the threshold's value
    k = S // 2
    x = flat.copy()
    for r in range(R):
        order = np.argsort(-x[r], kind="stable")
        v = x[r, order[k - 1]]
        below = order[k + 100:]
        x[r, rs.choice(below, 48, replace=False)] = v
    check("ties at the threshold", x, k, 2)
    # consecutive keys: 200 distinct values, ~650 each -- every window bin is ONE key (no local round), the quota cuts inside a value
    x = (np.float32(1.0).view(np.uint32) + rs.randint(0, 200, size=(R, S)).astype(np.uint32)).view(np.float32)
    check("consecutive keys", x, S // 2, 2)
    # keys over every exponent and both signs: a wide bracket, several local rounds
    x = (np.exp(12 * rs.standard_normal((R, S))) * rs.choice([-1.0, 1.0], size=(R, S))).astype(np.float32)
    check("all exponents", x, S // 2, 2)
    x = flat.copy()
    x[:, ::97] = np.inf
    x[:, 5::89] = -np.inf
    check("infinities", x, S // 2, 2)
    # --- rows on which the sample must mislead: the three-round form takes over, same indices
    check("k = 1", flat, 1, 0)
    check("k = S - 1", flat, S - 1, 0)
    check("20 distinct values", rs.randint(0, 20, size=(R, S)).astype(np.float32), S // 2, 1)
    check("sorted rows", np.sort(flat, axis=1), S // 2, 1)
    check("constant rows", np.full((R, S), 0.25, np.float32), S // 2, 1)


if __name__ == "__main__":
    assert os.environ.get("KVPRESS_HIP_LIB", "").endswith("faultinject.so"), "run me with the fault-injection library"
    {"barrier": barrier, "fused": fused, "stale": stale, "paths": paths}[sys.argv[1]]()
    print("CHILD_PASS", sys.argv[1])
