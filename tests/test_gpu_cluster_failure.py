"""The one-launch select of long rows must never return wrong indices silently.

`torch.topk` (kvpress/presses/scorer_press.py:95) either returns the right indices or raises.  The cluster select
(kvpress_amd/csrc/topk_cluster.hip) synchronises the 32 workgroups of a row inside the kernel; these tests pin what happens
when they are NOT all resident at once:
  * CUs held by another stream for a while -> the select waits, the indices are right;
  * a reduced CU set (HSA_CU_MASK, own process) -> right indices or a raised KvpressHipError, never garbage;
  * a barrier that really times out (fault-injection twin of the library, child process: one workgroup arrives late) -> indices -1,
    gathered rows NaN, the NEXT library call raises KVP_EASYNC once, the workspace cache is dropped, the call after that is correct.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import kvpress_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def native():
    from kvpress_amd import _native

    return _native


def _scores(R=8, S=131008, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(R, S, generator=g, dtype=torch.float32)


def test_select_waits_for_cus_held_by_another_stream():
    """200 x 1024-thread workgroups with 150 KiB of LDS each (no select workgroup fits beside one) hold 200 of the 256 CUs for
    30 ms on a second stream while the cluster select is enqueued on the first: only 56 of its 256 workgroups are resident at
    first, the rest as CUs free up; the result must be the oracle's (and the same when all 256 CUs are held)."""
    n = native()
    sc = _scores()
    want = O.topk_select(sc.numpy(), 65472)
    d = sc.to(DEV)
    side = torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    for rep in range(4):
        n.occupy_cus(200 if rep < 2 else 256, 1024, 150 * 1024, 30000, stream=side)
        got = n.topk_select(d, 65472)
        torch.cuda.synchronize()
        n.async_error_check()
        assert np.array_equal(got.cpu().numpy(), want), f"rep {rep}"


def test_cluster_selects_on_two_streams_of_one_process():
    """Round 6 (LAB R6.12): two cluster selects dispatched at the same moment from different queues can deadlock each other until the
    time-out (each holds CUs the other's rows wait for).  Within a process the library serialises them itself: a cluster launch on a
    stream other than the previous one's waits on the device for that previous launch.  300 un-synchronised pairs on two streams -- plain
    select, fused Knorm compress -- must all be right and leave no asynchronous failure; a third stream joins later."""
    n = native()
    a, b = _scores(seed=1), _scores(seed=2)
    want_a, want_b = O.topk_select(a.numpy(), 65472), O.topk_select(b.numpy(), 30000)
    da, db = a.to(DEV), b.to(DEV)
    g = torch.Generator(device=DEV); g.manual_seed(3)
    k = torch.randn((1, 8, 32768, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 8, 32768, 128), generator=g, device=DEV).to(torch.bfloat16)
    ko_ref, vo_ref = n.knorm_compress(k, v, 16384)
    sa, sb, sc = (torch.cuda.Stream(device=DEV) for _ in range(3))
    torch.cuda.synchronize()
    got = []
    for it in range(300):
        with torch.cuda.stream(sa):
            ga = n.topk_select(da, 65472)
        with torch.cuda.stream(sb):
            gb = n.topk_select(db, 30000) if it % 3 else None
            kc = n.knorm_compress(k, v, 16384) if it % 3 == 0 else None
        if it >= 150 and it % 5 == 0:
            with torch.cuda.stream(sc):
                got.append(("c", n.topk_select(db, 30000)))
        if it % 25 == 0:
            got.append(("a", ga))
            if gb is not None:
                got.append(("b", gb))
            if kc is not None:
                got.append(("k", kc))
    torch.cuda.synchronize()
    n.async_error_check()
    for tag, x in got:
        if tag == "a":
            assert np.array_equal(x.cpu().numpy(), want_a)
        elif tag in ("b", "c"):
            assert np.array_equal(x.cpu().numpy(), want_b)
        else:
            assert torch.equal(x[0], ko_ref) and torch.equal(x[1], vo_ref)


_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from kvpress_amd import _native as n
from oracle import kvpress_oracle as O
g = torch.Generator().manual_seed(0)
sc = torch.randn(8, 131008, generator=g, dtype=torch.float32)
want = O.topk_select(sc.numpy(), 65472)
d = sc.to("cuda:0")
try:
    for _ in range(3):
        got = n.topk_select(d, 65472)
        torch.cuda.synchronize()
        n.async_error_check()
        assert np.array_equal(got.cpu().numpy(), want), "WRONG INDICES"
    print("CHILD_OK correct")
except n.KvpressHipError as e:
    print("CHILD_OK raised", str(e)[:80])
"""


@pytest.mark.parametrize("mask", ["0:0-127", "0:0-63"])
def test_select_under_a_cu_mask_is_right_or_raises(mask):
    """Half / a quarter of the CUs (HSA_CU_MASK) in a process of its own: with a row's workgroups on consecutive blocks whole
    clusters still become resident -> correct indices (or, if the runtime reports the reduced CU count, the (chunk, row) passes);
    a raised KvpressHipError is acceptable, wrong indices are not.  A runtime that rejects the mask syntax skips the test."""
    env = dict(os.environ, HSA_CU_MASK=mask, PYTHONDONTWRITEBYTECODE="1", KVP_TC_TIMEOUT_US="300000")
    try:
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        pytest.skip(f"child under HSA_CU_MASK={mask} did not finish in 7 minutes on this runtime")
    out = r.stdout + r.stderr
    assert "WRONG INDICES" not in out, out[-2000:]
    if "CHILD_OK" not in out:
        pytest.skip(f"child did not run under HSA_CU_MASK={mask}: {out[-300:]}")


@pytest.mark.parametrize("scenario", ["barrier", "fused", "stale"])
def test_barrier_timeout_is_loud(scenario):
    """A barrier that REALLY times out: one workgroup of a cluster arrives 2 x timeout late.  The hook that makes it late exists only
    in the fault-injection twin of the library (kvpress_amd/build.py: topk_cluster.hip with -DKVP_TC_FAULT_INJECTION), loaded by a child
    process; the scenarios (tests/_fault_child.py): `barrier` -- kvp_topk_select: row 0 = -1, other rows correct, gathered rows NaN, the
    next call raises KVP_EASYNC exactly once, cached workspaces dropped, then correct again; `fused` -- the same through
    kvp_knorm_compress; `stale` -- a workspace reused as clean without a zero-fill is poisoned AND reported again."""
    from kvpress_amd import build

    assert os.path.exists(build.FAULT_LIB), "fault-injection twin not built (python -m kvpress_amd.build)"
    env = dict(os.environ, KVPRESS_HIP_LIB=build.FAULT_LIB, PYTHONDONTWRITEBYTECODE="1")
    for k in ("KVP_TC_TIMEOUT_US", "KVP_TC_TEST_DELAY_SLOT", "KVP_TK_CLUSTER"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_child.py"), scenario], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"CHILD_PASS {scenario}" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_both_forms_of_the_cluster_select_are_exercised():
    """Round 5: the cluster select has a two-hop form (sample-steered first digit, per-slot candidate records, local finish) and keeps
    the three-round form for rows on which the sample misleads.  The test twin of the library marks which form finished each
    (cluster, slot); tests/_fault_child.py `paths` runs rows built for each form -- flat BASELINE rows at five keep ratios, ties at
    the threshold across slots, single-key bins, keys over every exponent, infinities / k at an extreme, few distinct values, sorted
    and constant rows -- and checks the markers AND the indices (oracle) of every case."""
    from kvpress_amd import build

    assert os.path.exists(build.FAULT_LIB), "fault-injection twin not built (python -m kvpress_amd.build)"
    env = dict(os.environ, KVPRESS_HIP_LIB=build.FAULT_LIB, PYTHONDONTWRITEBYTECODE="1")
    for k in ("KVP_TC_TIMEOUT_US", "KVP_TC_TEST_DELAY_SLOT", "KVP_TK_CLUSTER"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fault_child.py"), "paths"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_PASS paths" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_product_library_has_no_fault_injection_hook(knobs):
    """The same knob on the PRODUCT library does nothing: the select is correct and nothing is reported."""
    n = native()
    sc = _scores(seed=2)
    want = O.topk_select(sc.numpy(), 65472)
    knobs(KVP_TC_TIMEOUT_US=20000, KVP_TC_TEST_DELAY_SLOT=5)
    got = n.topk_select(sc.to(DEV), 65472)
    torch.cuda.synchronize()
    n.async_error_check()
    assert np.array_equal(got.cpu().numpy(), want)


def test_gather_never_reads_through_a_bad_index():
    """Indices outside [0, S) -- not only -1 -- give NaN rows in every dtype and path (vector, scalar, with re-rotation)."""
    n = native()
    for dt, D in ((torch.bfloat16, 128), (torch.float16, 64), (torch.float32, 6), (torch.bfloat16, 6)):
        k = torch.randn(1, 2, 50, D, device=DEV).to(dt)
        v = torch.randn(1, 2, 50, D, device=DEV).to(dt)
        idx = torch.tensor([[[0, -1, 7, 50, 49], [3, 4, -7, 2**30, 1]]], dtype=torch.int32, device=DEV)
        ko, vo = n.gather_kv(k, v, idx)
        bad = torch.tensor([[[0, 1, 0, 1, 0], [0, 0, 1, 1, 0]]], dtype=torch.bool, device=DEV)
        for o, src in ((ko, k), (vo, v)):
            assert torch.isnan(o.float()[bad]).all(), (dt, D)
            good = ~bad
            want = torch.gather(src, 2, idx.clamp(0, 49).long()[..., None].expand(-1, -1, -1, D))
            assert torch.equal(o[good], want[good]), (dt, D)
    k = torch.randn(1, 2, 50, 128, device=DEV).to(torch.bfloat16)
    v = torch.randn(1, 2, 50, 128, device=DEV).to(torch.bfloat16)
    idx = torch.tensor([[[0, -1, 7, 20, 49], [3, 4, 9, -1, 11]]], dtype=torch.int32, device=DEV)
    inv = (1.0 / (10000 ** (torch.arange(0, 64, dtype=torch.float32) / 64))).to(DEV)
    ko, vo = n.gather_kv_rerotate(k, v, idx, inv)
    assert torch.isnan(ko[0, 0, 1].float()).all() and torch.isnan(vo[0, 1, 3].float()).all()
    assert torch.equal(vo[0, 0, 2], v[0, 0, 7]) and not torch.isnan(ko[0, 0, 2].float()).any()
