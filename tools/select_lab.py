#!/usr/bin/env python3
"""A/B of the select variants on the bench shapes (GPU): the cluster select (one launch) against the (chunk, row) passes, stand-alone
and inside the fused Knorm / SnapKV compress calls.  Back-to-back calls, wall clock over N repetitions after a warm-up.

    python tools/select_lab.py [--reps 300]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from kvpress_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def knobs(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    N.tuning_reload()


def timeit(fn, reps):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


def stamps(R, S, flat=True):
    """Phase time stamps of the cluster kernel (lab build with -DKVP_TC_TIMING, see tools/build_variants.sh tc_timing):
    run with KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_timing.so."""
    import ctypes

    import numpy as np

    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    sc = (2.0 ** -17 * (1 + 0.05 * torch.randn((R, S), generator=g, device=DEV))).float() if flat else torch.randn((R, S), generator=g, device=DEV)
    L = N.lib()
    nws = L.kvp_topk_workspace_bytes(R, S, S // 2)
    ws = torch.zeros(nws, dtype=torch.uint8, device=DEV)
    idx = torch.empty((R, S // 2), dtype=torch.int32, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    al = lambda x: (x + 255) // 256 * 256
    bar_off = (2 * al(R * 4096 * 4) + al(R * 256 * 4)) // 4 + 32 * 32 + 32
    names = ["start", "keys", "window", "hw flushed", "digit", "records", "barrier", "rec in LDS", "T", "offsets", "end"]
    for it in range(6):
        rc = L.kvp_topk_select(P(sc), R, S, S, S // 2, N.TOPK_WS_CLEAN, P(idx), P(ws), nws, st)
        assert rc == 0, L.kvp_last_error()
        torch.cuda.synchronize()
    w = ws.view(torch.int32).cpu().numpy().astype(np.int64)[bar_off:bar_off + 32 * 32].reshape(32, 32)[:, :12]
    t0 = w[:, 0].min()
    rel = (w - t0) * 0.01   # us
    print(f"cluster kernel phase stamps, R={R} S={S} {'flat' if flat else 'wide'} (us since the first workgroup of cluster 0 started; min / median / max over its 32 slots)")
    print(f"  two-hop form finished the select in {int(w[:, 11].sum())} of 32 slots")
    for i, n in enumerate(names):
        col = rel[:, i]
        print(f"  {n:12s} {col.min():7.2f} {np.median(col):7.2f} {col.max():7.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--stamps", action="store_true")
    args = ap.parse_args()
    if args.stamps:
        stamps(8, 131008, True)
        stamps(8, 131008, False)
        stamps(8, 40000, True)
        return
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    ALL = dict(KVP_TK_CLUSTER=None)
    # ---- stand-alone select ----
    for R, S in ((8, 131008), (8, 32768), (8, 65536), (1, 131072), (16, 131072)):
        flat = (2.0 ** -17 * (1 + 0.05 * torch.randn((R, S), generator=g, device=DEV))).float()
        wide = torch.randn((R, S), generator=g, device=DEV)
        for name, sc in (("flat", flat), ("wide", wide)):
            out = {}
            # (with tools/lab_patches/tc_speculation.diff applied, KVP_TC_SPEC=0/1 switches the speculative second digit: profiles/r04_select_spec_lab.txt)
            for var, kv in (("cluster", dict(KVP_TK_CLUSTER=None)), ("passes", dict(KVP_TK_CLUSTER=0))):
                knobs(**ALL)
                knobs(**kv)
                ref = N.topk_select(sc, S // 2)
                out[var] = (timeit(lambda: N.topk_select(sc, S // 2), args.reps), ref)
            same = torch.equal(out["cluster"][1], out["passes"][1])
            print(f"select R={R} S={S} {name}: cluster {out['cluster'][0]:.1f} us, passes {out['passes'][0]:.1f} us, identical={same}", flush=True)
    # ---- fused Knorm compress ----
    for S in (32768, 131072):
        k = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
        v = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
        res = {}
        for var, kv in (("cluster_knorm", {}), ("passes", dict(KVP_TK_CLUSTER=0))):
            knobs(**ALL)
            knobs(**kv)
            ko, vo = N.knorm_compress(k, v, S // 2)
            res[var] = (timeit(lambda: N.knorm_compress(k, v, S // 2), args.reps), ko, vo)
        same = all(torch.equal(res[a][1], res["passes"][1]) and torch.equal(res[a][2], res["passes"][2]) for a in res)
        print(f"knorm_compress S={S}: " + ", ".join(f"{a} {res[a][0]:.1f} us" for a in res) + f", identical={same}", flush=True)
    # ---- fused SnapKV compress ----
    S = 131072
    k = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=DEV).to(torch.bfloat16)
    q = torch.randn((1, 32, 64, 128), generator=g, device=DEV).to(torch.bfloat16)
    ang = torch.rand((1, 64, 128), generator=g, device=DEV)
    c, si = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    res = {}
    for var, kv in (("cluster_pool", {}), ("passes", dict(KVP_TK_CLUSTER=0))):
        knobs(**ALL)
        knobs(**kv)
        ko, vo = N.snapkv_compress_rope(q, c, si, k, v, 5, S // 2)
        res[var] = (timeit(lambda: N.snapkv_compress_rope(q, c, si, k, v, 5, S // 2), args.reps), ko, vo)
    same = all(torch.equal(res[a][1], res["passes"][1]) and torch.equal(res[a][2], res["passes"][2]) for a in res)
    print("snapkv_compress_rope S=131072: " + ", ".join(f"{a} {res[a][0]:.1f} us" for a in res) + f", identical={same}", flush=True)
    knobs(**ALL)
    print("done", flush=True)


if __name__ == "__main__":
    main()
