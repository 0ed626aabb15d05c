#!/usr/bin/env python3
"""Kernel lab for the SnapKV window-attention passes: times snapkv_p1_mfma / snapkv_p2_mfma (HIP events on the launch
stream, kvp_prof_*) under environment-selected variants on the BASELINE shape (B=1, H_q=32, H_kv=8, S=131072, D=128,
W=64, bf16, random data) and checks every variant's scores against variant 0.

    python tools/sk_lab.py "" "KVP_SK_SLOTS=128" ...
    KVPRESS_HIP_LIB=kvpress_amd/lib/variants/nodma.so python tools/sk_lab.py      (ablated builds: tools/build_variants.sh)

Each argument is one configuration: space-separated NAME=VALUE pairs put into the environment for that run (the
library reads these measurement knobs per launch).  Measurement aid, not part of the product path.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from kvpress_amd import _native  # noqa: E402


def stamps():
    """Cycle breakdown of pass 1's waves (lab build: apply tools/lab_patches/sk_stamp.diff, then tools/build_variants.sh "stamp=GEN_STAMP=1 KVP_VARIANT_CFLAGS=-DKVP_SK_STAMP",
    run with KVPRESS_HIP_LIB=kvpress_amd/lib/variants/stamp.so): the asm loop returns, per wave, the shader cycles of the whole loop,
    those between the two s_memtime stamps around every stage head's `s_waitcnt lgkmcnt(0)` (K fragments), around every tile's
    `s_waitcnt vmcnt` (K stream) and around every tile's `s_barrier`."""
    import ctypes

    import numpy as np

    S = int(os.environ.get("SK_LAB_S", 131072))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    keys = torch.randn((1, 8, S, 128), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q = (torch.randn((1, 32, 64, 128), generator=g, device=dev, dtype=torch.float32) * 1.3).to(torch.bfloat16)
    L = _native.lib()
    nws = L.kvp_snapkv_workspace_bytes(1, 32, 8, S, 64, 128)
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    sc = torch.empty((1, 8, S), dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(40):
        rc = L.kvp_snapkv_score(P(q), q.stride(0), q.stride(1), q.stride(2), P(keys), keys.stride(0), keys.stride(1), keys.stride(2), 2, 1, 32, 8, S, 64, 128, 5,
                                P(sc), P(ws), nws, st)
        assert rc == 0, L.kvp_last_error()
    torch.cuda.synchronize()
    nchunk = 32                                           # 256 workgroup slots / 8 kv-heads (snapkv_mfma_nchunk)
    rows = 32 * 64
    f = ws.view(torch.float32).cpu().numpy()
    off_m = 4096                                          # after the pool kernel's 4096 block maxima
    nchunk_max = 256
    pm = f[off_m: off_m + rows * nchunk].reshape(rows, nchunk)
    pz = f[off_m + rows * nchunk_max: off_m + rows * nchunk_max + rows * nchunk].reshape(rows, nchunk)
    total, lgkm = pm[0::2].ravel().astype(np.float64), pm[1::2].ravel().astype(np.float64)
    vm, bar = pz[0::2].ravel().astype(np.float64), pz[1::2].ravel().astype(np.float64)
    ntile = S // 128 // nchunk
    ok = (total > 0) & (total < 1e7) & (lgkm >= 0) & (lgkm < total) & (bar >= 0) & (bar < total) & (vm >= 0) & (vm < total)   # (the walks that end in masked tiles continue in C++)
    total, lgkm, vm, bar = total[ok], lgkm[ok], vm[ok], bar[ok]
    print(f"pass 1, {ntile} tiles ({4 * ntile} stages) per wave, {total.size // 16} waves (shader cycles per wave: mean, min .. max)")
    for name, v in (("whole loop", total), ("stage heads: wait for K fragments (lgkmcnt), incl. 2 stamps", lgkm), ("tile ends: wait for the K stream (vmcnt), incl. 1 stamp", vm),
                    ("tile ends: s_barrier, incl. 1 stamp", bar)):
        print(f"  {name:62s} {v.mean():10.0f}  {v.min():9.0f} .. {v.max():9.0f}   {100 * v.mean() / total.mean():5.1f} %")
    rest = total - lgkm - vm - bar
    print(f"  {'issue (MFMA + VALU + LDS / DMA requests), everything else':62s} {rest.mean():10.0f}  {rest.min():9.0f} .. {rest.max():9.0f}   {100 * rest.mean() / total.mean():5.1f} %")
    print(f"  per stage: {total.mean() / (4 * ntile):.0f} cycles in all; fragments {lgkm.mean() / (4 * ntile):.0f}, K stream {vm.mean() / ntile:.0f} per tile, barrier {bar.mean() / ntile:.0f} per tile")


def main():
    if "--stamps" in sys.argv:
        return stamps()
    cfgs = sys.argv[1:] or [""]
    S = int(os.environ.get("SK_LAB_S", 131072))
    reps = int(os.environ.get("SK_LAB_REPS", 12))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    keys = torch.randn((1, 8, S, 128), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q = (torch.randn((1, 32, 64, 128), generator=g, device=dev, dtype=torch.float32) * 1.3).to(torch.bfloat16)
    ref = None
    touched = set()
    for cfg in cfgs:
        for k in touched:
            os.environ.pop(k, None)
        for kv in cfg.split():
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        for _ in range(30):  # clocks
            sc = _native.snapkv_score(q, keys, 5)
        torch.cuda.synchronize()
        _native.prof_enable(True)
        for _ in range(reps):
            sc = _native.snapkv_score(q, keys, 5)
        torch.cuda.synchronize()
        t = {}
        for name, ms in _native.prof_records():
            t.setdefault(name, []).append(ms * 1e3)
        _native.prof_enable(False)
        # back-to-back wall time of the whole score call (no event overhead)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            sc = _native.snapkv_score(q, keys, 5)
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 1e3 / 50
        msg = ""
        if ref is None:
            ref = sc.clone()
        else:
            d = (sc[..., :-64] - ref[..., :-64]).abs() / ref[..., :-64].abs().clamp_min(1e-30)
            msg = f"max rel diff vs first cfg {float(d.max()):.2e}" + ("  (bit-identical)" if torch.equal(sc, ref) else "")

        def med(*names):
            v = sorted(sum((t.get(n, []) for n in names), []) or [0.0])
            return v[len(v) // 2]

        print(f"{cfg:44s} p1 {med('snapkv_p1_mfma', 'snapkv_p1_asm'):7.1f} us  p2 {med('snapkv_p2_mfma', 'snapkv_p2_asm'):7.1f} us  combine {med('softmax_combine_kernel'):5.1f}  "
              f"pool {med('snapkv_pool_kernel'):5.1f}  score call {wall:7.1f} us   {msg}", flush=True)


if __name__ == "__main__":
    main()
