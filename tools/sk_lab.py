#!/usr/bin/env python3
"""Kernel lab for the SnapKV window-attention passes: times snapkv_p1_mfma / snapkv_p2_mfma (HIP events on the launch
stream, kvp_prof_*) under environment-selected variants on the BASELINE shape (B=1, H_q=32, H_kv=8, S=131072, D=128,
W=64, bf16, random data) and checks every variant's scores against variant 0.

    python tools/sk_lab.py "KVP_SK_ASM=1" "KVP_SK_ASM=0" ...
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


def main():
    cfgs = sys.argv[1:] or ["KVP_SK_ASM=1"]
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
