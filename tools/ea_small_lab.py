#!/usr/bin/env python3
"""Lab: chunk plan of the small-head ExpectedAttention logits kernel (KVP_EA_SMALL_CHUNK x KVP_EA_SMALL_WGS), event-timed ea_score calls.
Needs a lab build of the library (hipcc ... -DKVP_EA_SMALL_LAB, see tools/build_variants.sh for the recipe; KVPRESS_HIP_LIB points at it): the
product has the plan's two numbers compiled in.  Record: profiles/r06_ea_small_heads.txt.  Measurement aid, not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kvpress_amd import _native as N  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for D, Hkv, S in ((64, 8, 32768), (64, 8, 131072), (96, 32, 32768), (64, 8, 4096)):
    Hq = 32
    k = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
    v = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
    mu = torch.randn((1, Hq, D), device=dev) * 0.3
    a = torch.randn((1, Hq, D, D), device=dev) * 0.05
    cov = a @ a.transpose(-1, -2)
    line = [f"D={D} Hkv={Hkv} S={S}: ea_score us"]
    for chunk in (256, 512, 1024, 2048, 4096):
        for wgs in (768, 2048, 8192):
            os.environ["KVP_EA_SMALL_CHUNK"], os.environ["KVP_EA_SMALL_WGS"] = str(chunk), str(wgs)
            N.tuning_reload()
            line.append(f"[{chunk},{wgs}] {timeit(lambda: N.ea_score(k, v, mu, cov, 4, True, 0.0)):.1f}")
    print("  ".join(line), flush=True)
