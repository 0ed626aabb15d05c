#!/usr/bin/env python3
"""Lab: ExpectedAttention's score call at head size 256 over the sequence length, with and without the covariance (the mean-only form runs the same
K stream, LDS reads and row-dots without the matrix instructions) -- how the D = 256 quadratic-form kernel's tile time splits (LAB_NOTEBOOK R6.12).
Measurement aid, not part of the product."""
import sys, torch
sys.path.insert(0, "/root/repo")
from kvpress_amd import _native as N
dev = torch.device("cuda", 0)
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
D, Hkv, Hq = 256, 8, 32
mu = torch.randn((1, Hq, D), device=dev) * 0.3
a = torch.randn((1, Hq, D, D), device=dev) * 0.04
cov = a @ a.transpose(-1, -2)
for S in (8192, 16384, 32768, 65536, 131072):
    k = torch.randn((1, Hkv, S, D), device=dev).bfloat16(); v = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
    t = timeit(lambda: N.ea_score(k, v, mu, cov, 4, True, 0.0))
    t0 = timeit(lambda: N.ea_score(k, v, mu, None, 4, True, 0.0))
    print(f"S={S}: ea_score {t:.1f} us, mean-only {t0:.1f} us", flush=True)
