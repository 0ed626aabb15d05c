"""Throughput of the auxiliary scorer kernels on Llama-3.1-8B geometry (B=1, H_kv=8, D=128, bf16): us per call (CUDA events,
20 calls after 5 warm-up) and algorithmic GB/s (bytes the kernel must read once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kvpress_amd import _native as N

dev = "cuda:0"


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for S in (32768, 131072):
    g = torch.Generator(device=dev); g.manual_seed(0)
    k = torch.randn((1, 8, S, 128), generator=g, device=dev).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=dev).to(torch.bfloat16)
    q = torch.randn((1, 32, 32, 128), generator=g, device=dev).to(torch.bfloat16)
    filt = torch.randn((8, 128), generator=g, device=dev).to(torch.bfloat16)
    kv_bytes = k.numel() * 2
    rows = [("rownorm (Knorm)", lambda: N.rownorm_score(k, -1.0), kv_bytes),
            ("rowdot (QFilter)", lambda: N.rowdot_score(k, filt, -1.0), kv_bytes),
            ("keydiff", lambda: N.keydiff_score(k), 2 * kv_bytes),
            ("cur kv_product", lambda: N.cur_score(k, v, "kv_product", 16, 4), 2 * kv_bytes),
            ("lagkv (rank)", lambda: N.lagkv_score(k, v, 4, 128, False), 4 * kv_bytes),
            ("think channel scores", lambda: N.think_channel_scores(q, k), kv_bytes)]
    idx = torch.arange(0, 128, 2, device=dev, dtype=torch.int32)[None, None].expand(1, 8, -1).contiguous()
    kk = k.clone()
    rows.append(("zero 64 of 128 channels", lambda: N.zero_channels_(kk, idx), kv_bytes // 2))
    for name, fn, nbytes in rows:
        us = timed(fn)
        print(f"S={S:6d} {name:26s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
# ||Wo v||_1 rows: [S, 4096] bf16 projected values of one q-head
for S in (32768,):
    x = torch.randn((S, 4096), device=dev).to(torch.bfloat16)
# observed attention: [1, 32, 4096, 4096] bf16 eager weights
a = torch.rand((1, 32, 4096, 4096), device=dev).to(torch.bfloat16)
us = timed(lambda: N.observed_attention_score(a, 8), 5)
print(f"S=  4096 {'observed attention':26s} {us:8.1f} us  {a.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
