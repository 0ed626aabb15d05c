"""Whole-compress time and per-kernel split vs sequence length (Llama-3.1-8B geometry, bf16, ratio 0.5):
kvp_knorm_compress and kvp_snapkv_compress_rope.  HIP events via kvp_prof_* for the split, CUDA events for the wall time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kvpress_amd import _native

dev = "cuda:0"
SIZES = [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768]
for S in SIZES:
    g = torch.Generator(device=dev); g.manual_seed(0)
    k = torch.randn((1, 8, S, 128), generator=g, device=dev).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=dev).to(torch.bfloat16)
    q = torch.randn((1, 32, 64, 128), generator=g, device=dev).to(torch.bfloat16)
    cos = torch.ones((1, 64, 128), device=dev, dtype=torch.bfloat16)
    sin = torch.zeros((1, 64, 128), device=dev, dtype=torch.bfloat16)
    n = S // 2
    for name, fn in (("knorm", lambda: _native.knorm_compress(k, v, n)), ("snapkv", lambda: _native.snapkv_compress_rope(q, cos, sin, k, v, 5, n))):
        for _ in range(100):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 10.0  # us per call
        _native.prof_enable(True)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = {}
        for kn, ms in _native.prof_records():
            t.setdefault(kn, []).append(ms * 1e3)
        _native.prof_enable(False)
        print(S, name, f"wall {wall:.1f} us", {kn: round(sum(x) / len(x), 1) for kn, x in t.items()}, flush=True)
