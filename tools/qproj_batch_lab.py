#!/usr/bin/env python3
"""Window projection for MORE than one batch element: the library kernel (csrc/qproj.hip, grid = columns x batch: the
weight slice is streamed once per element) against the model's own GEMM over all B x 64 rows (reads the weight once) + a RoPE launch.
Prints us per call for B = 1, 2, 4, 8, 128 (ChunkPress hands over 128 windows).  Lab tool, not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from kvpress_amd import _native  # noqa: E402


def timeit(fn, n=50, warm=10):
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


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    wq = (torch.randn((4096, 4096), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    ang = torch.rand((1, 64, 128), generator=g, device=dev)
    cos, sin = torch.cos(ang).to(torch.bfloat16), torch.sin(ang).to(torch.bfloat16)
    # something between the calls that evicts the weight from the L2s, as the passes and the gather of a real step do
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for B in (1, 2, 4, 8, 128):
        h = torch.randn((B, 64, 4096), generator=g, device=dev).to(torch.bfloat16)

        def lib():
            big.zero_()
            return _native.snapkv_qproj_rope(h, wq, cos, sin)

        def gemm():
            big.zero_()
            return F.linear(h, wq)

        def flush():
            big.zero_()

        t_f = timeit(flush)
        print(f"B={B:3d}: library qproj+rope {timeit(lib) - t_f:7.1f} us   model GEMM alone {timeit(gemm) - t_f:7.1f} us (+ ~5 us RoPE launch)   "
              f"[each after a 256 MiB memset that takes {t_f:.1f} us, subtracted]", flush=True)


if __name__ == "__main__":
    main()
