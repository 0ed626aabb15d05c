#!/usr/bin/env python3
"""Would two head-groups on two streams (each running the whole fused compress chain for half the kv-heads) finish sooner than
one call over all heads?  The launch-bound links of one chain (combine, pool, select) could then hide behind the other chain's
attention passes.  Measurement aid; prints wall time per compress for 1 stream x all heads and for G groups on G streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kvpress_amd import _native  # noqa: E402


def timeit(fn, n=40):
    for _ in range(30):
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
    S, W, D = 131072, 64, 128
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    keys = torch.randn((1, 8, S, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    values = torch.randn((1, 8, S, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q = (torch.randn((1, 32, W, D), generator=g, device=dev, dtype=torch.float32) * 1.3).to(torch.bfloat16)
    cos = torch.rand((1, W, D), generator=g, device=dev).to(torch.bfloat16)
    sin = torch.rand((1, W, D), generator=g, device=dev).to(torch.bfloat16)
    n = S // 2

    def whole():
        return _native.snapkv_compress_rope(q, cos, sin, keys, values, 5, n)

    print(f"one call, all 8 heads: {timeit(whole):7.1f} us", flush=True)
    main_s = torch.cuda.current_stream()
    for G in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(G)]
        hk = 8 // G

        def split():
            ev = torch.cuda.Event()
            ev.record(main_s)
            outs = []
            for i, st in enumerate(streams):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    outs.append(_native.snapkv_compress_rope(q[:, i * hk * 4:(i + 1) * hk * 4], cos, sin, keys[:, i * hk:(i + 1) * hk],
                                                             values[:, i * hk:(i + 1) * hk], 5, n))
            for st in streams:
                main_s.wait_stream(st)
            return outs

        def serial_groups():
            return [_native.snapkv_compress_rope(q[:, i * hk * 4:(i + 1) * hk * 4], cos, sin, keys[:, i * hk:(i + 1) * hk],
                                                 values[:, i * hk:(i + 1) * hk], 5, n) for i in range(G)]

        print(f"{G} head-groups: one stream {timeit(serial_groups):7.1f} us   {G} streams {timeit(split):7.1f} us", flush=True)
    ref = whole()
    outs = split()
    print("results equal:", torch.equal(torch.cat([o[0] for o in outs], 1), ref[0]) and torch.equal(torch.cat([o[1] for o in outs], 1), ref[1]))


if __name__ == "__main__":
    main()
