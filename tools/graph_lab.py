#!/usr/bin/env python3
"""Does a HIP graph shorten the dependent launch chain of one compress call?  Captures bench.py's step (press.compress on the
BASELINE shapes) in a torch.cuda.CUDAGraph on a side stream and replays it back to back against the eager loop.

    python tools/graph_lab.py [workload]

Measurement aid (the hook path cannot replay a graph: every layer has its own K / V addresses)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "snapkv128k"
    kind, S, ratio = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    bf = torch.bfloat16
    keys = torch.randn((1, bench.H_KV, S, bench.D), generator=gen, device=dev, dtype=torch.float32).to(bf)
    values = torch.randn((1, bench.H_KV, S, bench.D), generator=gen, device=dev, dtype=torch.float32).to(bf)
    hidden = torch.randn((1, S, bench.HIDDEN), generator=gen, device=dev, dtype=bf)
    att, rot = bench.build_module(dev)
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=dev)[None])
    press = bench.make_press(kind, ratio)

    def step():
        with torch.no_grad():
            return press.compress(att, hidden, keys, values, None, {"position_embeddings": pe})

    def timeit(fn, n=200):
        for _ in range(60):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    eager = [timeit(step) for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = step()
    ref = step()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    graph = [timeit(g.replay) for _ in range(3)]
    eager2 = [timeit(step) for _ in range(2)]
    print(f"{wl}: eager {min(eager + eager2):.4f} ms/step ({', '.join(f'{x:.4f}' for x in eager + eager2)})  graph replay {min(graph):.4f} "
          f"({', '.join(f'{x:.4f}' for x in graph)})  identical output: {same}")


if __name__ == "__main__":
    main()
