#!/usr/bin/env python3
"""Host-side cost of one press.compress() call (GPU box): in the decode regime (a 2k-token cache) the kernels of a
SnapKV compress take ~20 us of device time while the step takes ~70 us -- the rest is Python.  Prints, per workload, the
per-call ENQUEUE time (no sync inside the loop: the host's own cost as long as the GPU keeps up), the synced step time and
the cProfile top of the enqueue loop.

    python tools/host_overhead_probe.py [--workload decode_snapkv2k] [--calls 2000]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="decode_snapkv2k")
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    kind, S, ratio = bench.WORKLOADS[args.workload]
    keys, values, hidden, _ = bench.bench_inputs(args.workload, 0, dev)
    att, rot = bench.build_module(dev)
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=dev)[None])
    kwargs = {"position_embeddings": pe}
    press = bench.make_press(kind, ratio)

    def step():
        with torch.no_grad():
            return press.compress(att, hidden, keys, values, None, kwargs)

    for _ in range(200):
        step()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(args.calls):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{args.workload}: enqueue {1e6 * (t1 - t0) / args.calls:.1f} us/call, with the final sync {1e6 * (t2 - t0) / args.calls:.1f} us/call", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.calls):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(args.top)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
