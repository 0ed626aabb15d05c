#!/usr/bin/env python3
"""What the reference's algorithm costs on the SAME MI355X when it is expressed in plain PyTorch-ROCm ops (BASELINE.md §3,
"third column"), next to this package's press on the same tensors.

The reference itself cannot travel to the GPU box, so the op sequence of its three scorers and of ``ScorerPress.compress``
is restated here with torch calls, one line per reference line (cited) -- same intermediates, same dtypes, same rounding
points, nothing fused.  Measurement aid (under tests/ because it runs the oracle's torch restatement): nothing in the package, the test suite or
bench.py imports this file, and pytest does not collect it.

    python tests/torch_path_gpu.py [--workload snapkv128k|knorm32k|knorm128k|ea128k] [--reps 10]

Prints one JSON line per workload: ms/layer of the torch path and of the HIP path (CUDA events on the current stream,
3 warm-up calls), the peak extra memory of each, and the overlap of the two retained sets.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402  (module geometry + build_module)


from oracle.torch_path import SCORERS, torch_compress  # noqa: E402  (the cited restatement of the reference's op sequence)


WORKLOADS = {"knorm32k": ("knorm", 32768, 0.5), "knorm128k": ("knorm", 131072, 0.5), "snapkv128k": ("snapkv", 131072, 0.5),
             "ea128k": ("ea", 131072, 0.7)}


def timed(fn, reps):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (torch.cuda.max_memory_allocated() - base) / 2**20, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=["all"] + list(WORKLOADS))
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    att, rot = bench.build_module(dev)
    for wl in (list(WORKLOADS) if args.workload == "all" else [args.workload]):
        kind, S, ratio = WORKLOADS[wl]
        g = torch.Generator().manual_seed(0)
        keys = torch.randn((1, bench.H_KV, S, bench.D), generator=g).to(dev, torch.bfloat16)
        values = torch.randn((1, bench.H_KV, S, bench.D), generator=g).to(dev, torch.bfloat16)
        hidden = torch.randn((1, S, bench.HIDDEN), generator=g).to(dev, torch.bfloat16)
        with torch.no_grad():
            pe = rot(hidden, torch.arange(S, device=dev)[None])
            kwargs = {"position_embeddings": pe}
            press = bench.make_press(kind, ratio)
            t_torch, m_torch, (k1, v1, i1) = timed(lambda: torch_compress(SCORERS[kind], ratio, att, hidden, keys, values, kwargs), args.reps)
            t_hip, m_hip, (k2, v2) = timed(lambda: press.compress(att, hidden, keys, values, None, kwargs), args.reps)
        # retained sets: recover ours from the kept keys' positions is not possible without indices -> compare via scores
        sc = press.score(att, hidden, keys, values, None, kwargs)
        from kvpress_amd import _native
        i2 = _native.topk_select(sc, k2.shape[2]).long()
        n = i1.shape[-1]
        inter = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i1.reshape(-1, n).cpu(), i2.reshape(-1, n).cpu())) / (i1.numel())
        print(json.dumps({"workload": wl, "torch_rocm_ms_per_layer": round(t_torch, 3), "hip_ms_per_layer": round(t_hip, 4),
                          "speedup": round(t_torch / t_hip, 1), "torch_peak_extra_MiB": round(m_torch), "hip_peak_extra_MiB": round(m_hip),
                          "kept_set_overlap_vs_bf16_torch_path": round(inter, 4), "n_kept": int(n), "reps": args.reps,
                          "torch": torch.__version__}), flush=True)
        del keys, values, hidden, k1, v1, k2, v2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
