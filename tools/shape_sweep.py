#!/usr/bin/env python3
"""Performance evidence for the shapes OUTSIDE the benchmark's ("the fast paths are cut to the benchmark's shape ...
generic fallbacks that are parity-tested but have no performance evidence").  Times the fused compress calls of the three core scorers on
shapes that leave the hand-scheduled paths -- other head sizes, float32, other windows, GQA group sizes 1 / 2 / 3 / 8, many rows -- next to
the benchmark's own shape, and prints ms per call and the fraction of the 8 TB/s HBM roofline on SURVEY section 8(d)'s algorithmic bytes.
Lab tool (profiles/r06_shape_sweep.txt), not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kvpress_amd import _native as N  # noqa: E402

DEV = torch.device("cuda", 0)


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
    return e0.elapsed_time(e1) * 1e3 / n   # us


def case(kind, B, Hq, Hkv, S, D, dt, W=64, ks=5, ratio=0.5, note=""):
    g = torch.Generator(device=DEV)
    g.manual_seed(S + D + Hq)
    es = 4 if dt == torch.float32 else 2
    k = torch.randn((B, Hkv, S, D), generator=g, device=DEV).to(dt)
    v = torch.randn((B, Hkv, S, D), generator=g, device=DEV).to(dt)
    n = int(S * (1 - ratio))
    bytes_ = B * Hkv * D * es * (S + 4 * n)
    if kind == "knorm":
        fn = lambda: N.knorm_compress(k, v, n)
    elif kind == "snapkv":
        q = torch.randn((B, Hq, W, D), generator=g, device=DEV).to(dt)
        ang = torch.rand((1, W, D), generator=g, device=DEV)
        c, s = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
        fn = lambda: N.snapkv_compress_rope(q, c, s, k, v, ks, n)
    else:   # ea, kernel-only (Q given)
        q = torch.randn((B, S - 4, Hq * D), generator=g, device=DEV).to(dt).view(B, S - 4, Hq, D).transpose(1, 2)
        bytes_ += B * D * es * (Hkv * S + Hq * S)
        def fn():
            mu, cov = N.ea_qstats(q, True)
            sc = N.ea_score(k, v, mu, cov, 4, True, 0.0)
            return N.gather_kv(k, v, N.topk_select(sc, n))
    try:
        t = timeit(fn)
    except Exception as e:   # noqa: BLE001
        print(f"{kind:7s} B={B} Hq={Hq:2d} Hkv={Hkv:2d} S={S:6d} D={D:3d} {str(dt).split('.')[-1]:8s} W={W:2d} ks={ks} r={ratio}: FAILED {type(e).__name__}: {str(e)[:120]}   {note}", flush=True)
        return
    print(f"{kind:7s} B={B} Hq={Hq:2d} Hkv={Hkv:2d} S={S:6d} D={D:3d} {str(dt).split('.')[-1]:8s} W={W:2d} ks={ks} r={ratio}: {t:8.1f} us  "
          f"{bytes_ / (t * 1e-6) / 8e12:.3f} of 8 TB/s   {note}", flush=True)
    del k, v
    torch.cuda.empty_cache()


def main():
    bf, f16, f32 = torch.bfloat16, torch.float16, torch.float32
    S = 32768
    print("# fused compress calls, HIP events around 30 back-to-back calls; fraction = SURVEY 8(d) algorithmic bytes / time / 8 TB/s")
    case("snapkv", 1, 32, 8, 131072, 128, bf, note="the benchmark's shape (hand-scheduled passes, cluster select)")
    case("snapkv", 1, 32, 8, S, 128, bf, note="hand-scheduled passes, shorter cache")
    case("snapkv", 1, 32, 8, S, 128, f16, note="f16 variant of the hand-scheduled passes")
    case("snapkv", 1, 8, 8, S, 128, bf, note="G = 1 (MHA): compiler-scheduled MFMA passes, 2 of 8 waves active")
    case("snapkv", 1, 16, 8, S, 128, bf, note="G = 2: compiler-scheduled MFMA passes")
    case("snapkv", 1, 24, 8, S, 128, bf, note="G = 3: compiler-scheduled MFMA passes")
    case("snapkv", 1, 64, 8, S, 128, bf, note="G = 8: two group-blocks, second column-sum slab")
    case("snapkv", 1, 128, 8, S, 128, bf, note="G = 16 (Llama-3.1-405B geometry): four group-blocks (generic kernels until round 6)")
    case("snapkv", 1, 96, 8, S, 128, bf, note="G = 12 (Mistral-Large geometry)")
    case("snapkv", 1, 32, 8, S, 96, bf, note="D = 96 (Phi-3-mini): compiler-scheduled MFMA passes on 256-byte LDS rows (round 6)")
    case("snapkv", 1, 32, 8, S, 256, bf, note="D = 256 (Gemma): compiler-scheduled MFMA passes, two-buffer ring of 64 KiB tiles (round 6)")
    case("snapkv", 1, 32, 8, S, 64, bf, note="D = 64: compiler-scheduled MFMA passes (round 6; before: generic kernels, 3206 us)")
    case("snapkv", 1, 32, 8, S, 128, f32, note="float32 model: generic kernels")
    case("snapkv", 1, 32, 8, S, 128, bf, W=32, note="window 32: hand-scheduled passes on a padded 64-row block (round 6; before: generic kernels, 3812 us)")
    case("snapkv", 1, 32, 8, S, 128, bf, ks=7, note="kernel_size 7: pooling launch + HIST1 cluster select")
    case("snapkv", 4, 32, 8, S, 128, bf, note="batch 4 (32 rows: one cluster launch, four resident rounds)")
    case("snapkv", 8, 32, 8, S, 128, bf, note="batch 8 (64 rows: the (chunk, row) passes)")
    case("knorm", 1, 32, 8, S, 128, bf, note="config 2's shape")
    case("knorm", 1, 32, 8, S, 64, bf, note="D = 64: norm kernel + select, no fused stream")
    case("knorm", 1, 32, 8, S, 128, f32, note="float32: norm kernel + select")
    case("knorm", 8, 32, 8, S, 128, bf, note="batch 8 (64 rows)")
    case("knorm", 1, 12, 12, 2048, 64, f32, note="config 1's shape (OPT-125m geometry)")
    case("ea", 1, 32, 8, S, 128, bf, ratio=0.7, note="kernel-only, MFMA statistics + triangular quadratic form")
    case("ea", 1, 32, 8, S, 64, bf, ratio=0.7, note="D = 64 (round 6: statistics over pairs of neighbouring heads + small-head quadratic form on the matrix cores; before: 1060 us)")
    case("ea", 1, 32, 32, S, 96, bf, ratio=0.7, note="D = 96 (Phi-3-mini, MHA; round 6: statistics as zero-padded heads of 128 + small-head quadratic form; before: 1081 us with generic statistics)")
    case("ea", 1, 32, 8, 131072, 64, bf, ratio=0.7, note="D = 64 at 128k")
    case("ea", 1, 32, 8, S, 256, bf, ratio=0.7, note="D = 256 (Gemma; refused until round 6, generic kernels 13.6 ms; now statistics over pairs of quarters + its own quadratic-form kernel)")
    case("ea", 1, 32, 8, 8192, 128, f32, ratio=0.7, note="float32: generic kernels")


if __name__ == "__main__":
    main()
