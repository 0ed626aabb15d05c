#!/usr/bin/env python3
"""A/B of the walk shape of the streaming reductions (GPU): ||x|| rows (rownorm_vec_kernel / rownorm_slot_kernel), CUR's two energies in
one launch, KeyDiff's anchor / score passes and the ExpectedAttention query statistics' partial count, at 8 x 131072 x 128 bf16.
Per call: HIP events around the call; "cold" = a 512 MB device copy between calls (what the gather leaves behind in the bench loop),
"warm" = back to back.

    python tools/stream_lab.py [--reps 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from kvpress_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def knobs(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    N.tuning_reload()


def timed(fn, reps, evict=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    ts = []
    for _ in range(reps):
        if evict is not None:
            evict()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--knorm", action="store_true", help="only the fused Knorm compress (run once per library: KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_kn8.so)")
    args = ap.parse_args()
    if args.knorm:
        g = torch.Generator(device=DEV)
        g.manual_seed(0)
        print("knorm_compress (cluster select with the norm stream inside + gather), lib", os.environ.get("KVPRESS_HIP_LIB", "default"))
        for S in (32768, 131072):
            k = torch.randn((1, 8, S, 128), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
            v = torch.randn((1, 8, S, 128), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
            fn = lambda: N.knorm_compress(k, v, S // 2)
            print(f"  S={S}: back to back {timed(fn, 100):7.1f} us per call (events around each call)", flush=True)
        return
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    S = 131072
    k = torch.randn((1, 8, S, 128), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn((1, 8, S, 128), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    big_a = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    big_b = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    evict = lambda: big_b.copy_(big_a)

    def row(name, fn, ref=None):
        out = fn()
        same = "" if ref is None else ("  bits==base" if torch.equal(out, ref) else f"  max diff {float((out - ref).abs().max()):.3e}")
        print(f"  {name:44s} cold {timed(fn, args.reps, evict):7.1f} us   warm {timed(fn, args.reps):7.1f} us{same}", flush=True)
        return out

    print(f"rownorm_score K [1, 8, {S}, 128] bf16 (268 MB)")
    knobs(KVP_RN_SLOT=0)
    base = row("strided, 256 thr x 8 / CU (base)", lambda: N.rownorm_score(k, -1.0))
    for slot, thr, wgs, nt in ((0, 256, 8, 1), (1, 1024, 1, 0), (1, 1024, 1, 1), (1, 256, 4, 1), (1, 512, 2, 1), (1, 1024, 2, 1)):
        knobs(KVP_RN_SLOT=slot, KVP_RN_THREADS=thr, KVP_RN_WGS=wgs, KVP_RN_NT=nt)
        row(f"{'slot' if slot else 'strided'}, {thr} thr x {wgs} / CU{', nt' if nt else ''}", lambda: N.rownorm_score(k, -1.0), base)

    print("cur_score (K and V energies in one launch + normalize + combine; 537 MB)")
    knobs(KVP_RN_SLOT=0, KVP_RN_NT=0)
    cur = lambda: N.cur_score(k, v, "kv_product", 16, 4)
    base = row("strided (base)", cur)
    for slot, thr, wgs, nt in ((0, 256, 8, 1), (1, 1024, 1, 0), (1, 1024, 1, 1), (1, 1024, 2, 1), (1, 256, 8, 1)):
        knobs(KVP_RN_SLOT=slot, KVP_RN_THREADS=thr, KVP_RN_WGS=wgs, KVP_RN_NT=nt)
        row(f"{'slot' if slot else 'strided'}, {thr} thr x {wgs} / CU{', nt' if nt else ''}", cur, base)
    knobs(KVP_RN_SLOT=None, KVP_RN_THREADS=None, KVP_RN_WGS=None, KVP_RN_NT=None)

    print("keydiff_score (anchor pass + reduce + score pass; 2 x 268 MB)")
    knobs(KVP_KD_SLOT=0)
    kd = lambda: N.keydiff_score(k)
    base = row("strided, 256 thr x 8 / CU (base)", kd)
    for slot, thr, wgs, nt in ((0, 256, 8, 1), (0, 256, 8, 3), (1, 1024, 1, 0), (1, 1024, 1, 1), (1, 1024, 1, 2), (1, 1024, 1, 3), (1, 256, 4, 3)):
        knobs(KVP_KD_SLOT=slot, KVP_KD_THREADS=thr, KVP_KD_WGS=wgs, KVP_KD_NT=nt)
        row(f"{'slot' if slot else 'strided'}, {thr} thr x {wgs} / CU, nt mask {nt}", kd, base)
    knobs(KVP_KD_SLOT=None, KVP_KD_THREADS=None, KVP_KD_WGS=None, KVP_KD_NT=None)

    print(f"ea_qstats q [1, 32, {S}, 128] bf16 as the model lays it out (1.07 GB): partials per head, LDS ring, nt")
    q = torch.randn((1, S, 32 * 128), generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16).view(1, S, 32, 128).transpose(1, 2)
    base = None
    for nc, ring, nt in ((32, 3, 0), (32, 3, 1), (16, 3, 0), (8, 3, 0), (8, 3, 1), (8, 6, 0), (8, 6, 1), (16, 6, 0), (4, 6, 0)):
        knobs(KVP_EA_QCHUNKS=nc, KVP_EA_QRING=ring, KVP_EA_QNT=nt)
        fn = lambda: N.ea_qstats(q, True)[1]
        out = row(f"<= {nc} chunks per head, ring {ring}{', nt' if nt else ''}", fn, base)
        base = out if base is None else base
    knobs(KVP_EA_QCHUNKS=None, KVP_EA_QRING=None, KVP_EA_QNT=None)


if __name__ == "__main__":
    main()
