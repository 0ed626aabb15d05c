#!/usr/bin/env python3
"""Random-shape check of the SnapKV window-attention passes after round 6 widened them (test infrastructure: imports oracle/), on a GPU:
windows of 1 .. 300 rows, head sizes 64 / 96 / 128 / 256, GQA groups 1 .. 8, bf16 / f16, ragged lengths down to W + 1, batches, strided views
of K (a longer cache sliced, a [B, S, H, D] layout transposed), kernel sizes 1 .. 7 -- scores against the float64 numpy oracle (1e-3), the
pad columns, and the fused compress against score -> select -> gather (bit-identical).

    python tools/snapkv_shape_fuzz.py [--rounds 60] [--seed 0]   -> one line per round, "snapkv shape fuzz ok" at the end (exit 1 on a mismatch)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _inputs  # noqa: E402
from kvpress_amd import _native as N  # noqa: E402
from oracle import kvpress_oracle as O  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    bad = 0
    for it in range(args.rounds):
        D = int(rs.choice([64, 96, 128, 256]))
        G = int(rs.choice([1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 12, 16]))
        H = int(rs.choice([1, 2, 3]))
        B = int(rs.choice([1, 1, 2]))
        W = int(rs.choice([1, 2, 7, 31, 32, 33, 63, 64, 64, 65, 100, 127, 128, 129, 200, 300]))
        S = int(W + rs.choice([1, 2, 63, 129, 500, 1500, 4100, 9000])) + int(rs.randint(0, 60))
        ks = int(rs.choice([1, 3, 5, 5, 7]))
        dt = str(rs.choice(["bf16", "f16"]))
        tdt = torch.bfloat16 if dt == "bf16" else torch.float16
        layout = str(rs.choice(["plain", "sliced", "bshd"]))
        q = _inputs.round_to((rs.standard_normal((B, H * G, W, D)) * rs.choice([0.5, 1.0, 2.0])).astype(np.float32), dt)
        k = rs.standard_normal((B, H, S, D)).astype(np.float32)
        if rs.rand() < 0.5:
            k[:, :, : max(1, S // 4)] *= 2.5
        k = _inputs.round_to(k, dt)
        v = _inputs.round_to(rs.standard_normal((B, H, S, D)).astype(np.float32), dt)
        want = O.snapkv_score(q, k, ks)
        qd = torch.from_numpy(q).to(DEV, tdt)
        if layout == "plain":
            kd = torch.from_numpy(k).to(DEV, tdt)
        elif layout == "sliced":   # a view of a longer cache
            big = torch.zeros((B, H, S + 16, D), dtype=tdt, device=DEV)
            big[:, :, 8:8 + S] = torch.from_numpy(k).to(DEV, tdt)
            kd = big[:, :, 8:8 + S]
        else:                      # [B, S, H, D] storage, viewed as [B, H, S, D]
            kd = torch.from_numpy(np.ascontiguousarray(k.transpose(0, 2, 1, 3))).to(DEV, tdt).transpose(1, 2)
        got = N.snapkv_score(qd, kd, ks).cpu().numpy()
        err = float(np.max(np.abs(got[..., :-W] - want[..., :-W]) / np.maximum(np.abs(want[..., :-W]), 1e-30))) if S > W else 0.0
        ok = err <= 1e-3 and np.all(got[..., -W:] == np.float32(got[..., :-W].max()) + np.float32(1.0))
        # fused compress == modular, through the rope entry with identity tables
        vd = torch.from_numpy(v).to(DEV, tdt)
        ones, zeros = torch.ones((1, W, D), dtype=tdt, device=DEV), torch.zeros((1, W, D), dtype=tdt, device=DEV)
        n = int(rs.choice([max(1, W - 1), W, min(S, W + 1), S // 2, S - 1, S]))
        n = max(1, min(n, S))
        sc = N.snapkv_score_rope(qd, ones, zeros, kd, ks)
        ko, vo = N.snapkv_compress_rope(qd, ones, zeros, kd, vd, ks, n)
        wk, wv = N.gather_kv(kd, vd, N.topk_select(sc, n))
        fused_ok = bool(torch.equal(ko, wk) and torch.equal(vo, wv))
        print(f"round {it}: {dt} B={B} Hkv={H} G={G} S={S} D={D} W={W} ks={ks} {layout:6s} n={n}: score err {err:.1e}  fused==modular {fused_ok}"
              + ("" if ok and fused_ok else "   <-- MISMATCH"), flush=True)
        bad += not (ok and fused_ok)
        if bad >= 5:
            break
    N.async_error_check()
    if bad:
        print(f"snapkv shape fuzz: {bad} MISMATCHING round(s)")
        sys.exit(1)
    print(f"snapkv shape fuzz ok ({args.rounds} rounds, seed {args.seed})")


if __name__ == "__main__":
    main()
