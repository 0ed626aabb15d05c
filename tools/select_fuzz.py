#!/usr/bin/env python3
"""Random-row check of the cluster select (round 5: its two-hop form and the three-round form it declines to) against the numpy oracle
(test infrastructure: imports oracle/), on a GPU: random row counts, lengths over the kernel's whole range, k, value distributions that
exercise both forms (flat BASELINE-like rows, Gaussian, norm-like, heavy ties, a few distinct values, sorted stretches, infinities,
mixtures of scales), k smallest, strided rows, and the fused Knorm compress (its loader computes the keys from K).

    python tools/select_fuzz.py [--rounds 60] [--seed 0]     -> one line per round, "select fuzz ok" at the end (exit 1 on a mismatch)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from kvpress_amd import _native as N  # noqa: E402
from oracle import kvpress_oracle as O  # noqa: E402

DEV = "cuda:0"


def rows(rs, R, S):
    kind = int(rs.randint(9))
    if kind == 0:
        x = 2.0 ** rs.randint(-20, 5) * (1 + 0.05 * rs.standard_normal((R, S)))
    elif kind == 1:
        x = rs.standard_normal((R, S)) * 10.0 ** rs.randint(-3, 4)
    elif kind == 2:
        x = -np.sqrt(rs.chisquare(int(rs.choice([16, 64, 128])), size=(R, S)))
    elif kind == 3:   # heavy ties: rounded to few mantissa bits
        x = (rs.standard_normal((R, S)).astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000 if rs.randint(2) else 0xFFF00000)).view(np.float32)
    elif kind == 4:
        x = rs.randint(0, int(rs.choice([3, 40, 500, 5000])), size=(R, S))
    elif kind == 5:   # sorted stretches
        x = rs.standard_normal((R, S))
        cut = int(rs.randint(1, S))
        x[:, :cut] = np.sort(x[:, :cut], axis=1)[:, :: (1 if rs.randint(2) else -1)]
    elif kind == 6:
        x = rs.standard_normal((R, S))
        x[:, rs.choice(S, S // 50, replace=False)] = np.inf
        x[:, rs.choice(S, S // 70, replace=False)] = -np.inf
    elif kind == 7:   # two populations far apart
        x = np.where(rs.random_sample((R, S)) < rs.uniform(0.05, 0.95), rs.standard_normal((R, S)) * 1e-4, 5.0 + rs.standard_normal((R, S)))
    else:             # consecutive keys
        x = (np.float32(rs.uniform(0.5, 2.0)).view(np.uint32) + rs.randint(0, int(rs.choice([50, 300, 3000])), size=(R, S)).astype(np.uint32)).view(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    for it in range(args.rounds):
        R = int(rs.randint(1, 9)) if rs.randint(3) else int(rs.randint(9, 41))   # round 6: up to 32 rows per cluster launch, more in several
        S = int(rs.choice([rs.randint(16385, 20000), rs.randint(20000, 40000), rs.randint(40000, 70000), rs.randint(70000, 140000), rs.randint(140000, 262145), 131072]))
        x, kind = rows(rs, R, S)
        k = int(rs.choice([1, S - 1, S // 2, max(1, int(S * rs.uniform(0.02, 0.98))), max(1, int(S * rs.uniform(0.0, 1.0)))]))
        smallest = bool(rs.randint(4) == 0)
        pad = int(rs.choice([0, 0, 3, 5]))
        if pad:   # unaligned / strided rows: a view into a wider buffer
            buf = torch.zeros((R, S + 2 * pad), dtype=torch.float32, device=DEV)
            buf[:, pad:pad + S] = torch.from_numpy(x).to(DEV)
            t = buf[:, pad:pad + S]
        else:
            t = torch.from_numpy(x).to(DEV)
        flags = N.ORDER_POSITION | (N.TOPK_SMALLEST if smallest else 0)
        for rep in range(2):   # twice through the cached self-cleaning workspace
            got = N.topk_select(t, k, flags).cpu().numpy()
            want = O.topk_select(-x if smallest else x, k)
            assert np.array_equal(got, want), f"round {it} rep {rep}: R={R} S={S} k={k} kind={kind} smallest={smallest} pad={pad}: MISMATCH"
        msg = f"round {it}: R={R} S={S} k={k} kind={kind} smallest={smallest} pad={pad} ok"
        if it % 4 == 0:   # fused Knorm compress: keys from K inside the select's loader
            dt = torch.bfloat16 if it % 8 == 0 else torch.float16
            Sk = int(rs.choice([rs.randint(16385, 40000), 32768, rs.randint(40000, 140000)]))
            nk = max(1, int(Sk * rs.uniform(0.1, 0.9)))
            Bk, Hk = int(rs.choice([1, 1, 2, 3, 4])), int(rs.choice([8, 8, 2, 5]))   # round 6: batches (B * H rows in one cluster launch up to 32)
            kk = (torch.randn((Bk, Hk, Sk, 128), device=DEV) * float(rs.choice([0.3, 1.0, 3.0]))).to(dt)
            vv = torch.randn((Bk, Hk, Sk, 128), device=DEV).to(dt)
            ko, vo = N.knorm_compress(kk, vv, nk)
            sc = N.rownorm_score(kk, -1.0)                    # the same norms through the stand-alone kernel (tested against the oracle elsewhere)
            idx = torch.from_numpy(O.topk_select(sc.cpu().numpy().reshape(Bk * Hk, Sk), nk).reshape(Bk, Hk, nk)).to(DEV).long()
            assert torch.equal(ko, torch.gather(kk, 2, idx[..., None].expand(-1, -1, -1, 128))), f"round {it}: knorm_compress keys B={Bk} H={Hk} S={Sk} n={nk}"
            assert torch.equal(vo, torch.gather(vv, 2, idx[..., None].expand(-1, -1, -1, 128))), f"round {it}: knorm_compress values B={Bk} H={Hk} S={Sk} n={nk}"
            msg += f"  knorm_compress[{str(dt)[6:]} B={Bk} H={Hk} S={Sk} n={nk}] ok"
        torch.cuda.synchronize()
        N.async_error_check()
        print(msg, flush=True)
    print("select fuzz ok", flush=True)


if __name__ == "__main__":
    main()
