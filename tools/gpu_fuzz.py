#!/usr/bin/env python3
"""Random-geometry check of the kernels reworked in round 3 against the numpy oracle (test infrastructure: imports oracle/), on a GPU:
row norms / KeyDiff / CUR on the slot walks, ExpectedAttention's score chain (LDS-DMA pipeline of the quadratic form, one-pass value
norms + finalize), the one-pass gather + re-rotation, each over ragged lengths, strided views, group sizes and dtypes.

    python tools/gpu_fuzz.py [--rounds 24] [--seed 0]     -> one line per round, "fuzz ok" at the end (exit 1 on the first mismatch)
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


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    for it in range(args.rounds):
        dtype = ["bf16", "f16"][it % 2]
        dt = _inputs.torch_dtype(dtype)
        B = int(rs.choice([1, 1, 2]))
        Hkv = int(rs.choice([1, 2, 3, 8]))
        G = int(rs.choice([1, 2, 4, 4, 8]))
        S = int(rs.choice([rs.randint(70, 600), rs.randint(4096, 4400), rs.randint(4500, 12000), rs.randint(12000, 16500)]))
        if it % 8 == 7:   # round 6: long rows x many heads -- ExpectedAttention's logits take chunks of 8192 keys (more than 512 workgroups of 4096)
            B, Hkv, G, S = 1, 8, 4, int(rs.randint(66000, 80000))
        D = 128 if it % 8 == 7 else int(rs.choice([128, 128, 64, 96, 256]))   # round 6: ExpectedAttention's matrix-core paths for the other head sizes
        n_sink = int(rs.choice([0, 1, 4, 7]))
        kn = _inputs.round_to((rs.standard_normal((B, Hkv, S, D)) * rs.choice([0.3, 1.0, 2.0])).astype(np.float32), dtype)
        vn = _inputs.round_to(rs.standard_normal((B, Hkv, S, D)).astype(np.float32), dtype)
        strided = bool(rs.randint(2))
        if strided:   # [B, S, H, D] buffers seen as [B, H, S, D]
            k = torch.from_numpy(kn).to(DEV).to(dt).transpose(1, 2).contiguous().transpose(1, 2)
            v = torch.from_numpy(vn).to(DEV).to(dt).transpose(1, 2).contiguous().transpose(1, 2)
        else:
            k, v = torch.from_numpy(kn).to(DEV).to(dt), torch.from_numpy(vn).to(DEV).to(dt)
        msg = [f"round {it}: {dtype} B={B} Hkv={Hkv} G={G} S={S} D={D} sinks={n_sink} strided={strided}"]
        # row norms, KeyDiff, CUR
        e = rel(N.rownorm_score(k, -1.0).cpu().numpy(), O.knorm_score(kn))
        assert e <= 1e-5, ("rownorm", e)
        e = float(np.max(np.abs(N.keydiff_score(k).cpu().numpy() - O.keydiff_score(kn))))
        assert e <= 3e-6, ("keydiff", e)
        w = int(rs.choice([0, 4, 16, 5, 64]))
        lev = str(rs.choice(["key", "value", "kv_avg", "kv_product"]))
        got = N.cur_score(k, v, lev, w, min(n_sink, S)).cpu().numpy()
        want = O.cur_score(kn, vn, lev, w > 0, max(w, 1), min(n_sink, S))
        e = rel(got, want)
        assert e <= 5e-5, ("cur", lev, w, e)
        msg.append(f"cur[{lev},{w}] {e:.1e}")
        # ExpectedAttention score chain
        if S - n_sink >= 64:
            Hq = Hkv * G
            mu = (rs.standard_normal((B, Hq, D)) * 0.5).astype(np.float32)
            a = (rs.standard_normal((B, Hq, D, D)) * 0.03).astype(np.float32)
            cov = a @ a.transpose(0, 1, 3, 2)
            use_cov, use_vn = bool(rs.randint(4)), bool(rs.randint(4))
            eps = float(rs.choice([0.0, 0.01]))
            got = N.ea_score(k, v, torch.from_numpy(mu).to(DEV), torch.from_numpy(cov).to(DEV) if use_cov else None, n_sink, use_vn, eps).cpu().numpy()
            want = O.ea_score(kn, vn, mu, cov if use_cov else None, n_sink, use_vn, eps)
            e = rel(got[..., n_sink:], want[..., n_sink:])
            assert e <= 1e-3, ("ea_score", e)
            if n_sink:
                assert np.array_equal(got[..., :n_sink], np.broadcast_to(got[..., n_sink:].max() + np.float32(1.0), got[..., :n_sink].shape))
            msg.append(f"ea[cov={use_cov},vnorm={use_vn}] {e:.1e}")
        # query statistics (the matrix-core paths start at 4096 rows: pairs of heads / zero-padded heads / quarters for the head sizes off 128)
        Hq_s = Hkv * G
        if 4096 <= S and B * Hq_s * S * D * D <= 1.2e10:
            qn = _inputs.round_to((rs.standard_normal((B, Hq_s, S, D)) * np.exp(0.4 * rs.standard_normal((1, Hq_s, 1, D))) + rs.standard_normal((1, Hq_s, 1, D))).astype(np.float32), dtype)
            mu_w, cov_w = O.ea_query_stats(qn, True)
            qt = torch.from_numpy(np.ascontiguousarray(qn.transpose(0, 2, 1, 3))).to(DEV).to(dt).transpose(1, 2) if rs.randint(3) else torch.from_numpy(qn).to(DEV).to(dt)
            mu_g, cov_g = N.ea_qstats(qt, True)
            dd = np.sqrt(np.einsum("bhii->bhi", cov_w))
            e = float(np.max(np.abs(cov_g.cpu().numpy() - cov_w) / (dd[..., :, None] * dd[..., None, :])))
            assert e <= 1e-3 and np.abs(mu_g.cpu().numpy() - mu_w).max() <= 1e-5 * np.abs(mu_w).max() + 1e-6, ("ea_qstats", e)
            msg.append(f"qstats {e:.1e}")
        # gather + re-rotation in one pass == the oracle's gather then rerotate
        n = int(rs.randint(1, S + 1))
        pos = np.stack([np.sort(rs.choice(S, n, replace=False)) for _ in range(B * Hkv)]).reshape(B, Hkv, n).astype(np.int32)
        inv = (10000.0 ** (-np.arange(0, D, 2, dtype=np.float32) / D)).astype(np.float32)
        k1, v1 = N.gather_kv_rerotate(k, v, torch.from_numpy(pos).to(DEV), torch.from_numpy(inv).to(DEV))
        kg, vg = O.gather_kv(kn, vn, pos)
        assert np.array_equal(v1.float().cpu().numpy(), vg)
        want = O.rerotate_keys(kg, pos, inv, dtype)
        got = k1.float().cpu().numpy()
        bad = float(np.mean(got != want))
        assert bad < 4e-3, ("gather_rerotate", bad)   # numpy's cos / sin and the kernel's differ in the last bit of a few angles
        msg.append(f"rerotate n={n} differ {bad:.1e}")
        print("  ".join(msg), flush=True)
    print("fuzz ok", flush=True)


if __name__ == "__main__":
    main()
