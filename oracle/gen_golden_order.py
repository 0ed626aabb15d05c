#!/usr/bin/env python3
"""Fixtures of the reference's OWN ``compress()`` output tensors -- K', V' in the order `scores.topk(n_kept)` produces
(kvpress/presses/scorer_press.py:95-100: descending score) -- for the reference-exact mode ``ScorerPress.kept_order = "score"``.

Runs the REAL reference (/root/reference) in float32 mode ("O32": module and tensors `.float()` copies of the bf16 / f16
representable inputs of tests/_inputs.py) and stores, per case and ratio:
  ko_<i>, vo_<i>   the reference's compress() outputs [B,H,n,D]: float32, or the bit patterns (uint16) of the case's bf16 / f16 dtype
                   (exact: every element of the float32 run's output is a copy of an input value)
  idx_<i>          the reference's topk indices [B,H,n] int32 in ITS order (descending score)
  val_<i>          the kept scores in that order [B,H,n] float32, followed by the largest dropped score ([B,H,n+1]): where two
                   neighbours are closer than an implementation's score error -- or EQUAL: SnapKV's window and ExpectedAttention's
                   sinks all carry the pad constant max + 1, and torch.topk's order among equal scores is unspecified -- the order
                   inside that run is not defined by the reference; everywhere else it is, and the tensors must match exactly
  gap_<i>          per (b,h): the smallest relative distance between two neighbouring scores among the kept ones and the first
                   dropped one -- the margin an implementation's float32 score error has before the ORDER may legitimately differ
Test infrastructure only.  The GPU box only sees the committed tests/golden/order_*.npz.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_order.py [case ...]
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

ORDER_CASES = ["kn_readme", "kn_tiny_d6", "kn_d96_bf16", "sk_257_A", "sk_257_B", "sk_f16_d64", "ea_257_A", "ea_nocov"]


def main(argv):
    from gen_golden import _install_shims

    _install_shims()
    import numpy as np
    import torch
    from kvpress import ExpectedAttentionPress, KnormPress, SnapKVPress  # the reference

    import _inputs

    outdir = os.path.join(REPO, "tests", "golden")
    for name in argv or ORDER_CASES:
        s = _inputs.make_case(name)
        ratios = tuple(r for r in (0.25, 0.5, 0.8) if int(s["S"] * (1 - r)) >= 1)

        def make_press(r):
            if s["kind"] == "knorm":
                return KnormPress(compression_ratio=r)
            if s["kind"] == "snapkv":
                return SnapKVPress(compression_ratio=r, window_size=s["W"], kernel_size=s["ks"])
            return ExpectedAttentionPress(compression_ratio=r, n_future_positions=s["n_future"], n_sink=s["n_sink"],
                                          use_covariance=s["use_covariance"], use_vnorm=s["use_vnorm"], epsilon=s["epsilon"])

        att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
        keys = torch.from_numpy(s["keys"]).float()
        values = torch.from_numpy(s["values"]).float()
        kwargs = {"position_embeddings": pe}
        out = {"ratios": np.asarray(ratios, dtype=np.float64)}
        with torch.no_grad():
            for i, r in enumerate(ratios):
                p = make_press(r)
                ko, vo = p.compress(att, hidden, keys, values, None, kwargs)          # THE tensors the reference stores in the cache
                sc = p.score(att, hidden, keys, values, None, kwargs)
                n = ko.shape[2]
                top = sc.topk(n, dim=-1)
                e = top.indices.unsqueeze(-1).expand(-1, -1, -1, keys.shape[-1])
                assert torch.equal(ko, keys.gather(2, e)) and torch.equal(vo, values.gather(2, e))
                # margins: kept scores in descending order followed by the largest dropped score
                dropped = sc.scatter(-1, top.indices, float("-inf")).amax(-1, keepdim=True)
                chain = torch.cat([top.values, dropped], dim=-1).double()
                finite = torch.isfinite(chain[..., 1:])
                rel = (chain[..., :-1] - chain[..., 1:]).abs() / chain[..., :-1].abs().clamp_min(1e-300)
                rel = torch.where(finite, rel, torch.full_like(rel, float("inf")))
                # stored in the case's own 16-bit dtype as bit patterns (exact: the float32 run only ever copies input values)
                dt = _inputs.torch_dtype(s["dtype"])
                for nm, t in (("ko", ko), ("vo", vo)):
                    assert torch.equal(t.to(dt).float(), t)
                    out[f"{nm}_{i}"] = t.numpy().astype(np.float32) if dt == torch.float32 else t.to(dt).view(torch.int16).numpy().view(np.uint16)
                out[f"idx_{i}"] = top.indices.numpy().astype(np.int32)
                out[f"val_{i}"] = chain.float().numpy()
                out[f"gap_{i}"] = rel.amin(-1).numpy()
        path = os.path.join(outdir, f"order_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"order_{name}: {os.path.getsize(path)} bytes; min relative gaps per ratio: "
              f"{[float(out[f'gap_{i}'].min()) for i in range(len(ratios))]}")


# ---- chains (round 6): an order-dependent press behind a ScorerPress -----------------------------------------------------------------
# composed_press.py:56-62 runs the presses' hooks one after the other on the SAME cache layer, so the second press sees the survivors of
# the first in `scores.topk` order (descending score).  The fixtures hold the reference's final K' / V' after p1.compress -> p2.compress
# (float32 mode) for one chain with an exactly defined result (Knorm -> StreamingLLM: sinks + most recent OF THE SCORE-ORDERED cache) and
# one with a float score in the second stage (Knorm -> SnapKV: the window are the last W rows of the score-ordered cache), with the
# relative gap of the second stage's scores at its threshold.  Row order inside the result: the second stage's topk order, which for
# StreamingLLM's 0/1 scores is a pure tie (unspecified by torch) -- the tests compare the rows as SETS per (batch, head).
CHAIN_CASES = {
    "chain_kn_stream": ("kn_tiny_d6", 0.25, ("StreamingLLMPress", dict(compression_ratio=0.5, n_sink=3))),
    "chain_kn_snap": ("sk_257_A", 0.25, ("SnapKVPress", dict(compression_ratio=0.5))),
}


def main_chains():
    from gen_golden import _install_shims

    _install_shims()
    import numpy as np
    import torch
    import kvpress as K  # the reference

    import _inputs

    outdir = os.path.join(REPO, "tests", "golden")
    for cname, (case, r1, (p2name, p2kw)) in CHAIN_CASES.items():
        s = _inputs.make_case(case)
        att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
        keys = torch.from_numpy(s["keys"]).float()
        values = torch.from_numpy(s["values"]).float()
        kwargs = {"position_embeddings": pe}
        if p2name == "SnapKVPress":
            p2kw = dict(p2kw, window_size=s["W"], kernel_size=s["ks"])
        p1, p2 = K.KnormPress(compression_ratio=r1), getattr(K, p2name)(**p2kw)
        with torch.no_grad():
            k1, v1 = p1.compress(att, hidden, keys, values, None, kwargs)
            k2, v2 = p2.compress(att, hidden, k1, v1, None, kwargs)
            sc2 = p2.score(att, hidden, k1, v1, None, kwargs).double()
        n2 = k2.shape[2]
        top = sc2.topk(n2 + 1, dim=-1).values if n2 < sc2.shape[-1] else None
        gap = ((top[..., -2] - top[..., -1]) / top[..., -2].abs().clamp_min(1e-300)).numpy() if top is not None else np.full(k2.shape[:2], np.inf)
        dt = _inputs.torch_dtype(s["dtype"])
        out = {"gap2": gap}
        for nm, t_ in (("ko", k2), ("vo", v2)):
            assert torch.equal(t_.to(dt).float(), t_)
            out[nm] = t_.numpy().astype(np.float32) if dt == torch.float32 else t_.to(dt).view(torch.int16).numpy().view(np.uint16)
        path = os.path.join(outdir, f"order_{cname}.npz")
        np.savez_compressed(path, **out)
        print(f"order_{cname}: {tuple(k2.shape)}; {os.path.getsize(path)} bytes; min gap at the second threshold {float(np.min(gap)):.3e}")


if __name__ == "__main__":
    if sys.argv[1:] == ["chains"]:
        main_chains()
    else:
        main(sys.argv[1:])
