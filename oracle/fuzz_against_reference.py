#!/usr/bin/env python3
"""Fuzz the numpy oracle against the REAL reference on random small geometries (beyond the committed golden cases).
Test infrastructure only; runs in the build container (needs /root/reference, see gen_golden.py for the shims):

    PYTHONDONTWRITEBYTECODE=1 python oracle/fuzz_against_reference.py [n_rounds] [seed]

Every round draws a geometry (B, H, G, S, D, window, kernel, ...), builds a float32 LlamaAttention with seeded weights and
compares, per scorer, the reference press's ``score()`` with the oracle chain on the same inputs (scores rtol 2e-4; LagKV
ranks through the rank-swap tolerance of tests/_inputs.py).  Prints one line per round; exits non-zero on a mismatch.
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def main(argv):
    import gen_golden
    gen_golden._install_shims()
    import kvpress as R
    import numpy as np
    import torch

    import _inputs
    import kvpress_oracle as O

    n_rounds = int(argv[0]) if argv else 20
    rs = np.random.RandomState(int(argv[1]) if len(argv) > 1 else 0)
    bad = 0
    for it in range(n_rounds):
        D = int(rs.choice([6, 8, 16, 32, 64]))
        G = int(rs.choice([1, 2, 4]))
        H = int(rs.choice([1, 2, 3]))
        S = int(rs.randint(40, 400))
        W = int(rs.randint(1, min(33, S // 2)))
        ks = int(rs.choice([1, 3, 5, 7]))
        lag = int(rs.randint(4, max(5, S // 4)))
        name = f"fuzz{it}"
        _inputs.CASES[name] = dict(kind="snapkv", B=int(rs.choice([1, 2])), H=H, G=G, S=S, D=D, dtype="f32", data=str(rs.choice(["A", "B"])),
                                   seed=int(rs.randint(1 << 20)), W=W, ks=ks)
        try:
            s = _inputs.make_case(name)
        finally:
            del _inputs.CASES[name]
        att, rot, hidden, pe = _inputs.build_llama_attention(s, torch.float32)
        keys, values = torch.from_numpy(s["keys"]), torch.from_numpy(s["values"])
        kwargs = {"position_embeddings": pe}
        cos, sin = pe[0].numpy(), pe[1].numpy()
        q = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos, sin, s["Hq"], D, W)
        checks = []
        with torch.no_grad():
            def ref(press):
                return press.score(att, hidden, keys, values, None, kwargs).numpy()
            checks.append(("knorm", ref(R.KnormPress(0.5)), O.knorm_score(s["keys"]), 1e-5, 1e-30))
            checks.append(("snapkv", ref(R.SnapKVPress(0.5, window_size=W, kernel_size=ks))[..., :-W], O.snapkv_score(q, s["keys"], ks)[..., :-W], 2e-4, 1e-30))
            checks.append(("keydiff", ref(R.KeyDiffPress(0.5)), O.keydiff_score(s["keys"]), 0, 3e-6))
            lev = str(rs.choice(["key", "value", "kv_avg", "kv_product"]))
            loc, win, sinks = bool(rs.rand() < 0.7), int(rs.randint(2, 20)), int(rs.randint(0, 6))
            checks.append((f"cur/{lev}", ref(R.CURPress(0.5, num_sinks=sinks, leverage_type=lev, use_local_approximation=loc, local_window_size=win)),
                           O.cur_score(s["keys"], s["values"], lev, loc, win, sinks), 2e-4, 1e-30))
            q1 = O.snapkv_window_queries(s["hidden"], s["wq"], None, cos, sin, s["Hq"], D, 1)
            checks.append(("tova", ref(R.TOVAPress(0.5))[..., :-1], O.tova_score(q1, s["keys"])[..., :-1], 2e-4, 1e-30))
            fin = R.FinchPress(0.5, normalize_scores=bool(rs.rand() < 0.5))
            fin.window_size = W
            checks.append(("finch", ref(fin)[..., :-W], O.finch_score(q, s["keys"], fin.normalize_scores)[..., :-W], 2e-4, 1e-30))
            n_sink = int(rs.randint(0, 5))
            if S > n_sink + 2:
                ea = R.ExpectedAttentionPress(0.5, n_future_positions=int(rs.randint(1, 100)), n_sink=n_sink, use_covariance=bool(rs.rand() < 0.7),
                                              use_vnorm=bool(rs.rand() < 0.7), epsilon=float(rs.choice([0.0, 0.01])))
                h = s["hidden"][:, n_sink:].astype(np.float64)
                qq = (h @ s["wq"].astype(np.float64).T).reshape(s["B"], -1, s["Hq"], D).transpose(0, 2, 1, 3)
                mu, cov = O.ea_query_stats(qq, ea.use_covariance)
                c, si = rot(torch.zeros(1), torch.arange(S, S + ea.n_future_positions)[None])
                mu, cov = O.ea_avg_rope(mu, cov, c[0].numpy(), si[0].numpy())
                checks.append(("ea", ref(ea)[..., n_sink:], O.ea_score(s["keys"], s["values"], mu, cov, n_sink, ea.use_vnorm, ea.epsilon)[..., n_sink:], 3e-4, 1e-30))
            lg = R.LagKVPress(0.5, n_sink=int(rs.randint(0, 5)), lag_size=lag, cross_scoring=bool(rs.rand() < 0.5))
            got, want = O.lagkv_score(s["keys"], s["values"], lg.n_sink, lag, lg.cross_scoring), ref(lg)
            try:
                _inputs.assert_lag_scores_close(got, want, dict(S=S, n_sink=lg.n_sink, lag=lag, cross=lg.cross_scoring), "lagkv")
                ok_lag = True
            except AssertionError as e:
                ok_lag = str(e)
        msgs = []
        for what, r_, o_, rtol, atol in checks:
            try:
                np.testing.assert_allclose(o_, r_, rtol=rtol, atol=atol)
            except AssertionError as e:
                msgs.append(f"{what}: {str(e).splitlines()[3] if len(str(e).splitlines()) > 3 else e}")
        if ok_lag is not True:
            msgs.append(f"lagkv: {ok_lag}")
        bad += bool(msgs)
        print(f"round {it}: B={s['B']} H={H} G={G} S={S} D={D} W={W} ks={ks} lag={lag} -> {'OK' if not msgs else msgs}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
