#!/usr/bin/env python3
"""Golden outputs of the reference's ThinKPress (kvpress/presses/think_press.py) -> tests/golden/<think case>.npz.
Test infrastructure only; needs /root/reference (see gen_golden.py for the shims).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_think.py

Per case: qwin_f32 (the RoPE'd window queries of the float32 run) and, per ratio i and run (f32 / nat = the case dtype),
pruned_<run>_<i> [B,H,n]: the channels the reference zeroed, ascending (read off its output keys: the inputs have no
exact zeros).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main(argv):
    import gen_golden
    gen_golden._install_shims()
    import numpy as np
    import torch
    from kvpress import ThinKPress

    import _inputs

    outdir = os.path.join(REPO, "tests", "golden")
    for name in (argv or list(_inputs.THINK_CASES)):
        s = _inputs.make_think_case(name)
        assert not (s["keys"] == 0).any()
        out = {"ratios": np.asarray(s["ratios"], dtype=np.float64)}
        for mode, dt in (("f32", torch.float32), ("nat", _inputs.torch_dtype(s["dtype"]))):
            att, rot, hidden, pe = _inputs.build_llama_attention(s, dt)
            values = torch.from_numpy(s["values"]).to(dt)
            kwargs = {"position_embeddings": pe, "hidden_states": hidden}
            with torch.no_grad():
                for i, r in enumerate(s["ratios"]):
                    press = ThinKPress(key_channel_compression_ratio=r, window_size=s["W"])
                    if mode == "f32" and i == 0:
                        out["qwin_f32"] = press.compute_window_queries(att, hidden, pe).numpy()
                    keys = torch.from_numpy(s["keys"]).to(dt).clone()   # compress works in place
                    ko, vo = press.compress(att, hidden, keys, values, None, kwargs)
                    assert ko is keys and vo is values                      # in place (think_press.py:82)
                    zero = (ko == 0).all(dim=2)                             # [B,H,D]
                    n = int(s["D"] * r)
                    assert (zero.sum(-1) == n).all()
                    out[f"pruned_{mode}_{i}"] = torch.nonzero(zero)[:, 2].view(s["B"], s["H"], n).numpy().astype(np.int32)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), {k: v.shape for k, v in out.items() if k.startswith("pruned_")})


if __name__ == "__main__":
    main(sys.argv[1:])
