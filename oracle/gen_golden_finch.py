#!/usr/bin/env python3
"""Golden outputs of the reference's FinchPress (kvpress/presses/finch_press.py) -> tests/golden/<finch case>.npz.
Test infrastructure only; needs /root/reference (see gen_golden.py for the shims).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_finch.py

Per case: scores_f32 [B,H,S] and qwin_f32 (the RoPE'd question queries) of the float32 run; per ratio i and run
(f32 / nat = the case dtype): pos_<run>_<i> the kept positions, sorted (recovered from a VALUE tensor that stores each
token's position) and, for the re-rotating variant, kout_<run>_<i> the re-rotated keys (float32 storage).
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
    import kvpress
    import numpy as np
    import torch
    from kvpress.utils import get_prerope_query_states
    from transformers.models.llama.modeling_llama import rotate_half

    import _inputs

    outdir = os.path.join(REPO, "tests", "golden")
    for name in (argv or list(_inputs.FINCH_CASES)):
        s = _inputs.make_finch_case(name)
        out = {"ratios": np.asarray(s["ratios"], dtype=np.float64)}
        for mode, dt in (("f32", torch.float32), ("nat", _inputs.torch_dtype(s["dtype"]))):
            att, rot, hidden, pe = _inputs.build_llama_attention(s, dt)
            keys = torch.from_numpy(s["keys"]).to(dt)
            posv = torch.arange(s["S"], dtype=torch.float32)[None, None, :, None].expand(s["B"], s["H"], s["S"], s["D"]).contiguous()
            kwargs = {"position_embeddings": pe}
            with torch.no_grad():
                if mode == "f32":
                    W = s["W"]
                    q = get_prerope_query_states(att, hidden[:, -W:])
                    c, si = pe[0][:, -W:], pe[1][:, -W:]
                    out["qwin_f32"] = ((q * c.unsqueeze(1)) + (rotate_half(q) * si.unsqueeze(1))).numpy()
                    out["scores_f32"] = _inputs.make_finch_press(kvpress, s, 0.5).score(att, hidden, keys, posv, None, kwargs).float().numpy()
                for i, r in enumerate(s["ratios"]):
                    ko, vo = _inputs.make_finch_press(kvpress, s, r).compress(att, hidden, keys, posv, None, kwargs)
                    pos = vo[..., 0].round().to(torch.int64)
                    if s["rerotate"]:
                        assert torch.equal(pos, torch.sort(pos, dim=-1).values)   # finch_press.py:114
                        out[f"kout_{mode}_{i}"] = ko.float().numpy()
                    else:
                        order = torch.argsort(pos, dim=-1)
                        assert torch.equal(ko, keys.gather(2, pos.unsqueeze(-1).expand(-1, -1, -1, s["D"])))
                        pos = pos.gather(-1, order)
                    out[f"pos_{mode}_{i}"] = pos.numpy().astype(np.int32)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), {k: v.shape for k, v in out.items() if k.startswith("pos_")})


if __name__ == "__main__":
    main(sys.argv[1:])
