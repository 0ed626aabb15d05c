#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (NVIDIA/kvpress v0.5.4 imported
from /root/reference) on the seeded inputs of tests/_inputs.py.

Test infrastructure only (see oracle/kvpress_oracle.py header).  Runs in the build container
(where /root/reference is mounted); the GPU box only sees the committed .npz files.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]

Per case the fixture holds the reference's outputs only (inputs are regenerated from the seed):
  scores_f32   reference ``press.score()`` with module and tensors in float32 ("O32")
  scores_nat   reference ``press.score()`` in the case's native dtype (bf16/f16), stored as float32
  ratios       compression ratios tried
  nkept_<i>    ``compress()`` output length for ratio i  (pins int(S*(1-r)))
  idx_f32_<i>  torch.topk indices of the O32 scores for ratio i, sorted ascending, int32 [B,H,n]
  qwin_f32     (snapkv) RoPE'd window queries [B,Hq,W,D] of the O32 run
  mu_f32       (ea) post-RoPE query mean [B,Hq,D] of the O32 run
  cov_f32      (ea, small D only) post-RoPE covariance [B,Hq,D,D]
Shims: ``cachetools`` and ``fire`` are not installed -> stub modules; the hook is not
used (direct ``score()`` / ``compress()`` calls), so the transformers-5.x ``cache_position`` drift
does not matter here.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))


def _install_shims():
    ct = types.ModuleType("cachetools")

    class LRUCache(dict):
        def __init__(self, maxsize=128):
            super().__init__()

    def cached(cache=None, **kw):
        return lambda f: f

    ct.LRUCache, ct.cached = LRUCache, cached
    sys.modules.setdefault("cachetools", ct)
    sys.modules.setdefault("fire", types.ModuleType("fire"))
    sys.path.insert(0, "/root/reference")


def main(argv):
    _install_shims()
    import numpy as np
    import torch
    from kvpress import (CURPress, ExpectedAttentionPress, KeyDiffPress, KnormPress, LagKVPress, ObservedAttentionPress, PyramidKVPress,  # the reference
                         QFilterPress,
                         SnapKVPress, StreamingLLMPress, TOVAPress)

    import _inputs

    torch.manual_seed(0)
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    names = argv or list(_inputs.CASES)
    for name in names:
        s = _inputs.make_case(name)

        def make_press(ratio):
            if s["kind"] == "knorm":
                return KnormPress(compression_ratio=ratio)
            if s["kind"] == "snapkv":
                return SnapKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"])
            if s["kind"] == "keydiff":
                return KeyDiffPress(compression_ratio=ratio)
            if s["kind"] == "lagkv":
                return LagKVPress(compression_ratio=ratio, n_sink=s["n_sink"], lag_size=s["lag"], cross_scoring=s.get("cross", False))
            if s["kind"] == "observed":
                return ObservedAttentionPress(compression_ratio=ratio)
            if s["kind"] == "qfilter":   # the published filters need the hub: seeded stand-ins, assigned directly
                p = QFilterPress(compression_ratio=ratio)
                p.q_filters = torch.from_numpy(_inputs.make_qfilters(s)).to(cur_dtype[0])
                return p
            if s["kind"] == "cur":
                return CURPress(compression_ratio=ratio, num_sinks=s.get("sinks", 4), leverage_type=s["leverage"],
                                use_local_approximation=s.get("local", True), local_window_size=s.get("window", 16))
            if s["kind"] == "tova":
                return TOVAPress(compression_ratio=ratio)
            if s["kind"] == "pyramid":
                return PyramidKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"], beta=s["beta"])
            if s["kind"] == "streaming":
                return StreamingLLMPress(compression_ratio=ratio, n_sink=s["n_sink"])
            return ExpectedAttentionPress(
                compression_ratio=ratio, n_future_positions=s["n_future"], n_sink=s["n_sink"],
                use_covariance=s["use_covariance"], use_vnorm=s["use_vnorm"], epsilon=s["epsilon"])

        out = {"ratios": np.asarray(s["ratios"], dtype=np.float64)}
        captured = {}
        cur_dtype = [torch.float32]
        for mode, dt in (("f32", torch.float32), ("nat", _inputs.torch_dtype(s["dtype"]))):
            att, rot, hidden, pe = _inputs.build_llama_attention(s, dt)
            cur_dtype[0] = dt
            if s["kind"] == "qfilter":
                att.layer_idx = _inputs.QF_LAYER
            if s["kind"] == "pyramid":  # the budget reads the layer's position in the stack (pyramidkv_press.py:80-81)
                att.config.num_hidden_layers = s["n_layers"]
                att.layer_idx = s["layer_idx"]
            keys = torch.from_numpy(s["keys"]).to(dt)
            values = torch.from_numpy(s["values"]).to(dt)
            kwargs = {"position_embeddings": pe}
            attn = torch.from_numpy(_inputs.make_attentions(s)).to(dt) if s["kind"] == "observed" else None
            with torch.no_grad():
                press = make_press(0.5)
                if mode == "f32" and s["kind"] == "ea":
                    mu, cov = press.get_query_statistics(att, hidden)
                    captured["mu_f32"] = mu.numpy().astype(np.float32)
                    if s["D"] <= 64:
                        captured["cov_f32"] = cov.numpy().astype(np.float32) if cov is not None else np.zeros(0, np.float32)
                if mode == "f32" and s["kind"] in ("snapkv", "tova", "pyramid"):
                    from kvpress.utils import get_prerope_query_states
                    from transformers.models.llama.modeling_llama import rotate_half

                    q = get_prerope_query_states(att, hidden[:, -s["W"]:])
                    c, si = pe[0][:, -s["W"]:], pe[1][:, -s["W"]:]
                    captured["qwin_f32"] = ((q * c.unsqueeze(1)) + (rotate_half(q) * si.unsqueeze(1))).numpy()
                sc = press.score(att, hidden, keys, values, attn, kwargs)
                out[f"scores_{mode}"] = sc.float().numpy()
                if mode == "f32":
                    for i, r in enumerate(s["ratios"]):
                        p = make_press(r)
                        ko, vo = p.compress(att, hidden, keys, values, attn, kwargs)
                        assert ko.shape == vo.shape and ko.is_contiguous()
                        out[f"nkept_{i}"] = np.int64(ko.shape[2])
                        n = ko.shape[2]  # int(S * (1 - r)) except for per-layer budgets (PyramidKV)
                        sc_r = p.score(att, hidden, keys, values, None, kwargs) if s["kind"] == "streaming" else sc
                        idx = sc_r.topk(n, dim=-1).indices.sort(dim=-1).values
                        out[f"idx_f32_{i}"] = idx.numpy().astype(np.int32)
        out.update(captured)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {os.path.getsize(path)} bytes; nkept={[int(out[f'nkept_{i}']) for i in range(len(s['ratios']))]}")


if __name__ == "__main__":
    main(sys.argv[1:])
