#!/usr/bin/env python3
"""Golden outputs of the reference's selection wrappers (ChunkPress, KeyRerotationPress) around reference scorers ->
tests/golden/<wrap case>.npz.  Test infrastructure only; needs /root/reference (see gen_golden.py for the shims).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_wrappers.py

The kept positions are recovered by compressing a VALUE tensor that stores each token's position (the reference gathers
K and V with the same indices).  Per case and ratio i:  pos_<i> int32 [B,H,n] (sorted per chunk resp. overall, as a set
the order inside a chunk is torch.topk's), and for the rerotation wrapper  kout_f32_<i> / kout_nat_<i>: the re-rotated
keys of the float32 run and of the native-dtype run (float32 storage).
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
    from kvpress import AdaKVPress, BlockPress, ChunkKVPress, ChunkPress, CriticalAdaKVPress, CriticalKVPress, KeyDiffPress, KeyRerotationPress, KnormPress, SnapKVPress, StreamingLLMPress

    import _inputs

    outdir = os.path.join(REPO, "tests", "golden")
    for name in (argv or list(_inputs.WRAP_CASES)):
        s = _inputs.make_wrap_case(name)

        def inner(ratio):
            if s["kind"] == "knorm":
                return KnormPress(compression_ratio=ratio)
            if s["kind"] == "snapkv":
                return SnapKVPress(compression_ratio=ratio, window_size=s["W"], kernel_size=s["ks"])
            if s["kind"] == "keydiff":
                return KeyDiffPress(compression_ratio=ratio)
            return StreamingLLMPress(compression_ratio=ratio, n_sink=s["n_sink"])

        def wrap(ratio):
            if s["wrapper"] == "adakv":
                return AdaKVPress(inner(ratio), alpha_safeguard=s["alpha"])
            if s["wrapper"] == "block":
                return BlockPress(inner(ratio), block_size=s["block_size"])
            if s["wrapper"] == "critical":
                return CriticalKVPress(inner(ratio))
            if s["wrapper"] == "criticalada":
                return CriticalAdaKVPress(inner(ratio), alpha_safeguard=s["alpha"])
            if s["wrapper"] == "chunkkv":
                return ChunkKVPress(inner(ratio), chunk_length=s["chunk_length"])
            return ChunkPress(inner(ratio), chunk_length=s["chunk_length"]) if s["wrapper"] == "chunk" else KeyRerotationPress(inner(ratio))

        out = {"ratios": np.asarray(s["ratios"], dtype=np.float64)}
        for mode, dt in (("f32", torch.float32), ("nat", _inputs.torch_dtype(s["dtype"]))):
            att, rot, hidden, pe = _inputs.build_llama_attention(s, dt)
            att.rotary_emb = rot
            keys = torch.from_numpy(s["keys"]).to(dt)
            posv = torch.arange(s["S"], dtype=torch.float32)[None, None, :, None].expand(s["B"], s["H"], s["S"], s["D"]).contiguous()
            kwargs = {"position_embeddings": pe}
            if s["wrapper"] == "critical":
                # a ScorerPress that reads the VALUES (||Wo v||_1): real values; the float32 run's scores and the kept sets
                if mode == "f32":
                    values = torch.from_numpy(s["values"]).to(dt)
                    with torch.no_grad():
                        for i, r in enumerate(s["ratios"]):
                            p = wrap(r)
                            sc = p.score(att, hidden, keys, values, None, kwargs)
                            out[f"scores_{i}"] = sc.numpy()
                            ko, vo = p.compress(att, hidden, keys, values, None, kwargs)
                            n = ko.shape[2]
                            assert n == int(s["S"] * (1 - r))
                            out[f"pos_{i}"] = sc.topk(n, dim=-1).indices.sort(dim=-1).values.numpy().astype(np.int32)
                continue
            if s["wrapper"] in ("adakv", "criticalada"):
                # K/V stay untouched; the pruned (batch, head, position) triples land in module.masked_key_indices
                # (adakv_press.py:70-75).  Stored: sorted flat indices head * S + position per batch element (float32 run).
                if mode == "f32":
                    att.config._attn_implementation = "sdpa"
                    with torch.no_grad():
                        for i, r in enumerate(s["ratios"]):
                            vals = torch.from_numpy(s["values"]).to(dt) if s["wrapper"] == "criticalada" else posv
                            ko, vo = wrap(r).compress(att, hidden, keys, vals, None, kwargs)
                            assert ko is keys
                            bi, hi, si = att.masked_key_indices
                            flat = (hi * s["S"] + si).reshape(s["B"], -1)
                            out[f"masked_{i}"] = torch.sort(flat, dim=-1).values.numpy().astype(np.int64)
                continue
            with torch.no_grad():
                for i, r in enumerate(s["ratios"]):
                    ko, vo = wrap(r).compress(att, hidden, keys, posv, None, kwargs)
                    pos = vo[..., 0].round().to(torch.int64)
                    if mode == "f32" and s["wrapper"] == "block":
                        out[f"pos_{i}"] = pos.numpy().astype(np.int32)   # in the reference's own (descending-score) order
                    elif mode == "f32":
                        if s["wrapper"] == "chunk":   # order inside a chunk is torch.topk's: store every chunk sorted
                            L = s["chunk_length"]
                            pos = torch.sort(pos, dim=-1).values  # chunks are disjoint position ranges: a global sort = per-chunk sort
                        out[f"pos_{i}"] = pos.numpy().astype(np.int32)
                    if s["wrapper"] == "rerot":
                        out[f"kout_{mode}_{i}"] = ko.float().numpy()
                        if mode == "nat":
                            out[f"pos_nat_{i}"] = pos.numpy().astype(np.int32)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), {k: v.shape for k, v in out.items() if k.startswith(("pos_", "masked_"))})


if __name__ == "__main__":
    main(sys.argv[1:])
