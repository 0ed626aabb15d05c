#!/usr/bin/env python3
"""Reference outputs at BASELINE.json's FULL sizes (configs 2-4): runs the REAL reference (NVIDIA/kvpress imported from
/root/reference) on the host CPU over the CPU-seeded inputs of tests/_fullsize.py and commits compact fixtures
(tests/golden/full_*.npz) for tests/test_gpu_fullsize.py.  Test infrastructure only; the GPU box never sees /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py [case ...]

Per case:
  * "O32" run (module and tensors in float32 -- SURVEY §8c): every 8th score, the torch.topk membership bitmask, the
    threshold per row and all scores within 4e-3 of it (tests/_fullsize.py: pack_reference).
    SnapKV: the float32 run consumes the window queries of the bf16 MODEL (the reference's own lines snapkv_press.py:53-58
    executed in bf16, handed to the float32 run through a q_proj hook + identity rotary tables), because that is what a
    bf16 model hands the press and what the kernels consume; everything after the query projection and RoPE
    (snapkv_press.py:61-105) is the reference's code in float32.  The all-float32 run (float32 queries, never rounded) is
    kept as `sub_pure`: it differs from any bf16-model run by the rounding of q, cos and sin (up to ~5e-3 on structured keys);
  * "Obf" run (bf16 as users run it): the bf16 scores (bit patterns) and the reference's own top-k membership (pack_native);
  * timings of both runs on this container's cores (recorded in the fixture; BASELINE.md quotes them).
"""
import os
import sys
import time

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    from gen_golden import _install_shims

    _install_shims()
    import numpy as np
    import torch
    from kvpress import CURPress, ExpectedAttentionPress, KeyDiffPress, KnormPress, SnapKVPress  # the reference

    import _fullsize as F
    import bench

    outdir = os.path.join(REPO, "tests", "golden")
    for name in argv or list(F.FULL_CASES):
        spec = F.FULL_CASES[name]
        S, ratio = spec["S"], spec["ratio"]
        n_kept = int(S * (1 - ratio))
        keys, values = F.make_kv(spec)
        hidden = F.make_hidden(spec)
        att, rot = bench.build_module(torch.device("cpu"))   # bf16 module, seeded
        press = {"knorm": lambda: KnormPress(compression_ratio=ratio),
                 "snapkv": lambda: SnapKVPress(compression_ratio=ratio, window_size=F.WINDOW, kernel_size=5),
                 "ea": lambda: ExpectedAttentionPress(compression_ratio=ratio),
                 "keydiff": lambda: KeyDiffPress(compression_ratio=ratio),
                 "cur": lambda: CURPress(compression_ratio=ratio)}[spec["kind"]]()
        out = {}
        with torch.no_grad():
            pe_bf = rot(hidden, torch.arange(S)[None])
            # ---- Obf: the reference as users run it ---------------------------------------------------------------
            t0 = time.perf_counter()
            sc_nat = press.score(att, hidden, keys, values, None, {"position_embeddings": pe_bf})
            t_nat = time.perf_counter() - t0
            assert sc_nat.dtype == torch.bfloat16 and tuple(sc_nat.shape) == (1, F.H_KV, S)
            out.update(F.pack_native(sc_nat, n_kept))
            ko, vo = press.compress(att, hidden, keys, values, None, {"position_embeddings": pe_bf})
            assert tuple(ko.shape) == (1, F.H_KV, n_kept, F.D)
            del ko, vo
            # ---- O32: same code, float32 module and tensors --------------------------------------------------------
            att32 = att.float()
            att32.rotary_emb = rot
            h32, k32, v32 = hidden.float(), keys.float(), values.float()
            pe32 = rot(h32, torch.arange(S)[None])
            t0 = time.perf_counter()
            sc32 = press.score(att32, h32, k32, v32, None, {"position_embeddings": pe32})
            t_f32 = time.perf_counter() - t0
            if spec["kind"] == "snapkv":
                from kvpress.utils import get_prerope_query_states
                from transformers.models.llama.modeling_llama import rotate_half

                W = F.WINDOW
                att.rotary_emb = rot
                q = get_prerope_query_states(att.to(torch.bfloat16), hidden[:, -W:])                      # snapkv_press.py:53 (bf16 model)
                cos, sin = pe_bf[0][:, -W:], pe_bf[1][:, -W:]
                q_rot = (q * cos.unsqueeze(1)) + (rotate_half(q) * sin.unsqueeze(1))                     # :56-58 (bf16)
                att32 = att.float()
                handle = att32.q_proj.register_forward_hook(lambda m, i, o: q_rot.float().transpose(1, 2).reshape(1, W, F.H_Q * F.D))
                pe_id = (torch.ones((1, S, F.D)), torch.zeros((1, S, F.D)))
                out["sub_pure"] = sc32[0, :, F.SUB_OFFSET::F.SUBSAMPLE].numpy().astype(np.float32)
                sc32 = press.score(att32, h32, k32, v32, None, {"position_embeddings": pe_id})          # :61-105 in float32
                handle.remove()
        pad = {"knorm": (0, 0), "snapkv": (S - F.WINDOW, S), "ea": (0, 4), "keydiff": (0, 0), "cur": (0, 4)}[spec["kind"]]   # (CUR: its 4 sinks = 1.0)
        out.update(F.pack_reference(sc32, n_kept, *pad))
        out["ref_seconds"] = np.asarray([t_nat, t_f32])
        out["ref_threads"] = np.int64(torch.get_num_threads())
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        # calibration printout for the dtype-faithful check: how far from the bf16 threshold do O32-kept / O32-dropped positions sit?
        kept32 = torch.from_numpy(np.unpackbits(out["kept_bits"], axis=-1)[:, :S].astype(bool))
        nat = sc_nat[0].float()
        kept_nat = torch.from_numpy(np.unpackbits(out["nat_kept_bits"], axis=-1)[:, :S].astype(bool))
        t = nat.masked_fill(~kept_nat, float("inf")).amin(-1, keepdim=True)
        ulp = 2.0 ** -8 * t.abs()
        worst_drop = float((((nat - t) / ulp).masked_fill(kept32, 0)).amax())     # dropped by O32 although this many ulps above
        worst_keep = float((((t - nat) / ulp).masked_fill(~kept32, 0)).amax())    # kept by O32 although this many ulps below
        overlap = float((kept32 & kept_nat).sum()) / (F.H_KV * n_kept)
        print(f"{name}: {os.path.getsize(path)} bytes; reference {t_nat:.1f} s (bf16) / {t_f32:.1f} s (fp32) on {torch.get_num_threads()} threads; "
              f"band entries {len(out['band_pos'])}; O32-vs-Obf overlap {overlap:.4f}, worst ulps above/below threshold {worst_drop:.1f}/{worst_keep:.1f}", flush=True)


def gen_batch_case(name):
    """BATCH_CASES (tests/_fullsize.py): the reference's float32 run over a batch of B different elements in ONE call -- its pad
    constant is the maximum over the whole batch (snapkv_press.py:103).  As in the single-element SnapKV cases the float32 run
    consumes the bf16 model's own window queries (snapkv_press.py:53-58 in bf16, through a q_proj hook + identity rotary tables)."""
    from gen_golden import _install_shims

    _install_shims()
    import numpy as np
    import torch
    from kvpress import SnapKVPress  # the reference
    from kvpress.utils import get_prerope_query_states
    from transformers.models.llama.modeling_llama import rotate_half

    import _fullsize as F
    import bench

    spec = F.BATCH_CASES[name]
    assert spec["kind"] == "snapkv"
    S, ratio, B, W = spec["S"], spec["ratio"], len(spec["elements"]), F.WINDOW
    n_kept = int(S * (1 - ratio))
    kv = [F.make_kv(F.element_spec(spec, b)) for b in range(B)]
    keys, values = torch.cat([k for k, _ in kv]), torch.cat([v for _, v in kv])
    hidden = torch.cat([F.make_hidden(F.element_spec(spec, b)) for b in range(B)])
    att, rot = bench.build_module(torch.device("cpu"))
    att.rotary_emb = rot
    press = SnapKVPress(compression_ratio=ratio, window_size=W, kernel_size=5)
    with torch.no_grad():
        pe_bf = rot(hidden[:1], torch.arange(S)[None])
        q = get_prerope_query_states(att, hidden[:, -W:])                                              # snapkv_press.py:53 (bf16 model)
        cos, sin = pe_bf[0][:, -W:], pe_bf[1][:, -W:]
        q_rot = (q * cos.unsqueeze(1)) + (rotate_half(q) * sin.unsqueeze(1))                           # :56-58 (bf16)
        att32 = att.float()
        handle = att32.q_proj.register_forward_hook(lambda m, i, o: q_rot.float().transpose(1, 2).reshape(B, W, F.H_Q * F.D))
        pe_id = (torch.ones((1, S, F.D)), torch.zeros((1, S, F.D)))
        t0 = time.perf_counter()
        sc32 = press.score(att32, hidden.float(), keys.float(), values.float(), None, {"position_embeddings": pe_id})   # :61-105 in float32, B = 2
        dt = time.perf_counter() - t0
        handle.remove()
    assert tuple(sc32.shape) == (B, F.H_KV, S)
    pad_value = float(sc32[..., -1].max())
    assert bool((sc32[..., S - W:] == pad_value).all()), "one pad constant for the whole batch"
    out = {"B": np.int64(B), "pad_value": np.float32(pad_value), "elem_max": sc32[..., :S - W].amax(dim=(1, 2)).numpy().astype(np.float32)}
    for b in range(B):
        for k, v in F.pack_reference(sc32[b:b + 1], n_kept, S - W, S, subsample=F.BATCH_SUBSAMPLE).items():
            out[f"{k}__b{b}"] = v
    path = os.path.join(REPO, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)} bytes; reference {dt:.1f} s (float32, B = {B}); pad constant {pad_value:.6g}, per-element maxima {out['elem_max']}", flush=True)


if __name__ == "__main__":
    import _fullsize as _F

    args = sys.argv[1:]
    single = [a for a in args if a in _F.FULL_CASES] if args else list(_F.FULL_CASES)
    batch = [a for a in args if a in _F.BATCH_CASES] if args else list(_F.BATCH_CASES)
    if single:
        main(single)
    for name in batch:
        gen_batch_case(name)
