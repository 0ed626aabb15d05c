"""Seeded inputs at BASELINE.json's FULL sizes (configs 2-4), shared by oracle/gen_golden_fullsize.py (runs the real
reference on the host CPU of the build container) and tests/test_gpu_fullsize.py (GPU box, no reference there).

Everything random comes from a CPU ``torch.Generator`` ("CPU generation so the GPU box reproduces bits"),
is rounded to bf16 once, and is moved to the device afterwards; the attention module is ``bench.build_module`` (random
init under ``torch.manual_seed(0)`` on the CPU, Llama-3.1-8B geometry, llama3 RoPE scaling).
"""
from __future__ import annotations

import numpy as np
import torch

H_Q, H_KV, D, HIDDEN, WINDOW = 32, 8, 128, 4096, 64

# name -> spec.  data "A": flat N(0,1); "B": per-channel key scales, 4 heavy sink rows, log-normal value norms
FULL_CASES = {
    "full_knorm32k": dict(kind="knorm", S=32768, ratio=0.5, data="A", seed=102),                   # BASELINE config 2
    "full_snapkv128k": dict(kind="snapkv", S=131072, ratio=0.5, data="A", seed=103),               # BASELINE config 3 (the bench workload)
    "full_snapkv128k_B": dict(kind="snapkv", S=131072 - 1000 + 37, ratio=0.5, data="B", seed=113),  # ragged length, structured keys
    "full_ea128k": dict(kind="ea", S=131072, ratio=0.7, data="B", seed=104),                       # BASELINE config 4
    "full_ea128k_A": dict(kind="ea", S=131072, ratio=0.7, data="A", seed=104),                     # config 4 on bench.py's timed tensors (set A)
    # SURVEY §8(f-2) scorers at the BASELINE size (round 3: the kernels they run on were reshaped for this size)
    "full_keydiff128k": dict(kind="keydiff", S=131072, ratio=0.5, data="B", seed=105),
    "full_cur128k": dict(kind="cur", S=131072, ratio=0.5, data="B", seed=106),
}
# BASELINE config 5's shard shape: ONE reference run over a batch of two different elements (flat data, then structured data with
# ~100x larger score maxima), i.e. with the reference's pad constant `scores.max().item() + 1` taken over BOTH (snapkv_press.py:103).
# The kernels are then run per element (one shard per GPU: each shard only sees its own maximum) and on the whole batch: the
# retained sets must be the reference's either way.  Stored per element with pack_reference (float32 "O32" run only).
BATCH_CASES = {
    "full_snapkv128k_B2": dict(kind="snapkv", S=131072, ratio=0.5, elements=[dict(data="A", seed=103), dict(data="B", seed=213)]),
}
BATCH_SUBSAMPLE = 32   # (the dense per-element comparison is full_snapkv128k's job: these fixtures keep every 32nd score + the band)
SUBSAMPLE = 8      # the fixtures keep every 8th score (offset 5) + everything near the selection threshold
SUB_OFFSET = 5
BAND = 4e-3        # relative half-width of the stored threshold band (4x the 1e-3 score tolerance)


def make_kv(spec: dict):
    """K, V [1, H_kv, S, D] bf16 on the CPU."""
    g = torch.Generator().manual_seed(spec["seed"])
    S = spec["S"]
    k = torch.randn((1, H_KV, S, D), generator=g)
    v = torch.randn((1, H_KV, S, D), generator=g)
    if spec["data"] == "B":
        k = k * torch.exp(0.5 * torch.randn((1, H_KV, 1, D), generator=g))
        k[:, :, :4] *= 8.0
        v = v * torch.exp(0.7 * torch.randn((1, H_KV, S, 1), generator=g))
    return k.to(torch.bfloat16), v.to(torch.bfloat16)


def make_hidden(spec: dict):
    """hidden states [1, S, 4096] bf16 on the CPU.  SnapKV reads only the last 64 rows (snapkv_press.py:53), so only those
    are random there (the rest is zero); ExpectedAttention reads them all (expected_attention_press.py:70-71)."""
    g = torch.Generator().manual_seed(spec["seed"] + 1000)
    S = spec["S"]
    if spec["kind"] == "ea":
        h = torch.randn((1, S, HIDDEN), generator=g)
        if spec["data"] == "B":
            h += 0.25   # a non-zero query mean
        return h.to(torch.bfloat16)
    h = torch.zeros((1, S, HIDDEN), dtype=torch.bfloat16)
    if spec["kind"] == "snapkv":
        h[:, -WINDOW:] = torch.randn((1, WINDOW, HIDDEN), generator=g).to(torch.bfloat16)
    return h


def element_spec(spec: dict, b: int) -> dict:
    """the single-element spec of element b of a BATCH_CASES entry (for make_kv / make_hidden)"""
    return dict(kind=spec["kind"], S=spec["S"], ratio=spec["ratio"], **spec["elements"][b])


def pack_reference(scores: torch.Tensor, n_kept: int, pad_lo: int, pad_hi: int, subsample: int = SUBSAMPLE) -> dict:
    """Fixture arrays from reference float32 scores [1, H, S]: every SUBSAMPLE-th score, the top-k membership bitmask, the
    threshold per row and every score within BAND of it.  Columns [pad_lo, pad_hi) hold the reference's pad constant
    (max + 1: kept by construction) and are excluded from the numeric comparison."""
    sc = scores[0].float()
    H, S = sc.shape
    idx = sc.topk(n_kept, dim=-1).indices
    kept = torch.zeros((H, S), dtype=torch.bool)
    kept.scatter_(1, idx, True)
    t = sc.gather(1, idx).amin(-1)
    # EVERY score of the row, in half precision relative to a per-row power of two (the subsample pins 1 column in 8
    # to float32; this pins all of them to 2^-11): all16 = fp16(score * 2^e) with 2^e = 2^14 / (largest finite non-pad |score|)
    numeric_cols = torch.ones(S, dtype=torch.bool)
    numeric_cols[pad_lo:pad_hi] = False
    amax = sc[:, numeric_cols].abs().amax(-1).clamp_min(1e-30)
    e16 = torch.floor(14.0 - torch.log2(amax)).clamp(-100, 100)
    all16 = (sc * torch.exp2(e16)[:, None]).clamp(-65504, 65504).to(torch.float16)
    out = {
        "all16": all16.view(torch.int16).numpy().view(np.uint16), "all16_exp": e16.numpy().astype(np.float32),
        "sub": sc[:, SUB_OFFSET::subsample].numpy().astype(np.float32), "subsample": np.int64(subsample),
        "kept_bits": np.packbits(kept.numpy(), axis=-1),
        "threshold": t.numpy().astype(np.float32),
        "n_kept": np.int64(n_kept), "pad": np.asarray([pad_lo, pad_hi], dtype=np.int64),
    }
    pos, val, off = [], [], [0]
    for h in range(H):
        near = ((sc[h] - t[h]).abs() <= BAND * t[h].abs()).nonzero().flatten()
        pos.append(near.numpy().astype(np.int32))
        val.append(sc[h, near].numpy().astype(np.float32))
        off.append(off[-1] + near.numel())
    out["band_pos"], out["band_val"], out["band_off"] = np.concatenate(pos), np.concatenate(val), np.asarray(off, dtype=np.int64)
    return out


def check_against_reference(fx, scores: torch.Tensor, idx: torch.Tensor, rtol: float = 1e-3, atol: float = 0.0, dense_atol=None):
    """The kernel's float32 scores [1,H,S] and kept indices [1,H,n] against a fixture of pack_reference():
    (i) every stored reference score (subsample + threshold band) within rtol; (ii) tie-tolerant set parity:
    everything the reference keeps with a margin > rtol above its threshold is kept, nothing it drops with such a margin is
    kept, and the rest of the set differs only inside the band.  `atol`: absolute slack for scorers whose values cross zero (KeyDiff's
    cosines: a relative error is meaningless at |score| ~ 1e-6).  Returns (max rel err beyond atol, #positions where the sets differ)."""
    sc = scores[0].float().cpu()
    H, S = sc.shape
    n = int(fx["n_kept"])
    pad_lo, pad_hi = (int(x) for x in fx["pad"])
    assert idx.shape[-1] == n
    sub = torch.from_numpy(fx["sub"])
    cols = torch.arange(SUB_OFFSET, S, int(fx["subsample"]) if "subsample" in fx else SUBSAMPLE)
    numeric = (cols < pad_lo) | (cols >= pad_hi)
    got = sc[:, cols]
    rel = (((got - sub).abs() - atol).clamp_min(0) / sub.abs().clamp_min(1e-30))[:, numeric]
    worst = float(rel.max())
    assert worst <= rtol, f"scores differ from the reference by {worst:.3e} (subsample)"
    if "all16" in fx:   # every column: the reference's score in scaled half precision (2^-11 relative; subnormal results below 2^-24 of the row's maximum)
        ref16 = torch.from_numpy(fx["all16"].view(np.int16).copy()).view(torch.float16).double() * torch.exp2(-torch.from_numpy(fx["all16_exp"]).double())[:, None]
        allc = torch.arange(S)
        num_all = (allc < pad_lo) | (allc >= pad_hi)
        floor16 = torch.exp2(-torch.from_numpy(fx["all16_exp"]).double() - 24.0)[:, None]   # half an fp16 subnormal step, in score units
        # `dense_atol`: the absolute slack over ALL columns for scores that cross zero (KeyDiff's cosines: a 128-term dot product with
        # cancellation is good to ~1e-5 absolute in float32 whatever the result's size, in the reference's pipeline as in this one;
        # the maximum over a million columns is larger than over the subsample's eighth)
        slack = atol if dense_atol is None else dense_atol
        err = ((sc.double() - ref16).abs() - slack - floor16).clamp_min(0) / ref16.abs().clamp_min(1e-300)
        err = err[:, num_all]
        bad = int((err > rtol + 2.0 ** -11).sum())
        assert bad == 0, f"{bad} scores differ from the reference's (all columns, half precision) by more than {rtol + 2.0 ** -11:.2e}: worst {float(err.max()):.3e}"
    if "sub_pure" in fx:   # all-float32 reference run (queries never rounded to bf16): differs by the model's bf16 q / cos / sin
        pure = torch.from_numpy(fx["sub_pure"])
        rel_pure = ((got - pure).abs() / pure.abs().clamp_min(1e-30))[:, numeric]
        # measured when the fixtures were made: 3.0e-4 (flat data) / 4.5e-3 (structured keys) -- the bound leaves a third of headroom
        assert rel_pure.max() <= 6e-3, f"scores differ from the all-float32 reference by {float(rel_pure.max()):.3e}"
    kept_ref = torch.from_numpy(np.unpackbits(fx["kept_bits"], axis=-1)[:, :S].astype(bool))
    kept = torch.zeros((H, S), dtype=torch.bool)
    kept.scatter_(1, idx[0].long().cpu(), True)
    assert int(kept.sum()) == H * n
    differ = 0
    off = fx["band_off"]
    for h in range(H):
        p = torch.from_numpy(fx["band_pos"][off[h]:off[h + 1]]).long()
        v = torch.from_numpy(fx["band_val"][off[h]:off[h + 1]])
        r = (((sc[h, p] - v).abs() - atol).clamp_min(0) / v.abs().clamp_min(1e-30))
        m = (p < pad_lo) | (p >= pad_hi)
        if m.any():
            worst = max(worst, float(r[m].max()))
            assert r[m].max() <= rtol, f"row {h}: scores near the threshold differ by {float(r[m].max()):.3e}"
        t = float(fx["threshold"][h])
        d = (kept[h] != kept_ref[h]).nonzero().flatten()
        differ += d.numel()
        if d.numel():
            # any disagreement must sit inside the tolerance band around the reference threshold
            band = set(p[((v - t).abs() <= 2 * (rtol * abs(t) + atol))].tolist())   # |s_k - s_r| <= rtol moves a score AND the threshold
            bad = [int(x) for x in d.tolist() if int(x) not in band]
            assert not bad, f"row {h}: {len(bad)} kept/dropped positions disagree with the reference outside the {rtol:g} band, e.g. {bad[:5]}"
    return worst, differ


def pack_native(scores_nat: torch.Tensor, n_kept: int) -> dict:
    """The reference AS USERS RUN IT (bf16 module and tensors, "Obf" of SURVEY §8c): its bf16 scores (bit patterns) and its own
    torch.topk membership."""
    sc = scores_nat[0]
    assert sc.dtype == torch.bfloat16
    idx = sc.float().topk(n_kept, dim=-1).indices
    kept = torch.zeros(sc.shape, dtype=torch.bool)
    kept.scatter_(1, idx, True)
    return {"nat_bits": sc.view(torch.int16).numpy().view(np.uint16), "nat_kept_bits": np.packbits(kept.numpy(), axis=-1)}


def check_against_native(fx, idx: torch.Tensor, S: int, ulps: int, overlap_floor: float):
    """Dtype-faithful check (SURVEY §8c iii): the kernel's kept set against the reference's OWN bf16 scores with a band of
    `ulps` bf16 units in the last place around the reference's bf16 threshold (the reference rounds to bf16 after the matmul,
    the scaling, the softmax, each mean and the pooling, so its score of a position is only defined to a few ulps):
    every position whose bf16 score lies more than the band ABOVE the threshold is kept, none more than the band BELOW is;
    plus a floor on the overlap with the reference's own torch.topk choice.  Returns (overlap, #violations)."""
    nat = torch.from_numpy(fx["nat_bits"].view(np.int16).copy()).view(torch.bfloat16).float()
    H = nat.shape[0]
    n = int(fx["n_kept"])
    kept = torch.zeros((H, S), dtype=torch.bool)
    kept.scatter_(1, idx[0].long().cpu(), True)
    kept_ref = torch.from_numpy(np.unpackbits(fx["nat_kept_bits"], axis=-1)[:, :S].astype(bool))
    t = nat.masked_fill(~kept_ref, float("inf")).amin(-1, keepdim=True)     # the reference's bf16 threshold per row
    band = ulps * 2.0 ** -8 * t.abs()                                        # one bf16 ulp <= 2^-8 relative
    must_keep = nat > t + band
    must_drop = nat < t - band
    bad = int((must_keep & ~kept).sum() + (must_drop & kept).sum())
    overlap = float((kept & kept_ref).sum()) / (H * n)
    assert bad == 0, f"{bad} positions outside the {ulps}-ulp band are selected differently from the bf16 reference"
    assert overlap >= overlap_floor, f"overlap with the bf16 reference's top-k {overlap:.4f} < {overlap_floor}"
    return overlap, bad
