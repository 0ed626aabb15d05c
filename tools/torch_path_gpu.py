#!/usr/bin/env python3
"""What the reference's algorithm costs on the SAME MI355X when it is expressed in plain PyTorch-ROCm ops (BASELINE.md §3,
"third column"), next to this package's press on the same tensors.

The reference itself cannot travel to the GPU box, so the op sequence of its three scorers and of ``ScorerPress.compress``
is restated here with torch calls, one line per reference line (cited) -- same intermediates, same dtypes, same rounding
points, nothing fused.  Measurement aid only: nothing in the package, the tests or bench.py imports this file.

    python tools/torch_path_gpu.py [--workload snapkv128k|knorm32k|knorm128k|ea128k] [--reps 10]

Prints one JSON line per workload: ms/layer of the torch path and of the HIP path (CUDA events on the current stream,
3 warm-up calls), the peak extra memory of each, and the overlap of the two retained sets.
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.nn import functional as F  # noqa: E402

import bench  # noqa: E402  (module geometry + build_module)


def repeat_kv(x, n_rep):
    B, H, S, D = x.shape
    return x if n_rep == 1 else x[:, :, None].expand(B, H, n_rep, S, D).reshape(B, H * n_rep, S, D)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def knorm_score(module, hidden, keys, values, kwargs):
    return -keys.norm(dim=-1)                                                    # knorm_press.py:38


def snapkv_score(module, hidden, keys, values, kwargs, W=64, ks=5):
    B, H, S, D = keys.shape
    Hq = module.config.num_attention_heads
    q = module.q_proj(hidden[:, -W:]).view(B, W, Hq, D).transpose(1, 2)          # utils.py:43-46
    cos, sin = kwargs["position_embeddings"]
    cos, sin = cos[:, -W:], sin[:, -W:]
    q = (q * cos.unsqueeze(1)) + (rotate_half(q) * sin.unsqueeze(1))             # snapkv_press.py:56-58
    k = repeat_kv(keys, Hq // H)                                                 # :61
    attn = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D)                     # :62
    mask = torch.ones_like(attn) * float("-inf")                                 # :63
    attn = attn + torch.triu(mask, diagonal=S - W + 1)                           # :64-65
    attn = F.softmax(attn, dim=-1, dtype=torch.float32).to(q.dtype)              # :66
    attn = attn[..., :-W]                                                        # :67
    sc = attn.mean(dim=-2)                                                       # :95
    sc = F.avg_pool1d(sc, kernel_size=ks, padding=ks // 2, stride=1)             # :96
    sc = sc.view(B, H, Hq // H, S - W).mean(2)                                   # :99-100
    return F.pad(sc, (0, W), value=sc.max().item() + 1)                          # :103


def ea_score(module, hidden, keys, values, kwargs, n_future=512, n_sink=4):
    B, H, S, D = keys.shape
    Hq = module.config.num_attention_heads
    G = Hq // H
    keys, values = keys[:, :, n_sink:], values[:, :, n_sink:]                    # expected_attention_press.py:137-139
    h = hidden[:, n_sink:]                                                       # :70
    q = module.q_proj(h).view(B, S - n_sink, Hq, D).transpose(1, 2)              # :71
    mu = q.mean(dim=2, keepdim=True)                                             # :74
    qc = q - mu
    cov = torch.einsum("bnsi,bnsj->bnij", qc, qc) / h.shape[1]                   # :79-80
    mu = mu.squeeze(2)
    pos = torch.arange(S, S + n_future, device=keys.device).unsqueeze(0)         # :110
    cos, sin = module.rotary_emb(mu, pos)                                        # :112
    cos, sin = cos[0], sin[0]
    Id = torch.eye(D, device=cos.device, dtype=cos.dtype)                        # :114-119
    P = torch.zeros((D, D), device=cos.device, dtype=cos.dtype)
    P[D // 2:, : D // 2], P[: D // 2, D // 2:] = torch.eye(D // 2), -torch.eye(D // 2)
    R = (cos.unsqueeze(1) * Id + sin.unsqueeze(1) * P).mean(dim=0).to(mu.device)  # :120
    mu = torch.matmul(mu, R.T)                                                   # :121
    cov = torch.matmul(R, torch.matmul(cov, R.T))                                # :123
    kt = repeat_kv(keys, G).transpose(2, 3)                                      # :148
    sc = torch.matmul(mu.unsqueeze(2), kt).squeeze(2) / math.sqrt(D)             # :149
    sc = sc + torch.einsum("bhin,bhij,bhjn->bhn", kt, cov, kt) / D / 2           # :151
    sc = F.softmax(sc, dim=-1)                                                   # :152
    sc = sc.view(B, H, G, S - n_sink).mean(dim=2)                                # :155-156
    sc = sc * values.norm(dim=-1)                                                # :159-160 (epsilon = 0)
    return F.pad(sc, (n_sink, 0), value=sc.max().item() + 1)                     # :163


def torch_compress(score_fn, ratio, module, hidden, keys, values, kwargs):
    sc = score_fn(module, hidden, keys, values, kwargs)                          # scorer_press.py:90
    n_kept = int(keys.shape[2] * (1 - ratio))                                    # :93-94
    idx = sc.topk(n_kept, dim=-1).indices                                        # :95
    idx = idx.unsqueeze(-1).expand(-1, -1, -1, module.head_dim)                  # :96
    return keys.gather(2, idx).contiguous(), values.gather(2, idx).contiguous(), idx[..., 0]  # :99-100


WORKLOADS = {"knorm32k": ("knorm", 32768, 0.5), "knorm128k": ("knorm", 131072, 0.5), "snapkv128k": ("snapkv", 131072, 0.5),
             "ea128k": ("ea", 131072, 0.7)}
SCORERS = {"knorm": knorm_score, "snapkv": snapkv_score, "ea": ea_score}


def timed(fn, reps):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (torch.cuda.max_memory_allocated() - base) / 2**20, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=["all"] + list(WORKLOADS))
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    att, rot = bench.build_module(dev)
    for wl in (list(WORKLOADS) if args.workload == "all" else [args.workload]):
        kind, S, ratio = WORKLOADS[wl]
        g = torch.Generator().manual_seed(0)
        keys = torch.randn((1, bench.H_KV, S, bench.D), generator=g).to(dev, torch.bfloat16)
        values = torch.randn((1, bench.H_KV, S, bench.D), generator=g).to(dev, torch.bfloat16)
        hidden = torch.randn((1, S, bench.HIDDEN), generator=g).to(dev, torch.bfloat16)
        with torch.no_grad():
            pe = rot(hidden, torch.arange(S, device=dev)[None])
            kwargs = {"position_embeddings": pe}
            press = bench.make_press(kind, ratio)
            t_torch, m_torch, (k1, v1, i1) = timed(lambda: torch_compress(SCORERS[kind], ratio, att, hidden, keys, values, kwargs), args.reps)
            t_hip, m_hip, (k2, v2) = timed(lambda: press.compress(att, hidden, keys, values, None, kwargs), args.reps)
        # retained sets: recover ours from the kept keys' positions is not possible without indices -> compare via scores
        sc = press.score(att, hidden, keys, values, None, kwargs)
        from kvpress_amd import _native
        i2 = _native.topk_select(sc, k2.shape[2]).long()
        n = i1.shape[-1]
        inter = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i1.reshape(-1, n).cpu(), i2.reshape(-1, n).cpu())) / (i1.numel())
        print(json.dumps({"workload": wl, "torch_rocm_ms_per_layer": round(t_torch, 3), "hip_ms_per_layer": round(t_hip, 4),
                          "speedup": round(t_torch / t_hip, 1), "torch_peak_extra_MiB": round(m_torch), "hip_peak_extra_MiB": round(m_hip),
                          "kept_set_overlap_vs_bf16_torch_path": round(inter, 4), "n_kept": int(n), "reps": args.reps,
                          "torch": torch.__version__}), flush=True)
        del keys, values, hidden, k1, v1, k2, v2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
