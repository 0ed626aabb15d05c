#!/usr/bin/env python3
"""bench.py -- the kvpress score -> top-k -> gather hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload snapkv128k|knorm32k|knorm128k|ea128k]

A "step" is ONE pass of the hot path over one batch of synthetic input: one layer's
``press.compress()`` (score + top-k + gather) for B=1 per GPU, Llama-3.1-8B attention geometry
(H_q=32, H_kv=8, D=128, hidden 4096, bf16, llama3 RoPE), inputs resident in HBM.
Default workload = BASELINE.json's metric configuration (configs[2]): SnapKVPress(0.5), S=131072.

metric  : press ms/layer (``ms_per_step``) and press-only prefill tok/s (``value`` =
          n_gpus * S / (32 layers * t_layer)), as BASELINE.json / SURVEY.md §8(d) define them.
roofline: the dominant library kernel (largest average duration, HIP events on its launch
          stream via kvp_prof_*), its algorithmic bytes / duration vs the 8 TB/s HBM peak;
          ``path`` holds the same for the whole compress() against SURVEY §8(d)'s
          algorithmic bytes per layer.
cpu_baseline: the numpy oracle (oracle/kvpress_oracle.py, a port of the reference algorithm)
          timed on this box's host cores on the same workload (N=1, rank 0 only).
Multi-GPU: one process per GPU (torch.distributed.run), batch sharded one element per GPU, no
collective on the data path; only the timing is max-reduced over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

LAYERS = 32  # Llama-3.1-8B
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md); with real operands the chip power-limits to ~1.44 GHz

WORKLOADS = {
    # name: (press kind, S, ratio)
    "snapkv128k": ("snapkv", 131072, 0.5),
    "knorm32k": ("knorm", 32768, 0.5),
    "knorm128k": ("knorm", 131072, 0.5),
    "ea128k": ("ea", 131072, 0.7),
}
H_Q, H_KV, D, HIDDEN, WINDOW = 32, 8, 128, 4096, 64


def shard_batch(global_batch: int, world: int, rank: int):
    """Batch elements owned by ``rank`` (contiguous split; the path has no cross-element data)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate_time(local_seconds: float, world: int) -> float:
    """Step time of the job = max over ranks (all_reduce MAX); identity for one process."""
    if world == 1:
        return local_seconds
    import torch
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([local_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def algorithmic_bytes(kind: str, S: int, ratio: float, B: int = 1) -> dict:
    """bytes per layer the path must move (e = 2 bytes)."""
    n_kept = int(S * (1 - ratio))
    kread = B * S * H_KV * D * 2
    gather = B * 4 * n_kept * H_KV * D * 2  # read kept K,V rows + write K',V'
    extra = 0
    if kind == "ea":
        extra = B * (S * H_KV * D * 2 + S * H_Q * D * 2)  # V for ||v||, Q for the statistics
    return {"n_kept": n_kept, "score_read": kread + extra, "gather": gather, "total": kread + extra + gather}


def kernel_bytes(name: str, kind: str, S: int, ratio: float) -> float:
    """Algorithmic bytes of ONE launch of a library kernel (B=1)."""
    ab = algorithmic_bytes(kind, S, ratio)
    kbytes = S * H_KV * D * 2
    if name.startswith("gather"):
        return ab["gather"]
    if name.startswith(("snapkv_p1", "snapkv_p2", "rownorm", "ea_logits")):
        return kbytes
    return 0.0


def kernel_flops(name: str, S: int) -> float:
    """Algorithmic matrix-core flops of ONE launch (B=1; SURVEY §8d: 2 * H_q * W * S * D per QK^T pass)."""
    if name.startswith(("snapkv_p1", "snapkv_p2")):
        return 2.0 * H_Q * WINDOW * S * D
    return 0.0


def pmc_traffic(kernel_name: str, workload: str):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary of this same command
    (profiles/r01_pmc_summary_<workload>.txt; separate --pmc passes).  MI355X_MICROARCH.md §HBM: FETCH_SIZE and
    WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled."""
    path = os.path.join(ROOT, "profiles", f"r01_pmc_summary_{workload}.txt")
    if not os.path.exists(path):
        return None
    fetch = write = None
    inside = False
    for line in open(path):
        if line.startswith("=="):
            inside = kernel_name in line
        elif inside:
            f = line.split()
            if len(f) == 2 and f[0] == "FETCH_SIZE":
                fetch = float(f[1])
            if len(f) == 2 and f[0] == "WRITE_SIZE":
                write = float(f[1])
    if fetch is None or write is None:
        return None
    return int((2.0 * fetch + write) * 1024)


def build_module(device):
    import torch
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaAttention, LlamaRotaryEmbedding

    cfg = LlamaConfig(
        hidden_size=HIDDEN, num_attention_heads=H_Q, num_key_value_heads=H_KV, head_dim=D, num_hidden_layers=1,
        intermediate_size=14336, vocab_size=128, max_position_embeddings=131072 * 2, rope_theta=500000.0,
        rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 8192},
        attention_bias=False,
    )
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    att = LlamaAttention(cfg, layer_idx=0).to(device=device, dtype=torch.bfloat16)
    rot = LlamaRotaryEmbedding(cfg).to(device)
    att.rotary_emb = rot
    return att, rot


def make_press(kind, ratio):
    import kvpress_amd as P

    if kind == "snapkv":
        return P.SnapKVPress(compression_ratio=ratio, window_size=WINDOW, kernel_size=5)
    if kind == "knorm":
        return P.KnormPress(compression_ratio=ratio)
    return P.ExpectedAttentionPress(compression_ratio=ratio)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="snapkv128k", choices=list(WORKLOADS))
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed device pre-warm before the W warm-up steps: repeat the step for this long so that the clocks have "
                         "ramped (they take ~100 steps; with a short warm-up the same build reads 10 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-json", default=None, help="also dump the per-kernel HIP-event table here")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from kvpress_amd import _native

    _native.lib()  # fail loudly if the HIP extension is missing
    kind, S, ratio = WORKLOADS[args.workload]
    lo, hi = shard_batch(world, world, rank)  # global batch = one element per GPU (weak scaling)
    B = hi - lo
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + lo)
    bf = torch.bfloat16
    keys = torch.randn((B, H_KV, S, D), generator=gen, device=device, dtype=torch.float32).to(bf)
    values = torch.randn((B, H_KV, S, D), generator=gen, device=device, dtype=torch.float32).to(bf)
    hidden = torch.randn((B, S, HIDDEN), generator=gen, device=device, dtype=bf)
    att, rot = build_module(device)
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=device)[None])
    kwargs = {"position_embeddings": pe}
    press = make_press(kind, ratio)

    def step():
        with torch.no_grad():
            return press.compress(att, hidden, keys, values, None, kwargs)

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:  # untimed, back to back: lets the clock governor settle
        for _ in range(25):
            out = step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    local = time.perf_counter() - t0
    total = aggregate_time(local, world)
    t_step = total / args.steps
    n_kept = int(S * (1 - ratio))
    assert tuple(out[0].shape) == (B, H_KV, n_kept, D), out[0].shape

    # ---- per-kernel HIP-event timing (profiling on: separate from the timed region) -------------
    roofline, kern_table = None, {}
    if rank == 0:
        _native.prof_enable(True)
        nprof = 5
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        for name, ms in _native.prof_records():
            kern_table.setdefault(name, []).append(ms)
        _native.prof_enable(False)
        avg = {k: (sum(v) / len(v), len(v) / nprof) for k, v in kern_table.items()}
        cand = {k: a for k, (a, _) in avg.items() if kernel_bytes(k, kind, S, ratio) > 0}
        ab = algorithmic_bytes(kind, S, ratio)
        if cand:
            dom = max(cand, key=cand.get)
            kb = kernel_bytes(dom, kind, S, ratio) * B
            ach = kb / (cand[dom] * 1e-3) / 1e9
            roofline = {
                "kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.workload),
                "algorithmic_bytes_per_launch": kb, "avg_launch_us": round(cand[dom] * 1e3, 2),
                # secondary bound of the same kernel: the window-attention passes are matrix-core / VALU work
                "mfma": ({"achieved": round(kernel_flops(dom, S) * B / (cand[dom] * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": round(kernel_flops(dom, S) * B / (cand[dom] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
                         if kernel_flops(dom, S) else None),
                "path": {
                    "algorithmic_bytes_per_layer": ab["total"] * B,
                    "achieved": round(ab["total"] * B / t_step / 1e9, 1),
                    "frac": round(ab["total"] * B / t_step / 1e9 / HBM_PEAK_GBS, 4),
                    "kernels_us": {k: round(a * 1e3 * c, 2) for k, (a, c) in sorted(avg.items())},
                    "kernels_sum_us": round(sum(a * c for a, c in avg.values()) * 1e3, 2),
                },
            }
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump({"workload": args.workload, "ms_per_step": t_step * 1e3,
                           "kernels_avg_ms": {k: a for k, (a, _) in avg.items()},
                           "launches_per_step": {k: c for k, (_, c) in avg.items()}}, f, indent=1)

    # ---- CPU baseline: the numpy oracle on the same workload (rank 0, N=1 only) ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import kvpress_oracle as O

        k_np = keys.float().cpu().numpy()
        v_np = values.float().cpu().numpy()
        f32 = np.float32
        sample = f"full {args.workload} layer (B=1, H_kv={H_KV}, S={S}), numpy float32 port of the reference"
        with torch.no_grad():
            if kind == "snapkv":
                q_np = press.compute_window_queries(att, hidden, WINDOW, pe).float().cpu().numpy()
            elif kind == "ea":
                # the statistics need all S queries (1 GiB in fp32 per 8 heads): sample 1/8 of the heads' work
                mu, cov = press.get_query_statistics(att, hidden)
                mu_np, cov_np = mu.cpu().numpy(), cov.cpu().numpy()
                sample += "; query statistics taken from the GPU (kernel-only baseline)"
        t0 = time.perf_counter()
        if kind == "snapkv":
            sc = O.snapkv_score(q_np, k_np, 5, ctype=f32)
        elif kind == "knorm":
            sc = O.knorm_score(k_np, ctype=f32)
        else:
            sc = O.ea_score(k_np, v_np, mu_np, cov_np, 4, True, 0.0, ctype=f32)
        ko, vo, idx = O.compress(sc, k_np, v_np, ratio)
        t_cpu = time.perf_counter() - t0
        # cross-check while we are here: GPU retained set is a valid top-k of the oracle's scores
        gsc = press.score(att, hidden, keys, values, None, kwargs)
        gidx = _native.topk_select(gsc, n_kept).cpu().numpy()
        ok, msg = O.topk_is_valid(sc, gidx, n_kept, rel_band=1e-3)
        try:  # threads the port really used: numpy's BLAS pool for the matrix products, one thread for everything else
            from threadpoolctl import threadpool_info

            blas_threads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
        except Exception:
            blas_threads = 1
        sample += f"; numpy ({blas_threads} BLAS threads for the matrix products, 1 thread elsewhere; {os.cpu_count()} cores visible)"
        cpu = {"value": round(S / (LAYERS * t_cpu), 1), "unit": "tok/s", "cores": blas_threads, "kind": "port",
               "sample": sample, "ms_per_layer": round(t_cpu * 1e3, 1), "gpu_topk_valid_vs_oracle": bool(ok)}

    if rank == 0:
        value = world * B * S / (LAYERS * t_step)
        line = {
            "metric": "press ms/layer + prefill tok/s, Llama-3.1-8B 128k ctx, SnapKV ratio=0.5" if args.workload == "snapkv128k"
            else f"press ms/layer + prefill tok/s, Llama-3.1-8B, {args.workload}",
            "value": round(value, 1), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": args.workload, "press": kind, "compression_ratio": ratio, "batch_per_gpu": B,
                       "seq_len": S, "n_kept": n_kept, "h_q": H_Q, "h_kv": H_KV, "head_dim": D, "layers_for_tok_s": LAYERS,
                       "parallelism": f"batch-sharded x{world}, no collective", "prewarm_ms": args.prewarm_ms},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
