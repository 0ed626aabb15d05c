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
cpu_baseline: the reference's own op sequence in plain PyTorch (oracle/torch_path.py, pinned bit for bit to the
          real reference by tests/test_oracle_golden.py) timed on this box's host cores on the same workload, bf16
          (as users run it) and float32, torch.get_num_threads() threads (N=1, rank 0 only); the numpy oracle's time
          is kept as a secondary field.
Multi-GPU: one process per GPU, batch sharded one element per GPU, no collective on the data path; only
          the timing is max-reduced over ranks.  `python bench.py --gpus N` launches the N ranks itself (it re-executes
          under torch.distributed.run on 127.0.0.1 when WORLD_SIZE is not set); under an external launcher it reads
          RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
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
    if name.startswith("ea_qstats_mfma"):
        return S * H_Q * D * 2  # Q [B, S, H_q * D] read once for the statistics
    return 0.0


def kernel_flops(name: str, S: int) -> float:
    """Algorithmic matrix-core flops of ONE launch (B=1; SURVEY §8d: 2 * H_q * W * S * D per QK^T pass)."""
    if name.startswith(("snapkv_p1", "snapkv_p2")):
        return 2.0 * H_Q * WINDOW * S * D
    if name.startswith("ea_logits"):
        return 2.0 * H_Q * S * D * D  # k^T Sigma k per key and query head (the kernel's hi + lo split of Sigma executes twice that)
    return 0.0


def csrc_digest() -> str:
    """sha256 over the kernel sources: a PMC summary is only quoted for the build it was measured on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "kvpress_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name: str, workload: str):
    """(HBM bytes per launch of `kernel_name`, provenance).  The counters cannot be read from inside this process: they come
    from the rocprofv3 --pmc passes of this same command (`scripts/gpu_check.sh pmc`, separate passes, no tracing), whose
    summary is committed as profiles/r02_pmc_summary_<workload>.txt together with the digest of the kernel sources it was
    measured on.  A summary of a different build is NOT quoted (traffic = null).  MI355X_MICROARCH.md §HBM: FETCH_SIZE and
    WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled."""
    rel = os.path.join("profiles", f"r02_pmc_summary_{workload}.txt")
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f"{rel} missing"
    fetch = write = None
    inside = False
    digest = None
    for line in open(path):
        if line.startswith("# csrc_digest"):
            digest = line.split()[-1]
        if line.startswith("=="):
            inside = kernel_name in line
        elif inside:
            f = line.split()
            if len(f) == 2 and f[0] == "FETCH_SIZE":
                fetch = float(f[1])
            if len(f) == 2 and f[0] == "WRITE_SIZE":
                write = float(f[1])
    if digest != csrc_digest():
        return None, f"{rel} is from another build (digest {digest}, this build {csrc_digest()}): not quoted"
    if fetch is None or write is None:
        return None, f"{rel} has no FETCH_SIZE / WRITE_SIZE for {kernel_name}"
    return int((2.0 * fetch + write) * 1024), f"{rel} (rocprofv3 --pmc passes of this command on this build, digest {digest})"


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


def launcher_argv(n_gpus: int, argv: list) -> list:
    """Command line that turns `python bench.py --gpus N ...` into N ranks on this node: one process per GPU, rendezvous on
    127.0.0.1 (the container hostname may not resolve), a free port.  Matches the reference's process-per-GPU model
    (evaluation/evaluate.sh:15-28); the ranks exchange nothing but the step time."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def timed_steps(step, steps: int, warmup: int, world: int, sync) -> float:
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides; returns this rank's seconds."""
    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    return time.perf_counter() - t0


def result_line(args, world: int, B: int, S: int, t_step: float, config: dict, roofline, cpu, metric: str, dtype: str = "bf16") -> dict:
    assert world == args.gpus
    return {"metric": metric, "value": round(world * B * S / (LAYERS * t_step), 1), "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic", "config": config, "roofline": roofline, "cpu_baseline": cpu}


def cpu_baseline(workload, press, att, rot, hidden, keys, values, kwargs, n_kept):
    """The reference's pure-PyTorch path on the host CPU cores of THIS box, same run, same tensors (north_star):
    oracle/torch_path.py = ScorerPress.compress (scorer_press.py:76-102) + the scorer, restated op for op and pinned bit for
    bit to the real reference (tests/test_oracle_golden.py).  bf16 as users run it and float32 ("O32"); 1 warm-up + timed
    runs, median; threads = torch.get_num_threads().  ExpectedAttention's 4.4-TFLOP q_proj makes one run ~10 s: one timed
    run per dtype there.  The numpy float32 port (oracle/kvpress_oracle.py) is timed once as a secondary figure, and the GPU
    path's retained set is checked against the float32 CPU scores while they are at hand."""
    import numpy as np
    import torch

    from kvpress_amd import _native
    from oracle import kvpress_oracle as O
    from oracle import torch_path as TP

    kind, S, ratio = WORKLOADS[workload]
    threads = torch.get_num_threads()
    n_timed = 1 if kind == "ea" else 3
    att_cpu = {torch.bfloat16: build_module(torch.device("cpu"))[0]}
    att_cpu[torch.float32] = build_module(torch.device("cpu"))[0].float()
    rot_cpu = build_module(torch.device("cpu"))[1]
    res, sc32 = {}, None
    with torch.no_grad():
        for dt in (torch.bfloat16, torch.float32):
            m = att_cpu[dt]
            m.rotary_emb = rot_cpu
            h, k, v = hidden.cpu().to(dt), keys.cpu().to(dt), values.cpu().to(dt)
            kw = {"position_embeddings": tuple(t.cpu().to(dt) for t in kwargs["position_embeddings"])}
            times = []
            for i in range(1 + n_timed):
                t0 = time.perf_counter()
                ko, vo, idx = TP.torch_compress(TP.SCORERS[kind], ratio, m, h, k, v, kw)
                if i:
                    times.append(time.perf_counter() - t0)
            assert tuple(ko.shape) == (k.shape[0], H_KV, n_kept, D)
            res[dt] = sorted(times)[len(times) // 2]
            if dt == torch.float32:
                sc32 = TP.SCORERS[kind](m, h, k, v, kw).numpy()
            del h, k, v, ko, vo
    # GPU retained set vs the float32 CPU scores (tie-tolerant, 1e-3 band)
    gsc = press.score(att, hidden, keys, values, None, kwargs)
    gidx = _native.topk_select(gsc, n_kept).cpu().numpy()
    ok, _ = O.topk_is_valid(sc32, gidx, n_kept, rel_band=1e-3)
    # secondary: the numpy float32 port of the same algorithm (the round-1 baseline)
    port_ms = None
    if kind != "ea":
        k_np, v_np = keys.float().cpu().numpy(), values.float().cpu().numpy()
        if kind == "snapkv":
            with torch.no_grad():
                q_np = press.compute_window_queries(att, hidden, WINDOW, kwargs["position_embeddings"]).float().cpu().numpy()
        t0 = time.perf_counter()
        sc = O.snapkv_score(q_np, k_np, 5, ctype=np.float32) if kind == "snapkv" else O.knorm_score(k_np, ctype=np.float32)
        O.compress(sc, k_np, v_np, ratio)
        port_ms = round((time.perf_counter() - t0) * 1e3, 1)
    t = res[torch.bfloat16]
    return {"value": round(S / (LAYERS * t), 1), "unit": "tok/s", "cores": threads, "kind": "torch-restatement",
            "sample": f"one full {workload} layer (B=1, H_kv={H_KV}, S={S}) per run: score + topk + gather of the reference in plain PyTorch "
                      f"(oracle/torch_path.py), bf16 module and tensors, 1 warm-up + {n_timed} timed run(s), median; "
                      f"torch.get_num_threads() = {threads}, os.cpu_count() = {os.cpu_count()}; tok/s extrapolates one layer x {LAYERS}",
            "ms_per_layer": round(t * 1e3, 1), "ms_per_layer_fp32": round(res[torch.float32] * 1e3, 1),
            "numpy_port_ms_per_layer": port_ms, "gpu_topk_valid_vs_cpu_fp32_scores": bool(ok)}


def stub_main(args, world: int, rank: int):
    """(tests) The N-rank path of this file -- launcher, rank environment, batch sharding, barrier-bracketed timing, MAX over ranks,
    one JSON line from rank 0 -- with the press replaced by a host sleep; runs under gloo on CPU."""
    kind, S, ratio = WORKLOADS[args.workload]
    lo, hi = shard_batch(world, world, rank)
    B = hi - lo
    local = timed_steps(lambda: time.sleep(args.stub_step * 1e-3 * (1 + rank)), args.steps, args.warmup, world, lambda: None)
    t_step = aggregate_time(local, world) / args.steps
    if rank == 0:
        cfg = {"workload": args.workload, "press": kind, "batch_per_gpu": B, "seq_len": S, "stub_step_ms": args.stub_step,
               "parallelism": f"batch-sharded x{world}, no collective"}
        print(json.dumps(result_line(args, world, B, S, t_step, cfg, None, None, "stub (launcher test)", dtype="none")), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


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
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL); gloo + --stub-step exercises the launcher on CPU (tests)")
    ap.add_argument("--stub-step", type=float, default=None, metavar="MS",
                    help="(tests) replace the press by a host sleep of MS milliseconds: launcher / sharding / timing path without a GPU")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N`, no launcher around it: become N ranks (one process per GPU, torch.distributed.run) ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.execvp(sys.executable, launcher_argv(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU"
    stub = args.stub_step is not None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if stub:
        return stub_main(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    assert torch.cuda.device_count() > local_rank, f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible on this node"
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
    local = timed_steps(step, args.steps, args.warmup, world, torch.cuda.synchronize)
    out = step()
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
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args.workload)[0],
                "traffic_source": pmc_traffic(dom, args.workload)[1],
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
            m = roofline["mfma"]
            if m and m["frac"] > roofline["frac"]:   # the matrix-core roof is the nearer one (ExpectedAttention's quadratic form)
                roofline["hbm"] = {k: roofline[k] for k in ("achieved", "peak", "unit", "frac")}
                roofline.update(bound="mfma", achieved=m["achieved"], peak=m["peak"], unit=m["unit"], frac=m["frac"])
        if args.profile_json:
            with open(args.profile_json, "w") as f:
                json.dump({"workload": args.workload, "ms_per_step": t_step * 1e3,
                           "kernels_avg_ms": {k: a for k, (a, _) in avg.items()},
                           "launches_per_step": {k: c for k, (_, c) in avg.items()}}, f, indent=1)

    # ---- CPU baseline (rank 0, N=1 only): the reference's op sequence in plain PyTorch on this box's host cores -------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, press, att, rot, hidden, keys, values, kwargs, n_kept)

    if rank == 0:
        metric = ("press ms/layer + prefill tok/s, Llama-3.1-8B 128k ctx, SnapKV ratio=0.5" if args.workload == "snapkv128k"
                  else f"press ms/layer + prefill tok/s, Llama-3.1-8B, {args.workload}")
        cfg = {"workload": args.workload, "press": kind, "compression_ratio": ratio, "batch_per_gpu": B, "seq_len": S, "n_kept": n_kept,
               "h_q": H_Q, "h_kv": H_KV, "head_dim": D, "layers_for_tok_s": LAYERS, "parallelism": f"batch-sharded x{world}, no collective",
               "prewarm_ms": args.prewarm_ms}
        print(json.dumps(result_line(args, world, B, S, t_step, cfg, roofline, cpu, metric)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
