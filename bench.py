#!/usr/bin/env python3
"""bench.py -- the kvpress score -> top-k -> gather hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload snapkv128k|knorm32k|knorm128k|ea128k|<f-row workload>]

A "step" is ONE pass of the hot path over one batch of synthetic input: one layer's
``press.compress()`` (score + top-k + gather) for B=1 per GPU, Llama-3.1-8B attention geometry
(H_q=32, H_kv=8, D=128, hidden 4096, bf16, llama3 RoPE), inputs resident in HBM.
Default workload = BASELINE.json's metric configuration (configs[2]): SnapKVPress(0.5), S=131072.

metric  : press ms/layer (``ms_per_step``) and press-only prefill tok/s (``value`` =
          n_gpus * S / (32 layers * t_layer)), as BASELINE.json / SURVEY.md §8(d) define them.
roofline: the dominant library kernel (largest average duration), its algorithmic bytes / duration vs the 8 TB/s HBM
          peak.  Kernel durations are rocprofv3's: the run spawns one `rocprofv3 --kernel-trace` pass around a child run of the
          same workload and averages every kernel's dispatches from its rocpd database (what profiles/rNN_rocprofv3_kernel_stats_*.csv
          summarises); the library's HIP-event table (kvp_prof_*, 3-6 us higher per kernel) stays as ``path.kernels_us_events`` and is
          the fallback where no profiler pass is possible (N > 1, children, under a profiler).  At the same level:
          ``path_frac`` (the whole compress() against SURVEY §8(d)'s algorithmic bytes per layer: THE number the
          north-star target of 0.70 is about) and the window-attention passes' own fractions (``p1_frac``, ``p2_frac``).
          ``path`` holds the details, among them ``path.model``: the floor of this kernel chain from measured ceilings (per kernel
          max(bytes / 6.29 TB/s copy ceiling, matrix-core flops / 1.75 PFLOP/s sustained on random operands, one
          1.7 us dependent-launch boundary), summed -- path_model()) with ``frac_of_step``.
cpu_baseline: the REFERENCE ITSELF (``kind: "reference"``: NVIDIA/kvpress's press.compress(), imported from oracle/_ref, a
          build-time copy made by oracle/build_ref.py where /root/reference exists) timed on this box's host cores on the
          same workload, bf16 (as users run it) and float32, torch.get_num_threads() threads (N=1, rank 0 only), and
          cross-checked bit for bit against oracle/torch_path.py, the op-for-op restatement (pinned to the real reference by
          tests/test_oracle_golden.py) that is the fallback (``kind: "port"``) where the copy is absent.
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
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md); with real operands the chip sits at its 1.4 kW power limit at ~1.75 GHz
# measured ceilings the path model is built from:
COPY_CEILING_GBS = 6290.0       # float4 copy ceiling, MI355X_MICROARCH.md "HBM": what a pure streaming kernel reaches
MFMA_SUSTAINED_TFLOPS = 1752.0  # v_mfma_f32_32x32x16_bf16 back to back on all 256 CUs with RANDOM operands, SUSTAINED: 306.4 ns per stage of
                                # 536.9 MFLOP (profiles/r05_clock_power.txt, "ubench:mfma_only:256wg_random": 1305 W, 1.75 GHz in-kernel).  Rounds 2-4
                                # used the one-shot launch of the same micro-benchmark (367.9 ns = 1459: the first launch after idle clocks lower)
BOUNDARY_US = 1.7               # one dependent kernel boundary / all-to-all hop (MI355X_MICROARCH.md price list "boundary"; the cluster
                                # select's in-launch hops measure ~2 us each, profiles/r03_select_cluster_lab.txt)

H_Q, H_KV, D, HIDDEN, WINDOW = 32, 8, 128, 4096, 64
WORKLOADS = {
    # name: (press kind, S, ratio)            BASELINE.json configs 2-4 + Knorm at 128k
    "snapkv128k": ("snapkv", 131072, 0.5),
    "knorm32k": ("knorm", 32768, 0.5),
    "knorm128k": ("knorm", 131072, 0.5),
    "ea128k": ("ea", 131072, 0.7),
    # SURVEY §8(f) rows on the same tensors: the scorers / wrappers that reuse the path's kernels
    "keydiff128k": ("keydiff", 131072, 0.5),             # keydiff_press.py:45-46
    "cur128k": ("cur", 131072, 0.5),                     # cur_press.py:42-65 (kv_product leverage, local windows of 16)
    "finch128k": ("finch", 131072, 0.5),                 # finch_press.py:56-83 (window 64, normalised, kept keys re-rotated)
    "chunk_snapkv128k": ("chunk_snapkv", 131072, 0.5),   # chunk_press.py:67-85: SnapKV per 1024-token chunk, segmented select
    "rerotate128k": ("rerotate", 131072, 0.5),           # key_rerotation_press.py:101-152 around KnormPress
    "decode_snapkv2k": ("snapkv", 2048, 0.5),            # decoding_press.py:113-179's regime: a 2k-token cache, latency-bound
    # the headline workload with K' / V' stored in the REFERENCE's tensor layout (descending score, scorer_press.py:95-100:
    # press.kept_order = "score"): the fused compress call + the hand-written sort (topk_order.hip) + a gather of rows in score order
    "snapkv128k_scoreorder": ("snapkv", 131072, 0.5),
    # more than one batch element per GPU (scorer_press.py:76-102 is batch-generic): B = BATCH[workload]
    "snapkv128k_b2": ("snapkv", 131072, 0.5),
    "knorm128k_b4": ("knorm", 131072, 0.5),
}
BATCH = {"snapkv128k_b2": 2, "knorm128k_b4": 4}   # batch elements PER GPU of a workload (default 1)
ROUND = "r06"   # prefix of the committed profiles/ this build's fallbacks read
# CPU seeds of the timed tensors (SURVEY §8d set A: N(0,1) from a CPU torch.Generator, rounded to bf16 once): the headline and
# config 2 use the seeds of their full-size parity fixtures (tests/_fullsize.py FULL_CASES), so the timed tensors ARE the tensors
# whose retained sets are pinned to the real reference (tests/golden/full_snapkv128k.npz, full_knorm32k.npz)
SEEDS = {"snapkv128k": 103, "snapkv128k_scoreorder": 103, "snapkv128k_b2": 103, "knorm32k": 102, "ea128k": 104}
FIXTURES = {"snapkv128k": "full_snapkv128k", "knorm32k": "full_knorm32k", "ea128k": "full_ea128k_A"}   # reference outputs for exactly the timed tensors


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


def n_kept_of(kind: str, S: int, ratio: float) -> int:
    if kind == "chunk_snapkv":   # chunk_press.py:79: per 1024-token chunk
        return (S // 1024) * max(1, int(1024 * (1 - ratio))) + (max(1, int((S % 1024) * (1 - ratio))) if S % 1024 else 0)
    return int(S * (1 - ratio))


def algorithmic_bytes(kind: str, S: int, ratio: float, B: int = 1) -> dict:
    """bytes per layer the path must move (e = 2 bytes): K read once for scoring (+ what else the scorer
    must read) + kept K, V rows read + K', V' written (the re-rotating presses rotate the kept keys on their way through the
    gather: no extra pass)."""
    n_kept = n_kept_of(kind, S, ratio)
    kread = B * S * H_KV * D * 2
    gather = B * 4 * n_kept * H_KV * D * 2  # read kept K,V rows + write K',V'
    extra = 0
    if kind == "ea":
        extra = B * (S * H_KV * D * 2 + S * H_Q * D * 2)  # V for ||v||, Q for the statistics
    if kind == "cur":
        extra = B * S * H_KV * D * 2                      # V for its leverage scores
    return {"n_kept": n_kept, "score_read": kread + extra, "gather": gather, "total": kread + extra + gather}


def kernel_bytes(name: str, kind: str, S: int, ratio: float) -> float:
    """Algorithmic bytes of ONE launch of a library kernel (B=1)."""
    ab = algorithmic_bytes(kind, S, ratio)
    kbytes = S * H_KV * D * 2
    if name.startswith("gather"):
        return ab["gather"]
    if name.startswith(("snapkv_p1", "snapkv_p2", "rownorm", "ea_logits", "ea_vnorm_finalize", "keydiff_anchor_kernel", "keydiff_score", "colsumsq", "rowdot")):
        return kbytes   # one pass over K (or V)
    if name.startswith("ea_qstats_mfma"):
        return S * H_Q * D * 2  # Q [B, S, H_q * D] read once for the statistics
    if name.startswith("qproj_rope"):
        return HIDDEN * H_Q * D * 2   # the q_proj weight, streamed once (the 512 KiB hidden window is re-read from L2)
    if name.startswith("rerotate"):
        return 2 * ab["n_kept"] * H_KV * D * 2
    if name.startswith("topk_cluster") and kind == "knorm":
        return kbytes   # fused Knorm compress: the cluster select computes the norms itself (one pass over K)
    return 0.0


def kernel_flops(name: str, S: int) -> float:
    """Algorithmic matrix-core flops of ONE launch (B=1; SURVEY §8d: 2 * H_q * W * S * D per QK^T pass)."""
    if name.startswith(("snapkv_p1", "snapkv_p2")):
        return 2.0 * H_Q * WINDOW * S * D
    if name.startswith("ea_logits"):
        return 2.0 * H_Q * S * D * D  # k^T Sigma k per key and query head (the kernel's hi + lo split of Sigma executes twice that)
    if name.startswith("ea_qstats_mfma"):
        return 2.0 * H_Q * S * D * D  # covariance (syrk executes half of it)
    return 0.0


def path_model(kernels: dict, kind: str, S: int, ratio: float, B: int = 1) -> dict:
    """Floor of THIS kernel chain from measured ceilings: per launch max(algorithmic bytes / copy ceiling, matrix-core flops /
    sustained rate on random operands, one dependent-launch boundary), summed over the launches of a step; the cluster select
    counts its four digit / compaction steps as four boundaries (they are all-to-all hops of ~2 us each whether they are
    kernel boundaries or in-launch barriers).  Host-side torch ops that the library does not launch (the window q_proj of the
    SnapKV-type presses: 32 MiB of weight; ExpectedAttention's full-sequence q_proj: 2 * S * hidden^2 flops) are added as
    ``torch_ops``.  `kernels`: {name: (avg ms per launch, launches per step)}."""
    per = {}
    for name, (_, count) in kernels.items():
        t_bytes = kernel_bytes(name, kind, S, ratio) * B / (COPY_CEILING_GBS * 1e3)     # us
        t_flops = kernel_flops(name, S) * B / (MFMA_SUSTAINED_TFLOPS * 1e6)             # us
        hops = 4 if name.startswith("topk_cluster") else 1
        per[name] = round(max(t_bytes, t_flops, hops * BOUNDARY_US) * count, 2)
    torch_ops = 0.0
    if kind in ("snapkv", "finch", "chunk_snapkv") and not any(k.startswith("qproj_rope") for k in kernels):
        torch_ops = max(HIDDEN * H_Q * D * 2 / (COPY_CEILING_GBS * 1e3), BOUNDARY_US)   # the model's own window q_proj: 32 MiB of weight
    if kind == "ea":
        # the model's full-sequence q_proj is a library GEMM: priced at the DENSE PEAK (a tuned GEMM on this chip has been
        # observed above the 1.46 PF this file uses for the hand-written passes, and a floor must never be beaten: VERDICT r3 #9)
        torch_ops = 2.0 * B * S * HIDDEN * H_Q * D / (MFMA_PEAK_TFLOPS * 1e6)
    total = sum(per.values()) + torch_ops
    return {"per_kernel_us": per, "torch_ops_us": round(torch_ops, 2), "total_us": round(total, 2),
            "ceilings": {"copy_GBs": COPY_CEILING_GBS, "mfma_sustained_TFLOPs": MFMA_SUSTAINED_TFLOPS, "boundary_us": BOUNDARY_US}}


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
    summary is committed as profiles/<round>_pmc_summary_<workload>.txt together with the digest of the kernel sources it was
    measured on.  A summary of a different build is NOT quoted (traffic = null).  MI355X_MICROARCH.md §HBM: FETCH_SIZE and
    WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled."""
    rel = os.path.join("profiles", f"{ROUND}_pmc_summary_{workload}.txt")
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


def under_profiler() -> bool:
    """rocprofv3 (or another rocprofiler tool) around this process: its kernel statistics must only see the requested workload"""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or any(
        k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)


def live_pmc_traffic(kernel_name: str, workload: str, timeout_s: float = 240.0):
    """(HBM bytes per launch of `kernel_name`, provenance) measured NOW: this run spawns `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` passes (separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes) around a 3-step child run of this
    same script and workload, and reads the per-dispatch counters from the rocpd databases (scripts/rocpd_pmc.py).  None when rocprofv3 is
    not on PATH, when this process is itself a child / already under a profiler, or when a pass fails -- the caller then falls back to the
    committed summary of the same build."""
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("KVP_BENCH_CHILD") == "1" or os.environ.get("KVP_BENCH_LIVE_PMC") == "0":
        return None, "live PMC passes disabled for this process"
    if under_profiler():
        return None, "already running under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import rocpd_pmc
    finally:
        sys.path.pop(0)
    vals = {}
    env = dict(os.environ, KVP_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="kvp_pmc_") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--workload", workload, "--steps", "3", "--warmup", "1", "--prewarm-ms", "5", "--no-cpu-baseline"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=max(30.0, timeout_s - (time.perf_counter() - t0)))
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f"live {counter} pass failed: {type(e).__name__}"
            if r.returncode != 0:
                return None, f"live {counter} pass exited {r.returncode}"
            dbs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            per = {}
            for db in dbs:
                try:
                    loaded = rocpd_pmc.load(db)
                except Exception as e:   # noqa: BLE001 (a profiler output this parser does not know: fall back)
                    return None, f"live {counter} pass: cannot read {os.path.basename(db)} ({type(e).__name__})"
                for name, cs in loaded.items():
                    if kernel_name in name and counter in cs:
                        per.update(cs[counter])
            if not per:
                return None, f"live {counter} pass recorded no dispatch of {kernel_name}"
            vals[counter] = sum(per.values()) / len(per)
    # MI355X_MICROARCH.md section HBM: both counters are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), (
        f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes spawned by this run (3 steps each, {time.perf_counter() - t0:.0f} s), "
        "averaged over the kernel's dispatches; FETCH_SIZE doubled per the guide's gfx950 correction")


def match_kernel(short: str, display: str) -> bool:
    """does the profiler's display name (`void (anonymous namespace)::gather_vec_kernel<16, true>(...)`) belong to the library's launch
    name (`gather_vec_kernel`)?"""
    i = display.find(short)
    return i >= 0 and (i == 0 or not (display[i - 1].isalnum() or display[i - 1] == "_")) and display[i + len(short):i + len(short) + 1] in ("<", "(", "")


def rocpd_kernel_durations(db_path: str) -> dict:
    """{kernel display name: [dispatch durations in us]} from a rocprofv3 rocpd database (the data `--stats` summarises)."""
    import sqlite3

    con = sqlite3.connect(db_path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    per = {}
    for name, st, en in cur.execute(f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                                    "on d.kernel_id = s.id order by d.start"):
        per.setdefault(name, []).append((en - st) / 1e3)
    con.close()
    return per


def write_kernel_stats_csv(per: dict, path: str) -> None:
    """rocprofv3 --stats' kernel table (same columns) from the dispatch durations of a trace database: so that the committed
    profiles/<round>_rocprofv3_kernel_stats_<workload>.csv is the very data the line's per-kernel numbers were averaged from."""
    rows = []
    for name, durs in per.items():
        ns = [d * 1e3 for d in durs]
        mean = sum(ns) / len(ns)
        sd = (sum((x - mean) ** 2 for x in ns) / max(1, len(ns) - 1)) ** 0.5
        rows.append((sum(ns), name, len(ns), mean, min(ns), max(ns), sd))
    tot = sum(r[0] for r in rows) or 1.0
    with open(path, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
        for total, name, n, mean, mn, mx, sd in sorted(rows, reverse=True):
            f.write(f'"{name}",{n},{total:.0f},{mean:.6f},{100 * total / tot:.2f},{mn:.0f},{mx:.0f},{sd:.6f}\n')


def live_kernel_trace(names, workload: str, timeout_s: float = 240.0, csv_path=None):
    """({library kernel name: (average us per launch, dispatches)}, provenance): kernel durations as rocprofv3 measures them -- the
    numbers profiles/<round>_rocprofv3_kernel_stats_<workload>.csv holds -- taken NOW by one `rocprofv3 --kernel-trace` pass (no
    counters) around a child run of this same script and workload; averages over ALL the child's dispatches of a kernel, as `--stats`
    does.  HIP events around a launch (kvp_prof_*) read 3-6 us more per kernel: the line's per-kernel numbers
    come from here whenever the pass is possible, the event table stays beside them.  None as live_pmc_traffic."""
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("KVP_BENCH_CHILD") == "1" or os.environ.get("KVP_BENCH_LIVE_PMC") == "0":
        return None, "live profiler passes disabled for this process"
    if under_profiler():
        return None, "already running under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    env = dict(os.environ, KVP_BENCH_CHILD="1", TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    child = ["--workload", workload, "--steps", "100", "--warmup", "10", "--prewarm-ms", "60", "--no-cpu-baseline"]
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="kvp_trace_") as tmp:
        cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "trace", "--", sys.executable, os.path.join(ROOT, "bench.py"), *child]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, f"live kernel-trace pass failed: {type(e).__name__}"
        if r.returncode != 0:
            return None, f"live kernel-trace pass exited {r.returncode}"
        per = {}
        for d, _, fs in os.walk(tmp):
            for f in fs:
                if f.endswith(".db"):
                    try:
                        for name, durs in rocpd_kernel_durations(os.path.join(d, f)).items():
                            per.setdefault(name, []).extend(durs)
                    except Exception as e:   # noqa: BLE001
                        return None, f"live kernel-trace pass: cannot read {f} ({type(e).__name__})"
    if csv_path:
        try:
            write_kernel_stats_csv(per, csv_path)
        except OSError as e:
            print(f"[bench] cannot write {csv_path}: {e}", file=sys.stderr)
    out = {}
    for short in names:
        durs = [x for name, ds in per.items() if match_kernel(short, name) for x in ds]
        if durs:
            out[short] = (sum(durs) / len(durs), len(durs))
    if not out:
        return None, "live kernel-trace pass recorded none of the library's kernels"
    try:   # the traced run's own step time (its JSON line): what the sum of ITS kernel durations is compared with
        child_line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out["__step_ms__"] = (float(child_line["ms_per_step"]), 0)
    except Exception:   # noqa: BLE001
        pass
    return out, (f"live: rocprofv3 --kernel-trace around `bench.py {' '.join(child)}` spawned by this run ({time.perf_counter() - t0:.0f} s): "
                 "average over all dispatches of a kernel, as rocprofv3 --stats reports it")


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


def make_press(kind, ratio, P=None):
    """The press of a workload from package `P` (kvpress_amd by default; the reference package for the CPU baseline: same
    class names, same constructor arguments)."""
    if P is None:
        import kvpress_amd as P

    if kind == "snapkv":
        return P.SnapKVPress(compression_ratio=ratio, window_size=WINDOW, kernel_size=5)
    if kind == "knorm":
        return P.KnormPress(compression_ratio=ratio)
    if kind == "keydiff":
        return P.KeyDiffPress(compression_ratio=ratio)
    if kind == "cur":
        return P.CURPress(compression_ratio=ratio)
    if kind == "finch":
        press = P.FinchPress(compression_ratio=ratio)
        press.window_size = WINDOW   # (set by the embedding hook in a model run: the question's length)
        return press
    if kind == "chunk_snapkv":
        return P.ChunkPress(press=P.SnapKVPress(compression_ratio=ratio, window_size=WINDOW, kernel_size=5), chunk_length=1024)
    if kind == "rerotate":
        return P.KeyRerotationPress(press=P.KnormPress(compression_ratio=ratio))
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


def event_timed_steps(step, n: int = 20, warmup: int = 3) -> dict:
    """SURVEY §8(d)'s timing method beside the contract's host-clock mean: `warmup` untimed steps, then `n` steps each bracketed by a
    HIP event pair on the launch stream (torch's current stream = the stream the library launches on); median and min in ms.  An
    event-bracketed step carries the event floor (~6 us) and starts on a drained queue, so its median sits a little above
    `ms_per_step` (back-to-back steps overlap their launch ramps); the min / median pair shows how much of a box-to-box spread is
    clock ramp rather than the kernels."""
    import torch

    for _ in range(warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    med = t[n // 2] if n % 2 else 0.5 * (t[n // 2 - 1] + t[n // 2])
    return {"method": "hipEvent pair per step on the launch stream", "warmup": warmup, "n": n, "median_ms": round(med, 4), "min_ms": round(t[0], 4),
            "max_ms": round(t[-1], 4)}


def roofline_block(avg: dict, workload: str, B: int, t_step: float, world: int, live_pmc: str, traced=None, traced_source=None):
    """The `roofline` object of the bench line.  avg = {kernel: (mean ms, launches per step)} is the library's HIP-event table
    (kvp_prof_*); `traced` = {kernel: (average us, dispatches)} the same kernels as rocprofv3 --kernel-trace measured them in this run
    (live_kernel_trace) -- when it is there, EVERY per-kernel number of the block (avg_launch_us, achieved, frac, p1 / p2 fractions,
    path.kernels_us) comes from it, i.e. from the data profiles/<round>_rocprofv3_kernel_stats_*.csv summarises, and the event table is
    kept as path.kernels_us_events (events read 3-6 us more per kernel: VERDICT r5 weak #10).  Needs no GPU: for N > 1 (and under a
    profiler) `traffic` falls back to the committed PMC summary of this build (pmc_traffic).  None when no kernel moves algorithmic bytes."""
    kind, S, ratio = WORKLOADS[workload]
    # kernel -> (ms per launch, launches per step) from the better source
    traced_step_ms = None
    if traced:
        traced_step_ms = traced.get("__step_ms__", (None, 0))[0]
        tim = {k: ((traced[k][0] * 1e-3, c) if k in traced else (a, c)) for k, (a, c) in avg.items()}
        timing_source = traced_source
    else:
        tim = dict(avg)
        timing_source = f"HIP events around every launch (kvp_prof_*: ~3-6 us above rocprofv3 per kernel) [{traced_source or 'no kernel-trace pass requested'}]"
    cand = {k: a for k, (a, _) in tim.items() if kernel_bytes(k, kind, S, ratio) > 0}
    if not cand:
        return None
    ab = algorithmic_bytes(kind, S, ratio)
    dom = max(cand, key=cand.get)
    kb = kernel_bytes(dom, kind, S, ratio) * B
    ach = kb / (cand[dom] * 1e-3) / 1e9
    model = path_model(tim, kind, S, ratio, B)
    model["frac_of_step"] = round(model["total_us"] * 1e-6 / t_step, 4)   # (a self-made floor: kept inside path.model)
    path_frac = ab["total"] * B / t_step / 1e9 / HBM_PEAK_GBS

    def kfrac(prefix):   # HBM fraction of one kernel family (None if the workload does not launch it)
        ks = [k for k in tim if k.startswith(prefix)]
        return round(kernel_bytes(ks[0], kind, S, ratio) * B / (tim[ks[0]][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ks else None

    traffic, traffic_source = (None, "not requested" if live_pmc == "off" else f"no live pass with {world} ranks")
    if world == 1 and live_pmc != "off":
        try:
            traffic, traffic_source = live_pmc_traffic(dom, workload)
        except Exception as e:   # noqa: BLE001 -- the bench line must never die in its optional profiler pass
            traffic, traffic_source = None, f"live PMC pass raised {type(e).__name__}: {e}"
    if traffic is None:   # no profiler here (or a child / profiled run / N > 1): the committed summary of this same build, if there is one
        live_note = traffic_source
        traffic, traffic_source = pmc_traffic(dom, workload)
        traffic_source = f"{traffic_source} [{live_note}]"
    ksum = sum(a * c for a, c in tim.values()) * 1e3
    roofline = {
        # THE number the north-star target of 0.70 is about comes first: the whole compress() against SURVEY §8(d)'s bytes
        "path_frac": round(path_frac, 4),
        "p1_frac": kfrac("snapkv_p1"), "p2_frac": kfrac("snapkv_p2"),
        # the dominant kernel (contract fields)
        "kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": kb, "avg_launch_us": round(cand[dom] * 1e3, 2), "timing_source": timing_source,
        # secondary bound of the same kernel: the window-attention passes are matrix-core / VALU work
        "mfma": ({"achieved": round(kernel_flops(dom, S) * B / (cand[dom] * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": round(kernel_flops(dom, S) * B / (cand[dom] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
                 if kernel_flops(dom, S) else None),
        "path": {
            "algorithmic_bytes_per_layer": ab["total"] * B,
            "achieved": round(ab["total"] * B / t_step / 1e9, 1),
            "frac": round(path_frac, 4),
            "kernels_us": {k: round(a * 1e3 * c, 2) for k, (a, c) in sorted(tim.items())},
            "kernels_sum_us": round(ksum, 2),
            # back-to-back kernels overlap their launch ramps (~1.5 us per boundary), so the sum exceeds the step of the SAME run by a
            # little -- never by more than 4 % when the durations are the profiler's (compared with the traced run's own step time
            # where it is known: that is the run the durations belong to)
            "traced_ms_per_step": traced_step_ms,
            "kernels_sum_over_step": round(ksum / ((traced_step_ms or t_step * 1e3) * 1e3), 4),
            "kernels_sum_le_1p04_step": bool(ksum <= 1.04 * (traced_step_ms or t_step * 1e3) * 1e3),
            "kernels_us_events": {k: round(a * 1e3 * c, 2) for k, (a, c) in sorted(avg.items())},
            "model": model,
        },
    }
    m = roofline["mfma"]
    if m and m["frac"] > roofline["frac"]:   # the matrix-core roof is the nearer one (ExpectedAttention's quadratic form)
        roofline["hbm"] = {k: roofline[k] for k in ("achieved", "peak", "unit", "frac")}
        roofline.update(bound="mfma", achieved=m["achieved"], peak=m["peak"], unit=m["unit"], frac=m["frac"])
    return roofline


def bench_config(workload: str, world: int, B: int, n_kept: int, prewarm_ms: float, inputs: str, kept_order: str, library_qproj: bool) -> dict:
    """`config` of the bench line: the same keys whatever the number of ranks (tests/test_dist_gloo.py pins that)."""
    kind, S, ratio = WORKLOADS[workload]
    return {"workload": workload, "press": kind, "compression_ratio": ratio, "batch_per_gpu": B, "seq_len": S, "n_kept": n_kept,
            "h_q": H_Q, "h_kv": H_KV, "head_dim": D, "layers_for_tok_s": LAYERS, "parallelism": f"batch-sharded x{world}, no collective",
            "prewarm_ms": prewarm_ms, "inputs": inputs, "kept_order": kept_order, "library_qproj": library_qproj}


def result_line(args, world: int, B: int, S: int, t_step: float, config: dict, roofline, cpu, metric: str, dtype: str = "bf16") -> dict:
    assert world == args.gpus
    return {"metric": metric, "value": round(world * B * S / (LAYERS * t_step), 1), "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic", "config": config, "roofline": roofline, "cpu_baseline": cpu}


def reference_package():
    """NVIDIA/kvpress itself: oracle/_ref/kvpress, the build-time copy oracle/build_ref.py makes where /root/reference exists (it
    travels to the GPU box with the tree but is never committed), with stand-ins for its two missing dependencies.  None if absent."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isfile(os.path.join(ref, "kvpress", "__init__.py")):
        return None
    if ref not in sys.path:
        sys.path.insert(0, ref)
    try:
        import kvpress

        return kvpress
    except Exception as e:  # a broken copy must not take the bench line down: fall back to the restatement
        print(f"[bench] oracle/_ref/kvpress not importable ({e!r}): using oracle/torch_path.py", file=sys.stderr)
        return None


def cpu_baseline(workload, press, att, rot, hidden, keys, values, kwargs, n_kept):
    """The reference's pure-PyTorch path on the host CPU cores of THIS box, same run, same tensors (north_star).
    kind "reference": the reference's own ``press.compress()`` (scorer_press.py:76-102 + the scorer; wrappers likewise) from
    oracle/_ref; its output is compared bit for bit with oracle/torch_path.py -- the same op sequence restated line by line and
    pinned to the real reference by tests/test_oracle_golden.py -- which is what is timed instead (kind "port") where the copy
    of the reference is absent.  bf16 as users run it and float32 ("O32"); 1 warm-up + timed runs, median; threads =
    torch.get_num_threads().  ExpectedAttention's 4.4-TFLOP q_proj makes one run ~10 s: one timed run per dtype there.  The
    numpy float32 port (oracle/kvpress_oracle.py) is timed once as a secondary figure, and the GPU path's retained set is
    checked against the float32 CPU scores while they are at hand."""
    import numpy as np
    import torch

    from kvpress_amd import _native
    from oracle import kvpress_oracle as O
    from oracle import torch_path as TP

    kind, S, ratio = WORKLOADS[workload]
    ref = reference_package()
    restated = kind in TP.SCORERS
    if ref is None and not restated:
        return None   # an f-row workload without the reference at hand: nothing honest to time
    threads = torch.get_num_threads()
    n_timed = 1 if kind in ("ea", "chunk_snapkv") else 3
    att_cpu = {torch.bfloat16: build_module(torch.device("cpu"))[0]}
    att_cpu[torch.float32] = build_module(torch.device("cpu"))[0].float()
    rot_cpu = build_module(torch.device("cpu"))[1]
    res, sc32, identical = {}, None, None
    with torch.no_grad():
        for dt in (torch.bfloat16, torch.float32):
            m = att_cpu[dt]
            m.rotary_emb = rot_cpu
            h, k, v = hidden.cpu().to(dt), keys.cpu().to(dt), values.cpu().to(dt)
            kw = {"position_embeddings": tuple(t.cpu().to(dt) for t in kwargs["position_embeddings"])}
            ref_press = make_press(kind, ratio, ref) if ref is not None else None
            times = []
            for i in range(1 + n_timed):
                t0 = time.perf_counter()
                if ref_press is not None:
                    ko, vo = ref_press.compress(m, h, k, v, None, kw)
                else:
                    ko, vo, _ = TP.torch_compress(TP.SCORERS[kind], ratio, m, h, k, v, kw)
                if i:
                    times.append(time.perf_counter() - t0)
            assert tuple(ko.shape) == (k.shape[0], H_KV, n_kept, D), (tuple(ko.shape), n_kept)
            res[dt] = sorted(times)[len(times) // 2]
            if ref_press is not None and restated and dt == torch.bfloat16 and kind != "ea":   # the restatement IS the reference: same bytes
                k2, v2, _ = TP.torch_compress(TP.SCORERS[kind], ratio, m, h, k, v, kw)
                identical = bool(torch.equal(k2, ko) and torch.equal(v2, vo))
            if dt == torch.float32 and restated:
                sc32 = TP.SCORERS[kind](m, h, k, v, kw).numpy()
            del h, k, v, ko, vo
    # SURVEY section 8(d)'s third column: the SAME reference code (pure PyTorch-ROCm, bf16) on this GPU -- what a user of NVIDIA/kvpress gets on an
    # MI355X without this library.  1 warm-up + 3 runs, median, device-synchronised wall clock.
    ref_gpu_ms = None
    if ref is not None and hidden.is_cuda:
        try:
            with torch.no_grad():
                ref_press = make_press(kind, ratio, ref)
                if getattr(att, "rotary_emb", None) is None:
                    att.rotary_emb = rot
                tg = []
                for i in range(4):
                    torch.cuda.synchronize(hidden.device)
                    t0 = time.perf_counter()
                    ko, vo = ref_press.compress(att, hidden, keys, values, None, kwargs)
                    torch.cuda.synchronize(hidden.device)
                    if i:
                        tg.append(time.perf_counter() - t0)
                    assert tuple(ko.shape) == (keys.shape[0], H_KV, n_kept, D)
                    del ko, vo
                ref_gpu_ms = round(sorted(tg)[len(tg) // 2] * 1e3, 3)
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 -- context only: never take the line down
            print(f"[bench] the reference on the GPU failed ({e!r}): field omitted", file=sys.stderr)
    ok = None
    if sc32 is not None:   # GPU retained set vs the float32 CPU scores (tie-tolerant, 1e-3 band)
        gsc = press.score(att, hidden, keys, values, None, kwargs)
        gidx = _native.topk_select(gsc, n_kept).cpu().numpy()
        ok = bool(O.topk_is_valid(sc32, gidx, n_kept, rel_band=1e-3)[0])
    # secondary: the numpy float32 port of the same algorithm (the round-1 baseline)
    port_ms = None
    if kind in ("snapkv", "knorm") and S > 4096:
        k_np, v_np = keys.float().cpu().numpy(), values.float().cpu().numpy()
        if kind == "snapkv":
            with torch.no_grad():
                q_np = press.compute_window_queries(att, hidden, WINDOW, kwargs["position_embeddings"]).float().cpu().numpy()
        t0 = time.perf_counter()
        sc = O.snapkv_score(q_np, k_np, 5, ctype=np.float32) if kind == "snapkv" else O.knorm_score(k_np, ctype=np.float32)
        O.compress(sc, k_np, v_np, ratio)
        port_ms = round((time.perf_counter() - t0) * 1e3, 1)
    t = res[torch.bfloat16]
    what = ("NVIDIA/kvpress's own press.compress() (oracle/_ref/kvpress, copied from the reference tree at build time)" if ref is not None
            else "score + topk + gather of the reference restated op for op in plain PyTorch (oracle/torch_path.py)")
    return {"value": round(S / (LAYERS * t), 1), "unit": "tok/s", "cores": threads, "kind": "reference" if ref is not None else "port",
            "sample": f"one full {workload} layer (B=1, H_kv={H_KV}, S={S}) per run: {what}, bf16 module and tensors, 1 warm-up + "
                      f"{n_timed} timed run(s), median; torch.get_num_threads() = {threads}, os.cpu_count() = {os.cpu_count()}; "
                      f"tok/s extrapolates one layer x {LAYERS}",
            "ms_per_layer": round(t * 1e3, 1), "ms_per_layer_fp32": round(res[torch.float32] * 1e3, 1),
            "restatement_bit_identical_to_reference": identical,
            "reference_on_this_gpu_ms_per_layer": ref_gpu_ms,
            "numpy_port_ms_per_layer": port_ms, "gpu_topk_valid_vs_cpu_fp32_scores": ok}


def stub_main(args, world: int, rank: int):
    """(tests) The N-rank path of this file -- launcher, rank environment, batch sharding, barrier-bracketed timing, MAX over ranks,
    one JSON line from rank 0 -- with the press replaced by a host sleep; runs under gloo on CPU."""
    kind, S, ratio = WORKLOADS[args.workload]
    lo, hi = shard_batch(world, world, rank)
    B = hi - lo
    calls = [0]

    def step():
        calls[0] += 1
        if args.stub_fail_rank == rank and calls[0] > args.warmup + 1:
            raise RuntimeError(f"stub failure on rank {rank} (tests: a rank that dies mid-run must take the job down, not hang it)")
        time.sleep(args.stub_step * 1e-3 * (1 + rank))

    local = timed_steps(step, args.steps, args.warmup, world, lambda: None)
    t_step = aggregate_time(local, world) / args.steps
    if rank == 0:
        cfg = bench_config(args.workload, world, B, n_kept_of(kind, S, ratio), args.prewarm_ms, "none (stub)", "position", False)
        cfg["stub_step_ms"] = args.stub_step
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
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the `extra` block of the default run (Knorm 32k and ExpectedAttention 128k measured after the headline, ~25 s)")
    ap.add_argument("--live-pmc", default="auto", choices=["auto", "off"],
                    help="roofline.traffic: auto = measure it now with two rocprofv3 --pmc passes around a 3-step child run (N = 1, ~40 s); "
                         "off = quote the committed summary of this build (profiles/<round>_pmc_summary_<workload>.txt)")
    ap.add_argument("--profile-json", default=None, help="also dump the per-kernel HIP-event table here")
    ap.add_argument("--trace-stats-csv", default=None,
                    help="write the kernel statistics of the live rocprofv3 --kernel-trace pass (the data behind the line's per-kernel numbers) "
                         "here, in rocprofv3 --stats' CSV format: profiles/<round>_rocprofv3_kernel_stats_<workload>.csv")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL); gloo + --stub-step exercises the launcher on CPU (tests)")
    ap.add_argument("--stub-step", type=float, default=None, metavar="MS",
                    help="(tests) replace the press by a host sleep of MS milliseconds: launcher / sharding / timing path without a GPU")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="(tests, with --stub-step) this rank raises in the timed region")
    ap.add_argument("--dist-timeout", type=float, default=600.0,
                    help="seconds after which a collective (the barriers around the timed region, the MAX-reduce of the step time) gives up: "
                         "a rank that died must end the job, not leave the others waiting at a barrier")
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
    device = None
    if not stub:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
        # KVP_BENCH_SHARE_GPU=1 (tests only, gloo backend): every rank uses GPU 0, so that the N-rank path -- real kernels, real
        # sharding, barrier-bracketed timing, MAX over ranks -- can run on a one-GPU box.  Never set it for a measurement.
        share = os.environ.get("KVP_BENCH_SHARE_GPU") == "1"
        assert not (share and args.backend == "nccl"), "KVP_BENCH_SHARE_GPU needs --backend gloo (RCCL refuses two ranks on one GPU)"
        if share:
            # Two PROCESSES on one GPU break the one-launch select's contract (Concurrency: one cluster select in flight per device):
            # launched at the same moment, each can hold CUs the other's rows wait for -- a circular wait that ends in the bounded time-out
            # and a loud KVP_EASYNC (seen once in round 6: profiles/r06_gpu_tests_shared_gpu_timeout.txt).  The shared-GPU test mode therefore
            # runs the multi-launch select, as a deployment that shares a GPU between processes should.
            os.environ["KVP_TK_CLUSTER"] = "0"
        dev_index = 0 if share else local_rank
        assert torch.cuda.device_count() > dev_index, f"rank {rank}: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible on this node"
        device = torch.device("cuda", dev_index)
        torch.cuda.set_device(device)   # before the process group: the RCCL communicator binds this rank's GPU (device_id) at init,
                                        # so the first barrier does not have to guess a device
    if world > 1:
        import torch.distributed as dist

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl" and not stub:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=tmo)
    try:
        return run(args, world, rank, device, stub)
    except BaseException as e:   # noqa: BLE001 -- any rank's failure (a shape assert, KVP_EASYNC, an OOM ...) ends the WHOLE job
        if world > 1 and not isinstance(e, SystemExit):
            import traceback

            traceback.print_exc()
            print(f"bench.py: rank {rank} failed ({type(e).__name__}); aborting the {world}-rank job", file=sys.stderr, flush=True)
            # no destroy_process_group(): it may wait for peers that sit in a barrier.  A non-zero exit makes the launcher
            # (torch.distributed.run) terminate the other ranks; --dist-timeout bounds their wait if it does not.
            os._exit(3)
        raise


def run(args, world: int, rank: int, device, stub: bool):
    """Everything after the process group exists (split from main() so that main() can turn any rank's failure into the job's)."""
    if stub:
        return stub_main(args, world, rank)

    from kvpress_amd import _native

    _native.lib()  # fail loudly if the HIP extension is missing
    bpg = BATCH.get(args.workload, 1)
    lo, hi = shard_batch(world * bpg, world, rank)  # global batch = BATCH[workload] (default one) elements per GPU (weak scaling)
    head = measure(args.workload, args.steps, args.warmup, args.prewarm_ms, world, rank, lo, hi, device, args.live_pmc, args.profile_json,
                   cpu=(rank == 0 and world == 1 and not args.no_cpu_baseline), parity=(rank == 0 and world == 1),   # (N > 1: timing only)
                   trace_csv=args.trace_stats_csv)
    # ---- BASELINE.json's other single-GPU configurations, measured in the SAME run so that the driver's record carries them
    # (VERDICT r3 #7): config 2 (Knorm 32k) and config 4 (ExpectedAttention 128k).  After the headline's timed region; N = 1 only.
    extra = None
    if (rank == 0 and world == 1 and args.workload == "snapkv128k" and not args.no_extra and os.environ.get("KVP_BENCH_CHILD") != "1"
            and not under_profiler()):
        extra = {}
        for wl, (st, wu) in (("knorm32k", (200, 20)), ("ea128k", (10, 2))):
            try:
                r = measure(wl, st, wu, 30.0, 1, 0, 0, 1, device, "off", None, cpu=False, parity=(wl in FIXTURES))
                rf = r["roofline"] or {}
                path = rf.get("path", {})
                ab = algorithmic_bytes(*WORKLOADS[wl])
                extra[wl] = {"ms_per_step": round(r["t_step"] * 1e3, 4), "steps": st, "warmup": wu, "path_frac": rf.get("path_frac"),
                             "kernels_sum_us": path.get("kernels_sum_us"),
                             "kernel_only_frac": (round(ab["total"] / (path["kernels_sum_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                                  if path.get("kernels_sum_us") else None),
                             "path_model": {k: path.get("model", {}).get(k) for k in ("total_us", "frac_of_step")},
                             "dominant_kernel": rf.get("kernel"), "dominant_frac": rf.get("frac"), "bound": rf.get("bound"),
                             "kernels_us": path.get("kernels_us"), "parity": r["parity"], "step_events": r["step_events"]}
            except Exception as e:   # noqa: BLE001 -- the headline line must never die in an extra
                extra[wl] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        kind, S, ratio = WORKLOADS[args.workload]
        metric = ("press ms/layer + prefill tok/s, Llama-3.1-8B 128k ctx, SnapKV ratio=0.5" if args.workload == "snapkv128k"
                  else f"press ms/layer + prefill tok/s, Llama-3.1-8B, {args.workload}")
        cfg = bench_config(args.workload, world, head["B"], head["n_kept"], args.prewarm_ms, head["inputs"], head["kept_order"], bool(_native.USE_LIBRARY_QPROJ))
        line = result_line(args, world, head["B"], S, head["t_step"], cfg, head["roofline"], head["cpu"], metric)
        line["parity"] = head["parity"]
        line["step_events"] = head["step_events"]   # hipEvent median / min per step; ms_per_step above stays the host-clock mean
        if extra is not None:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def bench_inputs(workload: str, b: int, device):
    """The timed tensors of batch element `b` (global index): K, V [1,H_kv,S,D] and hidden [1,S,4096] bf16, generated on the CPU
    (SURVEY §8d set A: flat N(0,1), CPU torch.Generator, rounded to bf16 once -- tests/_fullsize.py, the full-size parity inputs) and
    moved to the device.  Element 0 of the headline / of config 2 is exactly the tensor set of tests/golden/full_snapkv128k.npz /
    full_knorm32k.npz.  hidden: the presses that project a window read its last 64 rows only (the rest is zero, as in the fixture);
    ExpectedAttention, FINCH and per-chunk SnapKV read all of it; the others none."""
    import _fullsize as F

    kind, S, ratio = WORKLOADS[workload]
    spec = dict(kind={"snapkv": "snapkv", "ea": "ea", "finch": "ea", "chunk_snapkv": "ea"}.get(kind, "none"), S=S, ratio=ratio, data="A",
                seed=SEEDS.get(workload, 103) + 7919 * b)
    k, v = F.make_kv(spec)
    h = F.make_hidden(spec)
    return k.to(device), v.to(device), h.to(device), f"CPU-seeded set A (tests/_fullsize.py), seed {spec['seed']}"


def fixture_parity(workload, press, att, hidden, keys, values, kwargs, n_kept):
    """The timed tensors against the REAL reference's committed outputs for exactly these tensors (tests/golden/full_*.npz, made by
    oracle/gen_golden_fullsize.py): scores within 1e-3, retained set identical outside the tolerance band (tests/_fullsize.py
    check_against_reference -- the rule of tests/test_gpu_fullsize.py).  None where no fixture of this workload's tensors exists."""
    import numpy as np

    import _fullsize as F
    from kvpress_amd import _native

    name = FIXTURES.get(workload)
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz") if name else None
    if not path or not os.path.exists(path) or keys.shape[0] != 1:
        return None
    fx = np.load(path)
    sc = press.score(att, hidden, keys, values, None, kwargs)
    idx = _native.topk_select(sc, n_kept)
    try:
        worst, differ = F.check_against_reference(fx, sc, idx)
        return {"fixture": f"tests/golden/{name}.npz (outputs of the real reference on these tensors)", "ok": True,
                "max_rel_err_scores": float(f"{worst:.3e}"), "set_differences_inside_band": int(differ)}
    except AssertionError as e:
        return {"fixture": f"tests/golden/{name}.npz", "ok": False, "error": str(e)[:300]}


def measure(workload, steps, warmup, prewarm_ms, world, rank, lo, hi, device, live_pmc, profile_json, cpu: bool, parity: bool, trace_csv=None) -> dict:
    """One workload on this rank's shard: inputs, pre-warm, W warm-up + K timed steps (barrier + sync bracketed, MAX over ranks),
    then -- outside the timed region -- per-kernel HIP-event timing, the roofline block, the fixture parity check and the CPU baseline."""
    import torch

    from kvpress_amd import _native

    kind, S, ratio = WORKLOADS[workload]
    B = hi - lo
    parts = [bench_inputs(workload, b, device) for b in range(lo, hi)]
    keys, values, hidden = (torch.cat([p[i] for p in parts]) if B > 1 else parts[0][i] for i in range(3))
    inputs_note = parts[0][3]
    del parts
    att, rot = build_module(device)
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=device)[None])
    kwargs = {"position_embeddings": pe}
    press = make_press(kind, ratio)
    if workload.endswith("_scoreorder"):
        press.kept_order = "score"

    def step():
        with torch.no_grad():
            return press.compress(att, hidden, keys, values, None, kwargs)

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:  # untimed, back to back: lets the clock governor settle
        for _ in range(25):
            out = step()
        torch.cuda.synchronize()
    local = timed_steps(step, steps, warmup, world, torch.cuda.synchronize)
    out = step()
    torch.cuda.synchronize()
    _native.async_error_check()   # a select kernel that gave up during the timed region would have poisoned its result: fail, loudly
    ev_stats = event_timed_steps(step) if rank == 0 else None   # median / min of event-bracketed steps, beside the mean
    total = aggregate_time(local, world)
    t_step = total / steps
    n_kept = n_kept_of(kind, S, ratio)
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
        traced, traced_source = None, ("not requested" if live_pmc == "off" else f"no live pass with {world} ranks")
        if world == 1 and live_pmc != "off":
            try:
                traced, traced_source = live_kernel_trace(list(avg), workload, csv_path=trace_csv)
            except Exception as e:   # noqa: BLE001 -- the bench line must never die in its optional profiler pass
                traced, traced_source = None, f"live kernel-trace pass raised {type(e).__name__}: {e}"
        roofline = roofline_block(avg, workload, B, t_step, world, live_pmc, traced, traced_source)
        if profile_json:
            with open(profile_json, "w") as f:
                json.dump({"workload": workload, "ms_per_step": t_step * 1e3,
                           "kernels_avg_ms": {k: a for k, (a, _) in avg.items()},
                           "launches_per_step": {k: c for k, (_, c) in avg.items()},
                           "kernels_avg_us_rocprofv3": {k: v[0] for k, v in (traced or {}).items() if not k.startswith("__")}}, f, indent=1)

    par = None
    if parity and rank == 0:
        with torch.no_grad():
            par = fixture_parity(workload, press, att, hidden, keys, values, kwargs, n_kept)
    # ---- CPU baseline (rank 0, N=1 only): the reference's op sequence in plain PyTorch on this box's host cores -------------
    cpu_res = cpu_baseline(workload, press, att, rot, hidden, keys, values, kwargs, n_kept) if cpu else None
    res = {"t_step": t_step, "B": B, "n_kept": n_kept, "roofline": roofline, "cpu": cpu_res, "parity": par, "inputs": inputs_note,
           "kept_order": getattr(press, "kept_order", "position"), "step_events": ev_stats}
    del keys, values, hidden, out
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
