#!/usr/bin/env python3
"""Per-kernel clock / power evidence for the SnapKV path ("prove or retire the power-wall claim").

One box, one run, three kinds of evidence side by side:
  (1) sustained loops (>= 1 s each) of: the window-attention score call (p1 + combine + p2 + pool), the gather, the whole
      compress step, a K-norm stream, and the stage micro-benchmark's variants (mfma_only / valu_only / math / math_lds_bar /
      math_lds_bar_dma on 256 workgroups with random operands, mfma_only on constant operands), while tools/smi_sampler.py logs
      socket power, per-XCD gfx clocks and throttle status through amdsmi at 50 Hz;
  (2) IN-KERNEL effective clocks of snapkv_p1_asm / snapkv_p2_asm / gather_vec_kernel (lab build tools/make_clock_lab.py:
      s_memtime / s_memrealtime stamps per workgroup, per XCD) inside the same loops;
  (3) the ubench's own in-kernel clock per variant.
PMC counters (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU) come from scripts/gpu_check.sh pmc.

    python tools/power_clock_lab.py > profiles/r05_clock_power.txt
"""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LOOP_S = float(os.environ.get("PCL_LOOP_S", "1.5"))
SAMPLES = "/tmp/pcl_samples.jsonl"


def child():
    """runs in a subprocess (so that KVPRESS_HIP_LIB selects the library): loops + stamp read-out; prints JSON lines"""
    import numpy as np
    import torch

    import bench
    from kvpress_amd import _native
    from kvpress_amd.utils import get_prerope_query_states

    dev = torch.device("cuda", 0)
    kind, S, ratio = bench.WORKLOADS["snapkv128k"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    bf = torch.bfloat16
    keys = torch.randn((1, 8, S, 128), generator=gen, device=dev, dtype=torch.float32).to(bf)
    values = torch.randn((1, 8, S, 128), generator=gen, device=dev, dtype=torch.float32).to(bf)
    hidden = torch.randn((1, S, 4096), generator=gen, device=dev, dtype=bf)
    att, rot = bench.build_module(dev)
    with torch.no_grad():
        pe = rot(hidden, torch.arange(S, device=dev)[None])
        q_pre = get_prerope_query_states(att, hidden[:, -64:])
    cos, sin = pe[0][:, -64:], pe[1][:, -64:]
    press = bench.make_press(kind, ratio)
    kw = {"position_embeddings": pe}
    lab = os.environ.get("KVPRESS_HIP_LIB", "").endswith("clocklab.so")
    L = _native.lib()
    MAXWG = 16384

    def read_stamps(fn_name, kind_):
        fn = getattr(L, fn_name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        tab = np.zeros((MAXWG, 4), dtype=np.uint64)
        xcc = np.zeros(MAXWG, dtype=np.uint32)
        torch.cuda.synchronize()
        rc = fn(kind_, tab.ctypes.data, xcc.ctypes.data)
        assert rc == 0, (fn_name, rc)
        ok = (tab[:, 2] > tab[:, 0]) & (tab[:, 3] > tab[:, 1])
        cyc = (tab[ok, 2] - tab[ok, 0]).astype(np.float64)
        rt = (tab[ok, 3] - tab[ok, 1]).astype(np.float64)
        x = xcc[ok]
        out = {"workgroups": int(ok.sum()), "mhz_mean": float((cyc.sum() / rt.sum()) * 100.0), "wg_us_mean": float(rt.mean() / 100.0), "cycles_mean": float(cyc.mean())}
        out["mhz_per_xcc"] = {int(i): round(float(cyc[x == i].sum() / rt[x == i].sum() * 100.0), 1) for i in sorted(set(x.tolist()))}
        span = (tab[ok, 3].max() - tab[ok, 1].min()) / 100.0
        out["launch_span_us"] = float(span)
        return out

    def loop(name, fn, stamp_readers=()):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        time.sleep(0.4)                       # idle gap: the samples show the transition
        t0 = time.time()
        n = 0
        while time.time() - t0 < LOOP_S:
            for _ in range(20):
                fn()
            n += 20
            if n % 200 == 0:
                torch.cuda.synchronize()      # bounded queue depth
        torch.cuda.synchronize()
        t1 = time.time()
        rec = {"phase": name, "t0": t0, "t1": t1, "iters": n, "us_per_iter": (t1 - t0) / n * 1e6, "lib": "clocklab" if lab else "production"}
        # clock probe right behind the loop (kvp_clock_probe: one wave, 20 us)
        try:
            rec["probe_after_mhz"] = float(_native.clock_probe(dev, 20).item())
        except Exception as e:  # noqa: BLE001
            rec["probe_error"] = repr(e)
        if lab:
            # stamps of the LAST launch of each kernel (tables are cleared by the reader): run one more iteration for a clean read
            for label, (fn_name, kind_) in stamp_readers:
                rec[label] = read_stamps(fn_name, kind_)
        print(json.dumps(rec), flush=True)

    with torch.no_grad():
        sc = _native.snapkv_score_rope(q_pre, cos, sin, keys, 5)
        idx = _native.topk_select(sc, S // 2)
        sk = (("p1_asm", ("kvp_lab_stamps_snapkv", 0)), ("p2_asm", ("kvp_lab_stamps_snapkv", 1)))
        ga = (("gather", ("kvp_lab_stamps_gather", 0)),)
        if lab:   # clear the tables
            for _, (fn_name, kind_) in sk + ga:
                try:
                    read_stamps(fn_name, kind_)
                except Exception:  # noqa: BLE001
                    pass
        loop("score_call(p1+combine+p2+pool)", lambda: _native.snapkv_score_rope(q_pre, cos, sin, keys, 5), sk)
        loop("gather", lambda: _native.gather_kv(keys, values, idx), ga)
        loop("compress_step", lambda: press.compress(att, hidden, keys, values, None, kw), sk + ga)
        loop("knorm_stream", lambda: _native.rownorm_score(keys, -1.0))
        if not lab and os.environ.get("PCL_EA", "1") != "0":
            # ExpectedAttention's kernels (is the quadratic form power-limited like the SnapKV passes?)
            qg = torch.randn((1, S, 32 * 128), generator=gen, device=dev, dtype=torch.float32).to(bf).view(1, S, 32, 128).transpose(1, 2)
            mu, cov = _native.ea_qstats(qg, True)
            loop("ea_qstats(1 GiB of Q)", lambda: _native.ea_qstats(qg, True))
            loop("ea_score(logits + |v| + finalize)", lambda: _native.ea_score(keys, values, mu, cov, 4, True, 0.0))
            loop("knorm_compress(128k)", lambda: _native.knorm_compress(keys, values, S // 2))


def ubench(variant, mode, loop_ms):
    exe = os.path.join(ROOT, "tools", "ubench_stage")
    r = subprocess.run([exe, variant, str(loop_ms), str(mode)], capture_output=True, text=True, timeout=120)
    out = {}
    for line in r.stdout.splitlines():
        if line.startswith("LOOP "):
            f = line.split()
            out.update({"phase": f"ubench:{variant}:{'256wg_random' if mode else '8wg_const'}", "t0": float(f[5]), "t1": float(f[7]), "launches": int(f[9]),
                        "ns_per_stage_sustained": float(f[11]), "in_kernel_mhz": float(f[13])})
        elif line.startswith(variant):
            out["one_shot"] = line.strip()
    if not out:
        out = {"phase": f"ubench:{variant}", "error": (r.stdout + r.stderr)[-400:]}
    return out


def join(rec, samples):
    w = [s for s in samples if rec["t0"] + 0.25 <= s["t"] <= rec["t1"] - 0.05 and s.get("power_w") is not None]
    if not w:
        return {}
    p = [s["power_w"] for s in w]
    c = [s["gfxclk_mhz"] for s in w if s.get("gfxclk_mhz")]
    out = {"samples": len(w), "power_w_mean": round(sum(p) / len(p), 1), "power_w_max": round(max(p), 1)}
    if c:
        out["smi_gfxclk_mhz_mean"] = round(sum(c) / len(c), 1)
        out["smi_gfxclk_mhz_min"] = round(min(c), 1)
    per = [s["gfxclks_mhz"] for s in w if s.get("gfxclks_mhz")]
    if per:
        n = min(len(x) for x in per)
        out["smi_gfxclk_per_xcd"] = [round(sum(x[i] for x in per) / len(per)) for i in range(n)]
    for k in ("uclk_mhz", "gfx_activity", "umc_activity", "temp_hotspot"):
        v = [s[k] for s in w if s.get(k) is not None]
        if v:
            out[k] = round(sum(v) / len(v), 1)
    thr = [s.get("throttle_status") for s in w if s.get("throttle_status") not in (None, 0, "N/A")]
    if thr:
        out["throttle_status_seen"] = sorted({json.dumps(t) for t in thr})[:4]
    it = [s.get("indep_throttle") for s in w if s.get("indep_throttle") not in (None, 0, "N/A")]
    if it:
        out["indep_throttle_seen"] = sorted({json.dumps(t) for t in it})[:4]
    return out


def main():
    if "--child" in sys.argv:
        return child()
    sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "smi_sampler.py"), SAMPLES, "0.02"])
    time.sleep(1.0)
    recs = []
    try:
        idle0 = time.time()
        time.sleep(1.0)
        recs.append({"phase": "idle", "t0": idle0 - 0.25, "t1": time.time() + 0.05})
        for libname in ("production", "clocklab"):
            env = dict(os.environ)
            if libname == "clocklab":
                env["KVPRESS_HIP_LIB"] = os.path.join(ROOT, "kvpress_amd", "lib", "variants", "clocklab.so")
                if not os.path.exists(env["KVPRESS_HIP_LIB"]):
                    print("# clocklab.so missing: run tools/make_clock_lab.py", flush=True)
                    continue
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=600)
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    recs.append(json.loads(line))
            if r.returncode:
                print(f"# child ({libname}) rc={r.returncode}: {r.stderr[-600:]}", flush=True)
        loop_ms = int(LOOP_S * 1000)
        for variant, mode in (("mfma_only", 1), ("valu_only", 1), ("math", 1), ("math_lds_bar", 1), ("math_lds_bar_dma", 1), ("mfma_lds", 1),
                              ("mfma_only", 0), ("math_lds_bar_dma", 0)):
            time.sleep(0.4)
            try:
                recs.append(ubench(variant, mode, loop_ms))
            except Exception as e:  # noqa: BLE001
                recs.append({"phase": f"ubench:{variant}", "error": repr(e)})
    finally:
        time.sleep(0.3)
        sampler.terminate()
        sampler.wait()
    samples = []
    first = None
    for line in open(SAMPLES):
        try:
            s = json.loads(line)
        except Exception:  # noqa: BLE001
            continue
        if s.get("first"):
            first = s
        else:
            samples.append(s)
    print("# tools/power_clock_lab.py -- sustained loops of %.1f s; amdsmi sampled at 50 Hz (samples from 0.25 s after a loop's start)" % LOOP_S)
    if first:
        gm = first.get("gpu_metrics") or {}
        keep = {k: gm.get(k) for k in ("current_socket_power", "average_socket_power", "current_gfxclk", "current_gfxclks", "current_uclk", "throttle_status", "temperature_hotspot") if k in gm}
        print("# sampler, first read:", json.dumps({**{k: v for k, v in first.items() if k not in ("gpu_metrics",)}, "gpu_metrics_subset": keep})[:1500])
    print(f"# {len(samples)} samples; read time per sample {1e3 * sum(s.get('dt_read', 0) for s in samples) / max(1, len(samples)):.2f} ms")
    for rec in recs:
        if "t0" in rec:
            rec["smi"] = join(rec, samples)
        rec.pop("t0", None)
        rec.pop("t1", None)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
