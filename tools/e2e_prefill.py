#!/usr/bin/env python3
"""End-to-end prefill under a press (SURVEY.md §8 f-1 / §8d "end-to-end prefill tok/s"): a random-init Llama-3.1-8B
(32 layers, bf16, no lm_head) pre-fills S synthetic tokens with and without ``with press(model):``; the difference is
what the press costs inside a real forward pass (hook dispatch, cache write-back, allocator traffic included).

    python tools/e2e_prefill.py [--seq-len 32768] [--press snapkv|knorm|ea] [--ratio 0.5] [--layers 32] [--reps 5]

Runs ALTERNATE (no press, press, no press, ...) after one warm-up each, so that clock / thermal drift hits both alike; reported
are median, min, max and the standard deviation of each, and the overhead per layer from the medians.  One more, instrumented
prefill brackets every layer's ``press.compress`` call with events on the model's stream and times the hook on the host: that
splits the overhead into the compress call itself (kernels + the model-owned q_proj), and the rest (cache write-back, freeing
the uncompressed K/V, allocator, hook dispatch).  Prints one JSON line.  Not part of bench.py's contract (which times the
per-layer hot path); run by hand on the GPU box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq-len", type=int, default=32768)
    ap.add_argument("--press", default="snapkv", choices=["snapkv", "knorm", "ea", "pyramidkv", "tova", "keydiff"])
    ap.add_argument("--ratio", type=float, default=0.5)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()

    import torch
    from transformers import DynamicCache, LlamaConfig, LlamaModel

    import kvpress_amd as P

    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(
        hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, head_dim=128, num_hidden_layers=args.layers,
        intermediate_size=14336, vocab_size=128256, max_position_embeddings=131072, rope_theta=500000.0,
        rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 8192},
        attention_bias=False, rms_norm_eps=1e-5,
    )
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    t0 = time.perf_counter()
    with torch.device(dev):
        torch.set_default_dtype(torch.bfloat16)
        model = LlamaModel(cfg).eval()
        torch.set_default_dtype(torch.float32)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    press = {"snapkv": P.SnapKVPress, "knorm": P.KnormPress, "ea": P.ExpectedAttentionPress, "pyramidkv": P.PyramidKVPress,
             "tova": P.TOVAPress, "keydiff": P.KeyDiffPress}[args.press](compression_ratio=args.ratio)
    ids = torch.randint(0, cfg.vocab_size, (1, args.seq_len), device=dev)

    def prefill(p):
        cache = DynamicCache()
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.no_grad():
            if p is None:
                model(input_ids=ids, past_key_values=cache, use_cache=True)
            else:
                with p(model):
                    model(input_ids=ids, past_key_values=cache, use_cache=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t, [cache.get_seq_length(i) for i in (0, args.layers - 1)]

    import statistics

    for p in (None, press):
        prefill(p)  # warm-up (allocator, kernel selection)
    ts = {"no_press": [], "press": []}
    lens = None
    for _ in range(args.reps):   # alternating
        dt, _l = prefill(None)
        ts["no_press"].append(dt)
        dt, lens = prefill(press)
        ts["press"].append(dt)

    def stats(v):
        return {"median_s": round(statistics.median(v), 4), "min_s": round(min(v), 4), "max_s": round(max(v), 4),
                "stdev_s": round(statistics.pstdev(v), 4), "all_s": [round(x, 4) for x in v]}

    res = {k: stats(v) for k, v in ts.items()}
    extra = res["press"]["median_s"] - res["no_press"]["median_s"]

    # ---- instrumented prefill: where does the per-layer overhead go? -------------------------------------------------------
    events, host = [], []
    orig_compress, orig_hook = type(press).compress, type(press).forward_hook

    class Instrumented(type(press)):
        def compress(self, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_compress(self, *a, **k)
            e1.record()
            events.append((e0, e1))
            return out

        def forward_hook(self, *a, **k):
            t = time.perf_counter()
            out = orig_hook(self, *a, **k)
            host.append(time.perf_counter() - t)
            return out

    ipress = Instrumented(compression_ratio=args.ratio)
    t_instr, _ = prefill(ipress)
    torch.cuda.synchronize()
    gpu_ms = [e0.elapsed_time(e1) for e0, e1 in events]
    print(json.dumps({
        "what": "end-to-end prefill, random-init Llama-3.1-8B body (no lm_head), bf16, sdpa attention; alternating runs", "seq_len": args.seq_len,
        "layers": args.layers, "press": args.press, "compression_ratio": args.ratio, "reps": args.reps, "model_build_s": round(t_build, 1),
        "no_press": res["no_press"], "press": res["press"],
        "tok_s_no_press": round(args.seq_len / res["no_press"]["median_s"], 1), "tok_s_press": round(args.seq_len / res["press"]["median_s"], 1),
        "press_overhead_ms_per_layer": round(extra * 1e3 / args.layers, 4),
        "overhead_vs_run_to_run_stdev": round(extra / max(1e-9, (res["press"]["stdev_s"] ** 2 + res["no_press"]["stdev_s"] ** 2) ** 0.5), 2),
        "breakdown_ms_per_layer": {
            "compress_call_on_stream": round(sum(gpu_ms) / len(gpu_ms), 4),           # kernels + q_proj of the window, inside the model
            "compress_call_min_max": [round(min(gpu_ms), 4), round(max(gpu_ms), 4)],
            "hook_host_time": round(sum(host) / len(host) * 1e3, 4),                  # Python + launches of the whole hook (asynchronous work excluded)
            "instrumented_prefill_s": round(t_instr, 4),
        },
        "cache_len_first_last": lens, "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
    }))


if __name__ == "__main__":
    main()
