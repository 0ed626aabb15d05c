#!/usr/bin/env python3
"""Fuzz the "kv-press-text-generation" pipeline of this package (host logic over the oracle-backed entry points, CPU) against
the REAL reference pipeline with the reference presses: random context lengths, questions, generation lengths and press
configurations on the tiny random-init Llama.  Answers and per-layer cache lengths must be identical.
Test infrastructure only; runs in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/fuzz_pipeline_against_reference.py [n_rounds] [seed]
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def main(argv):
    import gen_golden
    gen_golden._install_shims()
    import kvpress as R
    import numpy as np
    import torch
    from _pytest.monkeypatch import MonkeyPatch
    from kvpress.pipeline import KVPressTextGenerationPipeline as RefPipeline
    from transformers import DynamicCache

    import _inputs
    import conftest
    import types

    import kvpress_amd
    import kvpress_amd.contrib

    # one namespace with the reference's flat layout: the package proper + its contrib sub-package (presses outside SURVEY §8)
    P = types.SimpleNamespace(__name__="kvpress_amd", **{k: getattr(kvpress_amd, k) for k in kvpress_amd.__all__},
                              **{k: getattr(kvpress_amd.contrib, k) for k in kvpress_amd.contrib.__all__})
    from kvpress_amd.pipeline import KVPressTextGenerationPipeline as OurPipeline

    mp = MonkeyPatch()
    conftest.fake_native._get_wrapped_function()(mp)
    n_rounds = int(argv[0]) if argv else 20
    rs = np.random.RandomState(int(argv[1]) if len(argv) > 1 else 0)
    KN = lambda r=0.0: ("KnormPress", dict(compression_ratio=r))

    def inject_cache_position(module, args, kwargs):   # transformers 5.x no longer passes it; the reference hook reads it
        cache = kwargs["past_key_values"]
        past = cache.get_seq_length(module.layer_idx)
        kwargs["cache_position"] = torch.arange(past, past + kwargs["hidden_states"].shape[1])
        return args, kwargs

    def specs(r):
        w = int(rs.randint(2, 10))
        return [
            KN(r), ("SnapKVPress", dict(compression_ratio=r, window_size=w, kernel_size=int(rs.choice([1, 3, 5])))),
            ("ExpectedAttentionPress", dict(compression_ratio=r, n_sink=int(rs.randint(0, 5)), n_future_positions=int(rs.randint(1, 64)))),
            ("TOVAPress", dict(compression_ratio=r)), ("KeyDiffPress", dict(compression_ratio=r)), ("CURPress", dict(compression_ratio=r)),
            ("LagKVPress", dict(compression_ratio=r, n_sink=int(rs.randint(0, 4)), lag_size=int(rs.randint(4, 20)), cross_scoring=True)),
            ("KeyRerotationPress", dict(press=KN(r))), ("ChunkPress", dict(press=KN(r), chunk_length=int(rs.randint(8, 40)))),
            ("ChunkKVPress", dict(press=KN(r), chunk_length=int(rs.randint(4, 30)))), ("BlockPress", dict(press=KN(r), block_size=int(rs.randint(4, 40)))),
            ("AdaKVPress", dict(press=KN(r), alpha_safeguard=float(rs.choice([0.0, 0.2])))),
            ("ComposedPress", dict(presses=[KN(r / 2), ("ThinKPress", dict(key_channel_compression_ratio=0.5, window_size=w))])),
            ("DecodingPress", dict(base_press=KN(), compression_interval=int(rs.randint(2, 6)), target_size=int(rs.randint(20, 50)),
                                   hidden_states_buffer_size=int(rs.randint(0, 8)))),
            ("PrefillDecodingPress", dict(prefilling_press=KN(r), decoding_press=("DecodingPress", dict(
                base_press=KN(), compression_interval=int(rs.randint(2, 6)), target_size=int(rs.randint(15, 40)))))),
        ]

    bad = 0
    for it in range(n_rounds):
        r = float(rs.choice([0.2, 0.4, 0.5, 0.7]))
        all_specs = specs(r)
        spec = all_specs[int(rs.randint(len(all_specs)))]
        # the other supported families (q_norm, fused qkv_proj, projection biases); the offline statistics are sized for the Llama
        family = str(rs.choice(["llama", "llama", "qwen3", "phi3", "mistral", "qwen2"]))
        n_words = int(rs.randint(45, 140))
        single = spec[0] in ("DecodingPress", "PrefillDecodingPress") or rs.rand() < 0.5
        questions = [" ".join(f"w{int(x)}" for x in rs.randint(0, 56, int(rs.randint(1, 5)))) for _ in range(1 if single else 2)]
        max_new = int(rs.randint(3, 14))
        context = _inputs.tiny_context(n_words, seed=int(rs.randint(1 << 20)))
        res = []
        for ns, Pipe in ((R, RefPipeline), (P, OurPipeline)):
            model, tok = _inputs.make_tiny_model(family), _inputs.make_tiny_tokenizer()
            if ns is R:
                for layer in model.model.layers:
                    layer.self_attn.register_forward_pre_hook(inject_cache_position, with_kwargs=True)
            cache = DynamicCache()
            try:
                out = Pipe(model=model, tokenizer=tok)(context, questions=questions, press=_inputs.build_press(ns, spec), max_new_tokens=max_new, cache=cache)
                res.append((out["answers"], [int(cache.get_seq_length(i)) for i in range(len(cache))]))
            except Exception as e:   # both sides must fail alike (e.g. per-layer lengths that sdpa cannot decode with)
                res.append(("raised " + type(e).__name__, None))
        ok = res[0] == res[1]
        bad += not ok
        print(f"round {it}: {spec[0]} {family} r={r} ctx={n_words} q={len(questions)} new={max_new} -> {'OK' if ok else res}", flush=True)
    mp.undo()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
