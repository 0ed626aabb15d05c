#!/usr/bin/env python3
"""Golden answers for the pipeline test: the REAL reference pipeline (kvpress/pipeline.py) with the reference presses
on the tiny random-init Llama and the in-memory tokenizer of tests/_inputs.py -> tests/golden/pipeline.json.

Test infrastructure only.  Runs in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_pipeline.py
Shims: the cachetools / fire stubs of gen_golden.py, and -- transformers 5.x no longer passes ``cache_position`` to the
attention layers, which the reference hook reads (base_press.py:145) -- a forward pre-hook that re-creates it.
"""
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import gen_golden
    gen_golden._install_shims()
    import kvpress
    import torch
    from kvpress.pipeline import KVPressTextGenerationPipeline
    from transformers import DynamicCache

    import _inputs

    model = _inputs.make_tiny_llama()
    tok = _inputs.make_tiny_tokenizer()

    def inject_cache_position(module, args, kwargs):
        cache = kwargs["past_key_values"]
        past = cache.get_seq_length(module.layer_idx)
        q_len = kwargs["hidden_states"].shape[1]
        kwargs["cache_position"] = torch.arange(past, past + q_len)
        return args, kwargs

    for layer in model.model.layers:
        layer.self_attn.register_forward_pre_hook(inject_cache_position, with_kwargs=True)

    pipe = KVPressTextGenerationPipeline(model=model, tokenizer=tok)
    out = {}
    for name, (spec, n_words, questions, max_new) in _inputs.PIPELINE_CASES.items():
        press = _inputs.build_press(kvpress, spec)
        context = _inputs.tiny_context(n_words)
        cache = _inputs.make_pipeline_cache(name, model.config)
        res = pipe(context, questions=questions, press=press, max_new_tokens=max_new, cache=cache)
        out[name] = {
            "answers": res["answers"],
            "cache_lengths": [int(cache.get_seq_length(i)) for i in range(len(cache))],
            "context_tokens": int(tok.encode("<s>" + context, add_special_tokens=False).__len__()),
        }
        print(name, out[name])
    def ref_pipe(m, t):
        for layer in m.model.layers:
            layer.self_attn.register_forward_pre_hook(inject_cache_position, with_kwargs=True)
        return KVPressTextGenerationPipeline(model=m, tokenizer=t)

    for name in _inputs.FINCH_PIPELINE_CASES:
        cache = DynamicCache()
        res, press = _inputs.run_finch_pipeline(kvpress, ref_pipe, name, cache)
        out[name] = {"answers": res["answers"], "cache_lengths": [int(cache.get_seq_length(i)) for i in range(len(cache))],
                     "window_size": int(press.window_size)}
        print(name, out[name])
    for name, (family, spec, n_words, questions, max_new) in _inputs.FAMILY_PIPELINE_CASES.items():
        cache = DynamicCache()
        res = ref_pipe(_inputs.make_tiny_model(family), _inputs.make_tiny_tokenizer())(
            _inputs.tiny_context(n_words), questions=questions, press=_inputs.build_press(kvpress, spec), max_new_tokens=max_new, cache=cache)
        out[name] = {"answers": res["answers"], "cache_lengths": [int(cache.get_seq_length(i)) for i in range(len(cache))]}
        print(name, out[name])
    with open(os.path.join(REPO, "tests", "golden", "pipeline.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
