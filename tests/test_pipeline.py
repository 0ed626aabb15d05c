"""KVPressTextGenerationPipeline against the REAL reference pipeline's answers.

tests/golden/pipeline.json was produced by oracle/gen_golden_pipeline.py: the reference pipeline + reference presses on
the tiny random-init Llama / in-memory tokenizer of tests/_inputs.py.  Here the same calls go through
kvpress_amd.pipeline with this package's presses; on CPU the HIP entry points are replaced by the oracle-backed fakes
of conftest.py (host logic under test), on the GPU box the real kernels run (marked gpu)."""
import json
import os

import pytest
import torch

import _inputs

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pipeline.json")))


def _press(kind, kw):
    import kvpress_amd as P

    return {None: lambda **k: None, "knorm": P.KnormPress, "snapkv": P.SnapKVPress, "ea": P.ExpectedAttentionPress}[kind](**kw)


def _run(name, device="cpu", dtype=None):
    from transformers import DynamicCache, pipeline

    import kvpress_amd.pipeline  # noqa: F401  (registers the task)

    kind, kw, n_words, questions, max_new = _inputs.PIPELINE_CASES[name]
    model = _inputs.make_tiny_llama(dtype=dtype, device=device)
    pipe = pipeline("kv-press-text-generation", model=model, tokenizer=_inputs.make_tiny_tokenizer())
    cache = DynamicCache()
    res = pipe(_inputs.tiny_context(n_words), questions=questions, press=_press(kind, kw), max_new_tokens=max_new, cache=cache)
    return res, [int(cache.get_seq_length(i)) for i in range(len(cache))]


@pytest.mark.parametrize("name", list(_inputs.PIPELINE_CASES))
def test_pipeline_matches_reference_answers_cpu(name, fake_native):
    res, lengths = _run(name)
    assert lengths == GOLD[name]["cache_lengths"]      # compressed length, and the answers were removed from the cache
    assert res["answers"] == GOLD[name]["answers"]


def test_single_question_and_registry(fake_native):
    from transformers import pipeline

    import kvpress_amd as P
    from kvpress_amd.pipeline import KVPressTextGenerationPipeline

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    assert isinstance(pipe, KVPressTextGenerationPipeline)
    kind, kw, n_words, questions, max_new = _inputs.PIPELINE_CASES["pipe_knorm"]
    out = pipe(_inputs.tiny_context(n_words), question=questions[0], press=P.KnormPress(**kw), max_new_tokens=max_new)
    assert out == {"answer": GOLD["pipe_knorm"]["answers"][0]}
    with pytest.raises(AssertionError):
        pipe("w1 w2", question="w1", questions=["w2"])


def test_context_truncation(fake_native):
    from transformers import DynamicCache, pipeline

    import kvpress_amd.pipeline  # noqa: F401

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    cache = DynamicCache()
    pipe(_inputs.tiny_context(100), question="w1", max_new_tokens=2, max_context_length=30, cache=cache)
    assert cache.get_seq_length() == 30


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pipe_knorm", "pipe_snapkv", "pipe_ea", "pipe_none"])
def test_pipeline_matches_reference_answers_gpu(name):
    """Same calls, fp32 model on cuda:0, presses on the HIP kernels."""
    res, lengths = _run(name, device="cuda:0", dtype=torch.float32)
    assert lengths == GOLD[name]["cache_lengths"]
    assert res["answers"] == GOLD[name]["answers"]
