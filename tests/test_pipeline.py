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



def _run(name, device="cpu", dtype=None):
    from transformers import DynamicCache, pipeline

    import kvpress_amd.pipeline  # noqa: F401  (registers the task)

    import kvpress_amd

    spec, n_words, questions, max_new = _inputs.PIPELINE_CASES[name]
    model = _inputs.make_tiny_llama(dtype=dtype, device=device)
    pipe = pipeline("kv-press-text-generation", model=model, tokenizer=_inputs.make_tiny_tokenizer())
    cache = _inputs.make_pipeline_cache(name, model.config)
    res = pipe(_inputs.tiny_context(n_words), questions=questions, press=_inputs.build_press(kvpress_amd, spec), max_new_tokens=max_new,
               cache=cache)
    return res, [int(cache.get_seq_length(i)) for i in range(len(cache))]


@pytest.mark.parametrize("name", list(_inputs.PIPELINE_CASES))
def test_pipeline_matches_reference_answers_cpu(name, fake_native):
    res, lengths = _run(name)
    assert lengths == GOLD[name]["cache_lengths"]      # compressed length, and the answers were removed from the cache
    assert res["answers"] == GOLD[name]["answers"]


def _run_finch(name, device="cpu", dtype=None):
    from transformers import DynamicCache, pipeline

    import kvpress_amd

    cache = DynamicCache()
    res, press = _inputs.run_finch_pipeline(kvpress_amd, lambda m, t: pipeline("kv-press-text-generation", model=m, tokenizer=t), name, cache,
                                            dtype=dtype, device=device)
    assert press.window_size == GOLD[name]["window_size"]
    assert [int(cache.get_seq_length(i)) for i in range(len(cache))] == GOLD[name]["cache_lengths"]
    assert res["answers"] == GOLD[name]["answers"]


@pytest.mark.parametrize("name", list(_inputs.FINCH_PIPELINE_CASES))
def test_finch_pipeline_matches_reference_answers_cpu(name, fake_native):
    _run_finch(name)


def _run_family(name, device="cpu", dtype=None):
    from transformers import DynamicCache, pipeline

    import kvpress_amd

    family, spec, n_words, questions, max_new = _inputs.FAMILY_PIPELINE_CASES[name]
    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_model(family, dtype=dtype, device=device), tokenizer=_inputs.make_tiny_tokenizer())
    cache = DynamicCache()
    res = pipe(_inputs.tiny_context(n_words), questions=questions, press=_inputs.build_press(kvpress_amd, spec), max_new_tokens=max_new, cache=cache)
    assert [int(cache.get_seq_length(i)) for i in range(len(cache))] == GOLD[name]["cache_lengths"]
    assert res["answers"] == GOLD[name]["answers"]


@pytest.mark.parametrize("name", list(_inputs.FAMILY_PIPELINE_CASES))
def test_model_families_match_reference_answers_cpu(name, fake_native):
    """Qwen3 (q_norm), Phi3 (fused qkv_proj), Mistral, Qwen2 (projection biases): the supported families of base_press.py:27-34."""
    _run_family(name)


def test_single_question_and_registry(fake_native):
    from transformers import pipeline

    import kvpress_amd as P
    from kvpress_amd.pipeline import KVPressTextGenerationPipeline

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    assert isinstance(pipe, KVPressTextGenerationPipeline)
    spec, n_words, questions, max_new = _inputs.PIPELINE_CASES["pipe_knorm"]
    out = pipe(_inputs.tiny_context(n_words), question=questions[0], press=_inputs.build_press(P, spec), max_new_tokens=max_new)
    assert out == {"answer": GOLD["pipe_knorm"]["answers"][0]}
    with pytest.raises(AssertionError):
        pipe("w1 w2", question="w1", questions=["w2"])


def test_decoding_press_rejects_multiple_questions(fake_native):
    from transformers import pipeline

    import kvpress_amd as P

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    with pytest.raises(ValueError):
        pipe("w1 w2 w3", questions=["w1", "w2"], press=P.DecodingPress(P.KnormPress(), 2, 8, 0), max_new_tokens=2)


def test_per_layer_compression_press_reference_shapes(fake_native):
    """The reference's own test (tests/test_per_layer_compression_press.py:13-20): ratios [0.1, 1] on its 2-layer unit-test
    geometry -> [5, 2, 230, 6] and [5, 2, 0, 6].  (Through a decode the reference itself fails with sdpa on this
    transformers version -- different per-layer lengths -- so only the prefill is pinned.)"""
    from transformers import DynamicCache

    import kvpress_amd as P

    model = _inputs.make_tiny_llama()
    press = P.PerLayerCompressionPress(compression_ratios=[0.1, 1], press=P.KnormPress())
    ids = torch.randint(3, 59, (5, 256))
    with torch.no_grad(), press(model):
        cache = model(ids, past_key_values=DynamicCache()).past_key_values
    assert cache.layers[0].keys.shape == torch.Size([5, 2, 230, 6])
    assert cache.layers[1].keys.shape == torch.Size([5, 2, 0, 6])
    assert press.compression_ratio == 0.55 and press.press.compression_ratio == 0.0
    with pytest.raises(AttributeError):
        press.compression_ratio = 0.3


def test_find_target_compression_ratio():
    import kvpress_amd as P

    f = P.DecodingPress._find_target_compression_ratio
    for q_len in (41, 100, 1000, 4097, 131072):
        for target in (1, 7, 40, q_len - 1):
            if target < q_len:
                assert int(q_len * (1 - f(q_len, target))) == target
    assert f(10, 10) == 0.0 and f(5, 9) == 0.0


def test_context_truncation(fake_native):
    from transformers import DynamicCache, pipeline

    import kvpress_amd.pipeline  # noqa: F401

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    cache = DynamicCache()
    pipe(_inputs.tiny_context(100), question="w1", max_new_tokens=2, max_context_length=30, cache=cache)
    assert cache.get_seq_length() == 30


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_inputs.PIPELINE_CASES))
def test_pipeline_matches_reference_answers_gpu(name):
    """Same calls, fp32 model on cuda:0, presses on the HIP kernels."""
    res, lengths = _run(name, device="cuda:0", dtype=torch.float32)
    assert lengths == GOLD[name]["cache_lengths"]
    assert res["answers"] == GOLD[name]["answers"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_inputs.FINCH_PIPELINE_CASES))
def test_finch_pipeline_matches_reference_answers_gpu(name):
    _run_finch(name, device="cuda:0", dtype=torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_inputs.FAMILY_PIPELINE_CASES))
def test_model_families_match_reference_answers_gpu(name):
    _run_family(name, device="cuda:0", dtype=torch.float32)
