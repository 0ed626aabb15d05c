"""The reference's own behavioural tests, re-created on the tiny random-init Llama (its unit-test model needs the hub):
tests/presses/test_presses.py (every scorer under every wrapper, kept = highest scores, ChunkPress lengths),
tests/presses/test_finch_press.py, tests/test_generate.py (pipeline == model.generate under the press),
tests/test_pipeline.py (logged lengths, no press, empty question, cache invariance).
CPU: host logic over the oracle-backed entry points; GPU (marked): the same calls on the HIP kernels."""
import logging
from dataclasses import dataclass

import pytest
import torch

import _inputs


def _scorer_configs(P):
    """tests/default_presses.py restricted to the presses of this package: (class, [easy kwargs, hard kwargs])."""
    r = lambda **kw: [dict(compression_ratio=0.2, **kw), dict(compression_ratio=0.8, **kw)]
    return [(P.KnormPress, r()), (P.KeyDiffPress, r()), (P.RandomPress, r()), (P.StreamingLLMPress, r()), (P.ExpectedAttentionPress, r()),
            (P.SnapKVPress, r(window_size=2)), (P.TOVAPress, r()), (P.PyramidKVPress, r(window_size=2)), (P.CURPress, r()),
            (P.QFilterPress, r())]


def _make(P, cls, kwargs, model):
    press = cls(**kwargs)
    if cls is P.QFilterPress:   # the published filters need the hub (reference: post_init_from_model): seeded stand-ins
        cfg = model.config
        press.q_filters = torch.randn(cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim, generator=torch.Generator().manual_seed(0)).to(model.device)
        press.post_init_from_model = lambda m: None
    return press


def _presses_run(device, dtype):
    """test_presses_run (test_presses.py:64-110): every scorer, bare and under every wrapper, both ratios."""
    from transformers import DynamicCache

    import kvpress_amd as P
    from kvpress_amd import contrib as C

    model = _inputs.make_tiny_llama(dtype=dtype, device=device)
    ids = torch.randint(3, 59, (1, 128), generator=torch.Generator().manual_seed(0)).to(device)
    for wrapper in (None, P.ComposedPress, P.KeyRerotationPress, P.AdaKVPress, P.ChunkPress, C.BlockPress, C.ChunkKVPress):
        for cls, kw_list in _scorer_configs(P):
            for kwargs in kw_list:
                press = _make(P, cls, kwargs, model)
                if wrapper is P.ComposedPress:
                    press = P.ComposedPress(presses=[press])
                elif wrapper is P.ChunkPress:
                    press = P.ChunkPress(press=press, chunk_length=24)
                elif wrapper is C.BlockPress:
                    press = C.BlockPress(press=press, block_size=32)
                elif wrapper is not None:
                    press = wrapper(press=press)
                press.post_init_from_model(model)
                cache = DynamicCache()
                with torch.no_grad(), press(model):
                    model(ids, past_key_values=cache)
                assert hasattr(press, "compression_ratio")
                n = cache.get_seq_length()
                if wrapper is P.AdaKVPress:
                    assert n == 128                                # head-wise pruning masks, nothing is removed
                    for layer in model.model.layers:
                        layer.self_attn.masked_key_indices = None
                elif wrapper in (None, P.KeyRerotationPress, C.BlockPress) and cls is not P.PyramidKVPress:
                    assert n == int(128 * (1 - kwargs["compression_ratio"])), (cls.__name__, wrapper, n)
                else:
                    assert 0 < n < 128


def test_presses_run_cpu(fake_native):
    _presses_run("cpu", None)


def test_chunk_press_lengths(fake_native):
    """test_chunk_press (test_presses.py:42-50)."""
    from transformers import DynamicCache

    import kvpress_amd as P

    model = _inputs.make_tiny_llama()
    ids = torch.randint(3, 59, (1, 256), generator=torch.Generator().manual_seed(1))
    for chunk_length in (2, 4, 8, 128):
        cache = DynamicCache()
        with torch.no_grad(), P.ChunkPress(press=P.KnormPress(compression_ratio=0.5), chunk_length=chunk_length)(model):
            model(ids, past_key_values=cache)
        assert cache.get_seq_length() == 128


def _keep_highest(device, dtype):
    """test_presses_keep_highest_score (test_presses.py:143-162): batch 5, five ratios, a user-defined ScorerPress."""
    from transformers import DynamicCache

    import kvpress_amd as P

    @dataclass
    class StoreKnormPress(P.ScorerPress):
        def __post_init__(self):
            self.scores = []

        def score(self, module, hidden_states, keys, values, attentions, kwargs):
            scores = -keys.norm(dim=-1)
            self.scores.append(scores)
            return scores

    model = _inputs.make_tiny_llama(dtype=dtype, device=device)
    for ratio in (0.0, 0.2, 0.4, 0.6, 0.8):
        press = StoreKnormPress(compression_ratio=ratio)
        ids = torch.randint(3, 59, (5, 256), generator=torch.Generator().manual_seed(2)).to(device)
        cache = DynamicCache()
        with torch.no_grad(), press(model):
            model(ids, past_key_values=cache)
        for scores, layer in zip(press.scores, cache.layers):
            kept = -layer.keys.norm(dim=-1)
            n = kept.shape[-1]
            assert n == int(256 * (1 - ratio))
            assert torch.allclose(scores.float().sort(dim=-1).values[..., -n:] if n else scores[..., :0].float(), kept.float().sort(dim=-1).values)


def test_presses_keep_highest_score_cpu(fake_native):
    _keep_highest("cpu", None)


def _finch(device, dtype):
    """test_finch_press (test_finch_press.py:10-21): 10 tokens, the delimiter at position 8, no cache passed in."""
    import kvpress_amd as P

    model = _inputs.make_tiny_llama(dtype=dtype, device=device)
    for press in (P.FinchPress(0.5), P.FinchPress(0.5, rerotate_keys=False), P.FinchPress(0.5, normalize_scores=False), P.FinchPress(0.2, chunk_length=5)):
        press.delimiter_token_id = model.config.eos_token_id
        ids = torch.arange(10, 20).to(device)
        ids[8] = press.delimiter_token_id
        with torch.no_grad(), press(model):
            out = model(ids.unsqueeze(0))
        assert press.window_size == 1
        n = out.past_key_values.get_seq_length()
        assert n == (int(9 * 0.5) if press.chunk_length is None else sum(max(1, int(c * 0.8)) for c in (5, 4)))


def test_finch_press_cpu(fake_native):
    _finch("cpu", None)


def _generate(device, dtype):
    """test_generate (test_generate.py:9-26): ``model.generate`` under ``with press(model)`` -- the prompt is compressed once
    while it is pre-filled, the decoding steps are left alone, the hooks are gone afterwards.  (The reference also compares
    the text with the pipeline's answer; its pipeline compresses the context WITHOUT the trailing newline token, generate
    compresses the whole prompt, so that equality is a property of its unit-test checkpoint, not of the code.)"""
    from transformers import DynamicCache

    import kvpress_amd as P

    model, tok = _inputs.make_tiny_llama(dtype=dtype, device=device), _inputs.make_tiny_tokenizer()
    ids = tok.encode(tok.bos_token + _inputs.tiny_context(40), return_tensors="pt", add_special_tokens=False).to(device)
    n = ids.shape[1]
    press = P.KnormPress(compression_ratio=0.4)
    cache = DynamicCache()
    with torch.no_grad(), press(model):
        out = model.generate(input_ids=ids, attention_mask=torch.ones_like(ids), past_key_values=cache, max_new_tokens=10, do_sample=False,
                             pad_token_id=0)
    new = out.shape[1] - n
    assert 1 <= new <= 10
    assert cache.get_seq_length() == int(n * (1 - 0.4)) + new - 1      # the last token is never fed back
    assert all(len(layer.self_attn._forward_hooks) == 0 for layer in model.model.layers)
    assert isinstance(tok.decode(out[0, n:], skip_special_tokens=True), str)


def test_pipeline_logs_and_variants(fake_native, caplog):
    """test_pipeline / test_pipeline_single_or_no_question / test_pipeline_no_press_works / test_pipeline_compresses_context
    (tests/test_pipeline.py:19-31,83-98,132-142): logged context lengths, '' as question, no press, two questions."""
    from transformers import DynamicCache, pipeline

    import kvpress_amd as P

    pipe = pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(), tokenizer=_inputs.make_tiny_tokenizer())
    context = _inputs.tiny_context(22)   # 23 tokens with bos, like the reference's test sentence
    # (the reference also asks the empty question, which its tokenizer turns into the "\n" suffix token; the in-memory word
    # tokenizer has no such token, so that variant would feed zero tokens)
    for kwargs in (dict(questions=["w1 w2"]), dict(question="w1 w2"), dict(questions=["w1", "w1"])):
        caplog.clear()
        with caplog.at_level(logging.DEBUG):
            res = pipe(context, press=P.ExpectedAttentionPress(compression_ratio=0.4), cache=DynamicCache(), **kwargs)
        answers = res["answers"] if "questions" in kwargs else [res["answer"]]
        assert len(answers) == len(kwargs.get("questions", [0])) and all(isinstance(a, str) for a in answers)
        messages = [r.message for r in caplog.records]
        assert "Context Length: 23" in messages and "Compressed Context Length: 13" in messages, messages
    assert isinstance(pipe(context, question="w3")["answer"], str)       # no press


def test_pipeline_context_cache_is_invariant(fake_native):
    """test_pipeline_context_cache_is_invariant (tests/test_pipeline.py:145-166)."""
    from transformers import DynamicCache

    from kvpress_amd.pipeline import KVPressTextGenerationPipeline

    model, tok = _inputs.make_tiny_llama(), _inputs.make_tiny_tokenizer()
    pipe = KVPressTextGenerationPipeline(model=model, tokenizer=tok)
    q_ids = tok("w1 w2 w3", return_tensors="pt", add_special_tokens=False)["input_ids"]
    with torch.no_grad():
        cache = DynamicCache()
        model(input_ids=torch.randint(3, 59, (1, 256), generator=torch.Generator().manual_seed(3)), past_key_values=cache)
        keys = [l.keys.clone() for l in cache.layers]
        values = [l.values.clone() for l in cache.layers]
        lengths = [cache.get_seq_length(i) for i in range(len(cache))]
        pipe.generate_answer(q_ids, cache, context_length=22, max_new_tokens=10)
        pipe._remove_answer_from_cache(cache, lengths)
    assert cache.get_seq_length() == 256
    assert all(torch.equal(k, l.keys) for k, l in zip(keys, cache.layers)) and all(torch.equal(v, l.values) for v, l in zip(values, cache.layers))


def test_generate_cpu(fake_native):
    _generate("cpu", None)


@pytest.mark.gpu
def test_presses_run_gpu():
    _presses_run("cuda:0", torch.bfloat16)


@pytest.mark.gpu
def test_presses_keep_highest_score_gpu():
    _keep_highest("cuda:0", torch.float32)


@pytest.mark.gpu
def test_finch_press_gpu():
    _finch("cuda:0", torch.float32)


@pytest.mark.gpu
def test_generate_gpu():
    _generate("cuda:0", torch.bfloat16)
