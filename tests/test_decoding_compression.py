"""DecodingPress / PrefillDecodingPress: the reference's own behavioural tests
(tests/test_decoding_compression.py:51-320) re-created on the tiny random-init Llama (their model needs the hub).
CPU: host logic over the oracle-backed entry points; GPU (marked): the same calls on the HIP kernels."""
import pytest
import torch

import _inputs


def _pipe(device="cpu", dtype=None):
    from transformers import pipeline

    import kvpress_amd  # noqa: F401  (registers the task)

    return pipeline("kv-press-text-generation", model=_inputs.make_tiny_llama(dtype=dtype, device=device), tokenizer=_inputs.make_tiny_tokenizer())


def _sizes(cache):
    return [layer.keys.shape[2] for layer in cache.layers]


def _check_decoding_compression(device, dtype):
    from transformers import DynamicCache

    import kvpress_amd as P

    pipe = _pipe(device, dtype)
    # test_decoding_compression (:51-83): the cache never exceeds target + interval - 1
    for target in (32, 64, 128):
        press = P.DecodingPress(base_press=P.KnormPress(compression_ratio=0.5), compression_interval=4, target_size=target)
        cache = DynamicCache()
        pipe(_inputs.tiny_context(100), question="w1 w2 w3", press=press, cache=cache, max_new_tokens=20)
        assert all(n <= target + press.compression_interval - 1 for n in _sizes(cache)), (target, _sizes(cache))
    # test_prefill_decoding_press_calls_both_phases (:86-117)
    combined = P.PrefillDecodingPress(prefilling_press=P.KnormPress(compression_ratio=0.6),
                                      decoding_press=P.DecodingPress(base_press=P.KnormPress(), compression_interval=3, target_size=48))
    cache = DynamicCache()
    pipe(_inputs.tiny_context(130), question="w4 w5", press=combined, cache=cache, max_new_tokens=15)
    assert all(48 <= n <= 48 + 3 - 1 for n in _sizes(cache)), _sizes(cache)
    # test_decoding_press_without_prefill (:120-148)
    press = P.DecodingPress(base_press=P.KnormPress(compression_ratio=0.4), compression_interval=5, target_size=64)
    cache = DynamicCache()
    pipe(_inputs.tiny_context(90), question="w6", press=press, cache=cache, max_new_tokens=25)
    assert all(64 <= n <= 64 + 5 - 1 for n in _sizes(cache)), _sizes(cache)
    # test_prefill_decoding_press_decoding_only (:151-184)
    combined = P.PrefillDecodingPress(prefilling_press=None,
                                      decoding_press=P.DecodingPress(base_press=P.KnormPress(compression_ratio=0.6), compression_interval=4, target_size=56))
    cache = DynamicCache()
    pipe(_inputs.tiny_context(100), question="w7 w8", press=combined, cache=cache, max_new_tokens=12)
    assert all(56 <= n <= 56 + 4 - 1 for n in _sizes(cache)), _sizes(cache)
    # test_decoding_press_equivalence (:187-232): standalone == PrefillDecodingPress(decoding only)
    mk = lambda: P.DecodingPress(base_press=P.KnormPress(compression_ratio=0.5), compression_interval=3, target_size=52)
    c1, c2 = DynamicCache(), DynamicCache()
    r1 = pipe(_inputs.tiny_context(80), question="w9", press=mk(), cache=c1, max_new_tokens=10)
    r2 = pipe(_inputs.tiny_context(80), question="w9", press=P.PrefillDecodingPress(prefilling_press=None, decoding_press=mk()), cache=c2,
              max_new_tokens=10)
    assert _sizes(c1) == _sizes(c2) and r1["answer"] == r2["answer"]


def test_decoding_compression_cpu(fake_native):
    _check_decoding_compression("cpu", None)


def _scorers(P):
    # the reference's default configurations (tests/default_presses.py; PyramidKV is skipped there: its per-layer budgets
    # do not meet one target size)
    return [P.KnormPress(0.2), P.KeyDiffPress(0.2), P.RandomPress(0.2), P.StreamingLLMPress(0.2), P.TOVAPress(0.2), P.CURPress(0.2),
            P.SnapKVPress(0.2, window_size=2), P.ExpectedAttentionPress(0.2)]


def _check_all_scorers_and_reuse(device, dtype):
    """test_all_presses_work_with_decoding_press (:278-330) and test_decoding_press_reuse_across_sequences (:333-350)."""
    from transformers import DynamicCache

    import kvpress_amd as P

    pipe = _pipe(device, dtype)
    for base in _scorers(P):
        press = P.DecodingPress(base_press=base, compression_interval=3, target_size=48)
        cache = DynamicCache()
        res = pipe(_inputs.tiny_context(70), question="w1 w2", press=press, cache=cache, max_new_tokens=12)
        assert isinstance(res["answer"], str)
        assert all(48 <= n <= 48 + 3 - 1 for n in _sizes(cache)), (type(base).__name__, _sizes(cache))
    # one press object, two sequences: the per-layer buffers and step counters start afresh
    press = P.DecodingPress(base_press=P.KnormPress(), compression_interval=3, target_size=40)
    outs = []
    for _ in range(2):
        cache = DynamicCache()
        outs.append((pipe(_inputs.tiny_context(60), question="w3", press=press, cache=cache, max_new_tokens=9)["answer"], _sizes(cache)))
    assert outs[0] == outs[1]


def test_all_scorers_under_decoding_press_cpu(fake_native):
    _check_all_scorers_and_reuse("cpu", None)


def test_decoding_press_argument_checks(fake_native):
    import kvpress_amd as P

    with pytest.raises(AssertionError):
        P.DecodingPress(base_press=P.ChunkPress(P.KnormPress()), compression_interval=2, target_size=8)
    with pytest.raises(AssertionError):
        P.DecodingPress(base_press=P.KnormPress(), compression_interval=0, target_size=8)
    with pytest.raises(AssertionError):
        P.DecodingPress(base_press=P.KnormPress(), compression_interval=2, target_size=0)


@pytest.mark.gpu
def test_decoding_compression_gpu():
    _check_decoding_compression("cuda:0", torch.float32)


@pytest.mark.gpu
def test_all_scorers_under_decoding_press_gpu():
    _check_all_scorers_and_reuse("cuda:0", torch.bfloat16)


def test_compression_ratio_decoding_press_target_size(fake_native):
    """The reference's unit tests of CompressionRatioDecodingPress (tests/test_decoding_compression.py:235-270), verbatim."""
    import kvpress_amd as P

    press = P.CompressionRatioDecodingPress(base_press=P.KnormPress(), target_compression_ratio=0.5)
    target = press._resolve_target_size({"position_ids": torch.tensor([[107]])})
    assert target == 54
    assert press._find_target_compression_ratio(58, target) == pytest.approx(1 - (54 / 58))
    # logical positions win over the (compressed) cache position
    assert press._resolve_target_size({"position_ids": torch.tensor([[107]]), "cache_position": torch.tensor([57])}) == 54
    with pytest.raises(NotImplementedError, match="requires logical position_ids"):
        press._resolve_target_size({"cache_position": torch.tensor([57])})
    assert press._find_target_compression_ratio(50, target) == 0.0     # already below the target: no-op
    with pytest.raises(AssertionError):
        P.CompressionRatioDecodingPress(base_press=P.KnormPress(), target_compression_ratio=1.0)
