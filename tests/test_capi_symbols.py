"""The C-ABI library builds, loads, and exports every symbol include/kvpress_hip.h declares.
No compute is launched (no GPU here): only calls that return before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kvpress_amd import build as B

    B.build()
    from kvpress_amd import _native

    return _native.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kvpress_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kvp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kvpress_hip.h but not exported"


def test_binding_covers_header(lib):
    from kvpress_amd import _native

    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_version_and_errors(lib):
    assert lib.kvp_version() == 100
    # k > S is rejected before any HIP call
    rc = lib.kvp_topk_select(None, 1, 10, 10, 11, 0, None, None, 0, None)
    assert rc == -1 and b"bad shape" in lib.kvp_last_error()
    rc = lib.kvp_topk_select(None, 1, 10, 10, 5, 7, None, None, 0, None)   # no such order
    assert rc == -1 and b"bad order" in lib.kvp_last_error()
    rc = lib.kvp_topk_select(None, 1, 10, 10, 5, 1, None, None, 0, None)   # KVP_ORDER_SCORE: arguments are still checked
    assert rc == -1 and b"null pointer" in lib.kvp_last_error()
    rc = lib.kvp_rownorm_score(None, 7, 1, 1, 1, 1, 1, 1, 1, ctypes.c_float(1.0), None, None)
    assert rc == -1 and b"dtype" in lib.kvp_last_error()
    # snapkv: S must exceed the window (snapkv_press.py:84-86), kernel_size odd
    rc = lib.kvp_snapkv_score(None, 0, 0, 0, None, 0, 0, 0, 2, 1, 4, 2, 64, 64, 128, 5, None, None, 0, None)
    assert rc == -1 and b"greater than the window size" in lib.kvp_last_error()
    rc = lib.kvp_snapkv_score(None, 0, 0, 0, None, 0, 0, 0, 2, 1, 4, 2, 100, 64, 128, 4, None, None, 0, None)
    assert rc == -1 and b"odd" in lib.kvp_last_error()
    # ea: more tokens than n_sink (expected_attention_press.py:137)
    rc = lib.kvp_ea_score(None, 0, 0, 0, None, 0, 0, 0, 2, None, None, 1, 4, 2, 4, 128, 4, 1, ctypes.c_float(0.0), None, None, 0, None)
    assert rc == -1 and b"n_sink" in lib.kvp_last_error()


def test_workspace_sizes(lib):
    # Llama-3.1-8B, 128k: a few MiB of scratch, never anything of size [Hq, W, S]
    assert 0 < lib.kvp_topk_workspace_bytes(8, 131072, 65536) < 2 << 20
    ws = lib.kvp_snapkv_workspace_bytes(1, 32, 8, 131072, 64, 128)
    assert 4 << 20 < ws < 16 << 20
    assert lib.kvp_ea_score_workspace_bytes(1, 32, 8, 131072, 128) < 32 << 20


def test_no_cpu_fallback():
    import torch

    import kvpress_amd as P
    from kvpress_amd._native import KvpressHipError

    k = torch.zeros(1, 2, 16, 8)
    with pytest.raises(KvpressHipError):
        P.KnormPress(0.5).compress(None, None, k, k, None, {})
    # ratio 0 short-circuits before any kernel (scorer_press.py:86-87)
    k0, v0 = P.KnormPress(0.0).compress(None, None, k, k, None, {})
    assert k0 is k and v0 is k
    with pytest.raises(AssertionError):
        P.KnormPress(1.0)


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "kvpress_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                t = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "kvpress_oracle" in t:
                    bad.append(f)
    assert not bad, f"product files reference the oracle: {bad}"
