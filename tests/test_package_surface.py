"""The package's import surface next to the reference's (kvpress/__init__.py:7-49): §8 presses at top level, the five out-of-scope
presses that exist here reachable under the reference's top-level names (lazily, from kvpress_amd.contrib), the rest absent with a
message that says why."""
import subprocess
import sys

import pytest


def test_contrib_presses_resolve_lazily_under_the_reference_names():
    code = ("import sys, kvpress_amd as P\n"
            "assert 'kvpress_amd.contrib' not in sys.modules\n"
            "from kvpress_amd import ThinKPress, LagKVPress, ObservedAttentionPress, ChunkKVPress, BlockPress\n"
            "import kvpress_amd.contrib as C\n"
            "assert ThinKPress is C.ThinKPress and P.BlockPress is C.BlockPress\n"
            "assert 'ThinKPress' in dir(P)\n"
            "assert len(P.SUPPORTED_MODELS) == 6\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_presses_outside_the_package_say_so():
    import kvpress_amd as P

    for name in ("DMSPress", "CriticalKVPress", "KVzipPress", "SimLayerKVPress"):
        with pytest.raises(AttributeError, match="outside the hot-path scope"):
            getattr(P, name)
        with pytest.raises(ImportError):
            exec(f"from kvpress_amd import {name}")
    with pytest.raises(AttributeError):
        P.NoSuchPress
    for name in P.__all__:
        assert getattr(P, name) is not None
