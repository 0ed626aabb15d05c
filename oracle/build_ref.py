#!/usr/bin/env python3
"""oracle/_ref: the REAL reference package, for bench.py's ``cpu_baseline`` (kind "reference") and nothing else.

    python oracle/build_ref.py            (called by __graft_entry__.build() when /root/reference exists)

The reference (NVIDIA/kvpress v0.5.4) is pure Python: "building" it means making its package importable where the benchmark
runs.  /root/reference does not exist on the GPU box, so this recipe copies ``/root/reference/kvpress`` -- the package
directory, unmodified, sources where they lie -- into ``oracle/_ref/kvpress`` and writes stand-ins for the two imports the image
lacks (``cachetools``: duo_attention_press.py:10, lukv_press.py:11; ``fire``: expected_attention_with_stats.py:10 -- SURVEY §8c).
``oracle/_ref/`` is git-ignored (no reference source ever enters the history) but not gpurun-ignored, so the copy travels
with the tree like the built ``.so``.  TEST / MEASUREMENT INFRASTRUCTURE: only bench.py's cpu_baseline leg imports it; the
product (kvpress_amd/) never does (tests/test_capi_symbols.py::test_product_never_imports_oracle)."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("KVPRESS_REFERENCE", "/root/reference")
DST = os.path.join(HERE, "_ref")

CACHETOOLS = '''"""stand-in for cachetools (not installed): just enough for `from cachetools import LRUCache, cached`"""


class LRUCache(dict):
    def __init__(self, maxsize=128):
        super().__init__()


def cached(cache=None, **kw):
    return lambda f: f
'''


def build(verbose: bool = False) -> str | None:
    src = os.path.join(REF_SRC, "kvpress")
    if not os.path.isfile(os.path.join(src, "__init__.py")):
        return None          # the GPU box: use the copy that travelled with the tree (if any)
    os.makedirs(DST, exist_ok=True)
    dst = os.path.join(DST, "kvpress")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(src, dst, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    with open(os.path.join(DST, "cachetools.py"), "w") as f:
        f.write(CACHETOOLS)
    with open(os.path.join(DST, "fire.py"), "w") as f:
        f.write('"""stand-in for fire (not installed; only a CLI entry point of the reference imports it)"""\n')
    if verbose:
        n = sum(len(fs) for _, _, fs in os.walk(dst))
        print(f"oracle/_ref: copied {n} files of {src}")
    return dst


if __name__ == "__main__":
    out = build(verbose=True)
    if out is None:
        print(f"{REF_SRC}/kvpress not found: nothing to do", file=sys.stderr)
