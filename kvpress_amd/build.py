"""Build libkvpress_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m kvpress_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written to kvpress_amd/lib/ (git-ignored,
but it travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libkvpress_hip.so")
# Test-only twin of the library: identical objects except topk_cluster.hip compiled with -DKVP_TC_FAULT_INJECTION (one workgroup of the
# cluster select can be made to arrive late at its barrier: tests/test_gpu_cluster_failure.py loads it in a child process through
# KVPRESS_HIP_LIB).  The product library carries no test hooks.
FAULT_LIB = os.path.join(LIBDIR, "libkvpress_hip_faultinject.so")
# Kernels of the presses OUTSIDE SURVEY section 8 (kvpress_amd.contrib: LagKV, ThinK, ObservedAttention; csrc/contrib/*.hip,
# include/kvpress_hip_extra.h): their own shared library, linked against the product library for its error / launch plumbing.
CONTRIB_LIB = os.path.join(LIBDIR, "libkvpress_hip_contrib.so")
CONTRIB_SRC = os.path.join(CSRC, "contrib")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# MFMA results straight into VGPRs (gfx950 has a unified register file): the softmax epilogues read every
# accumulator element with VALU ops, and v_accvgpr_read was 16 of their 73 instructions per 16 logits.
FILE_FLAGS = {"snapkv_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "ea_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def flags_for(src: str):
    return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def contrib_sources():
    return sorted(os.path.join(CONTRIB_SRC, f) for f in os.listdir(CONTRIB_SRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hdrs


def up_to_date() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(FAULT_LIB) or not os.path.exists(CONTRIB_LIB):
        return False
    t = min(os.path.getmtime(LIB), os.path.getmtime(FAULT_LIB), os.path.getmtime(CONTRIB_LIB))
    return all(os.path.getmtime(p) <= t for p in sources() + contrib_sources() + _deps() + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = max(os.path.getmtime(p) for p in _deps())

    def compile_one(src):
        obj = os.path.join(OBJDIR, ("contrib_" if os.path.dirname(src) == CONTRIB_SRC else "") + os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc, *flags_for(src), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        all_objs = list(ex.map(compile_one, sources() + contrib_sources()))
    objs, cobjs = all_objs[:len(sources())], all_objs[len(sources()):]
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    # the out-of-scope presses' kernels: resolved against the product library next to it ($ORIGIN)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", CONTRIB_LIB + ".tmp", *cobjs,
           "-L" + LIBDIR, "-l:libkvpress_hip.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed (contrib):\n{r.stdout}\n{r.stderr}")
    os.replace(CONTRIB_LIB + ".tmp", CONTRIB_LIB)
    # the fault-injection twin (tests only)
    src = os.path.join(CSRC, "topk_cluster.hip")
    fobj = os.path.join(OBJDIR, "topk_cluster.faultinject.obj")
    cmd = [hipcc, *flags_for(src), "-DKVP_TC_FAULT_INJECTION", "-c", src, "-o", fobj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src} (fault injection):\n{r.stdout}\n{r.stderr}")
    fobjs = [fobj if os.path.basename(o) == "topk_cluster.o" else o for o in objs]
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", FAULT_LIB + ".tmp", *fobjs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed (fault injection):\n{r.stdout}\n{r.stderr}")
    os.replace(FAULT_LIB + ".tmp", FAULT_LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
