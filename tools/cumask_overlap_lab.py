#!/usr/bin/env python3
"""VERDICT r5 #2 probe: does the power-bound window-attention work of ONE head group run beside the HBM-bound gather of the
OTHER head group faster than back to back, when each is confined to its own CUs (hipExtStreamCreateWithCUMask)?

The product chain is per-(batch, kv-head) local, so heads 4-7 could be scored while heads 0-3 are gathered.  round 2's
tools/overlap_lab.py ran the two on plain streams (both chains then want all CUs: the 8-wave / 256-register pass workgroups
leave no register file for a gather wave on their CU) and lost.  Here the gather stream owns `ng` CUs and the score stream the
other 256 - ng (KVP_SK_SLOTS sizes the passes' grid to that), masks laid out round-robin over the XCDs the way KFD maps
cu_mask bits (bit i -> XCC i % 8), so both streams keep all 8 L2s / all HBM channels.

    python tools/cumask_overlap_lab.py            # prints one line per CU split + the unmasked reference lines

Measurement aid (tools/), not part of the product path."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kvpress_amd import _native  # noqa: E402

_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found")
    return _hip


def masked_stream(cu_bits):
    """stream whose queues run only on the CUs whose bit is set (list of 256 booleans, KFD bit order)"""
    words = (ctypes.c_uint32 * 8)()
    for i, on in enumerate(cu_bits):
        if on:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip().hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value)


def timeit(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    S = 131072
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    keys = torch.randn((1, 8, S, 128), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    values = torch.randn((1, 8, S, 128), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    q = (torch.randn((1, 32, 64, 128), generator=g, device=dev, dtype=torch.float32) * 1.3).to(torch.bfloat16)
    kA, vA, kB, qB = keys[:, :4], values[:, :4], keys[:, 4:], q[:, 16:]
    sc = _native.snapkv_score(q, keys, 5)
    idxA = _native.topk_select(sc[:, :4].contiguous(), S // 2)
    main_s = torch.cuda.current_stream()

    def score_all():
        _native.snapkv_score(q, keys, 5)

    def gather_all():
        _native.gather_kv(keys, values, _IDX_ALL)

    def score_B():
        _native.snapkv_score(qB, kB, 5)

    def gather_A():
        _native.gather_kv(kA, vA, idxA)

    global _IDX_ALL
    _IDX_ALL = _native.topk_select(sc, S // 2)

    os.environ["KVP_SK_SLOTS"] = "256"
    _native.tuning_reload()
    print(f"unmasked, whole chip: score(8 heads) {timeit(score_all):6.1f}  gather(8 heads) {timeit(gather_all):6.1f}  "
          f"score(heads 4-7) {timeit(score_B):6.1f}  gather(heads 0-3) {timeit(gather_A):6.1f}  "
          f"back to back score(4-7); gather(0-3) {timeit(lambda: (score_B(), gather_A())):6.1f} us", flush=True)

    # which CUs a mask bit enables is probed by tools/cumask_map.hip; both plausible layouts are run: "rr" = bit i is CU i / 8 of XCD i % 8
    # (KFD's round-robin order), "blk" = bit i is CU i % 32 of XCD i / 32.  Either way the gather gets ng / 8 CUs of EVERY XCD.
    for layout, ng in [(l, n) for l in ("rr", "blk") for n in (64, 96, 128)]:
        nb = 256 - ng
        if layout == "rr":
            bits_g = [(i // 8) < (ng // 8) for i in range(256)]
        else:
            bits_g = [(i % 32) < (ng // 8) for i in range(256)]
        bits_s = [not b for b in bits_g]
        sg, ss = masked_stream(bits_g), masked_stream(bits_s)
        os.environ["KVP_SK_SLOTS"] = str(nb)
        _native.tuning_reload()

        def on(stream, fn):
            def run():
                ev = torch.cuda.Event()
                ev.record(main_s)
                stream.wait_event(ev)
                with torch.cuda.stream(stream):
                    fn()
                main_s.wait_stream(stream)
            return run

        def conc():
            ev = torch.cuda.Event()
            ev.record(main_s)
            ss.wait_event(ev)
            sg.wait_event(ev)
            with torch.cuda.stream(ss):
                score_B()
            with torch.cuda.stream(sg):
                gather_A()
            main_s.wait_stream(ss)
            main_s.wait_stream(sg)

        t_s, t_g, t_c = timeit(on(ss, score_B)), timeit(on(sg, gather_A)), timeit(conc)
        print(f"[{layout}] gather on {ng:3d} CUs | score on {nb:3d} CUs (KVP_SK_SLOTS={nb}): score(4-7) alone {t_s:6.1f}  gather(0-3) alone {t_g:6.1f}  "
              f"concurrent {t_c:6.1f} us", flush=True)
    os.environ["KVP_SK_SLOTS"] = "256"
    _native.tuning_reload()


if __name__ == "__main__":
    main()
