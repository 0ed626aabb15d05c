#!/usr/bin/env python3
"""Do the issue-bound window-attention passes and the HBM-bound gather run faster side by side than back to back?
Launches kvp_snapkv_score (p1, combine, p2, pool) on one stream and kvp_gather_kv of an independent selection on
another, at the BASELINE shape, and prints wall times: each alone, back to back on one stream, and concurrent.
Measurement aid (tools/), not part of the product path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kvpress_amd import _native  # noqa: E402


def timeit(fn, n=30):
    for _ in range(10):
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
    sc = _native.snapkv_score(q, keys, 5)
    idx = _native.topk_select(sc, S // 2)
    for prio in (False, True):
        sa = torch.cuda.Stream(priority=-1 if prio else 0)
        sb = torch.cuda.Stream()
        main_s = torch.cuda.current_stream()

        def score():
            _native.snapkv_score(q, keys, 5)

        def gather():
            _native.gather_kv(keys, values, idx)

        def both_serial():
            score(); gather()

        def both_conc():
            ev = torch.cuda.Event()
            ev.record(main_s)
            sa.wait_event(ev); sb.wait_event(ev)
            with torch.cuda.stream(sa):
                score()
            with torch.cuda.stream(sb):
                gather()
            main_s.wait_stream(sa); main_s.wait_stream(sb)

        for cfg in sys.argv[1:] or [""]:
            for kv in cfg.split():
                k, v = kv.split("=")
                os.environ[k] = v
            t_s, t_g, t_ser, t_con = timeit(score), timeit(gather), timeit(both_serial), timeit(both_conc)
            print(f"prio={int(prio)} {cfg:40s} score {t_s:6.1f}  gather {t_g:6.1f}  serial {t_ser:6.1f}  concurrent {t_con:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
