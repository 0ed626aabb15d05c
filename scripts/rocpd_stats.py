#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/min/max duration) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats` on ROCm 7.x writes <name>_results.db).

usage: rocpd_stats.py results.db [skip_first_n_dispatches_per_kernel]
"""
import sqlite3
import sys


def main(path, skip=0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    q = f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""
    per = {}
    for name, st, en in cur.execute(q):
        per.setdefault(name, []).append((en - st) / 1e3)
    rows = []
    for name, ds in per.items():
        ds = ds[skip:] if len(ds) > skip else ds
        rows.append((sum(ds), name, len(ds), sum(ds) / len(ds), min(ds), max(ds)))
    tot = sum(r[0] for r in rows)
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_us':>10s} {'%':>6s}")
    for t, name, n, avg, mn, mx in sorted(rows, reverse=True):
        short = name if len(name) <= 72 else name[:69] + "..."
        print(f"{short:72s} {n:6d} {avg:9.2f} {mn:9.2f} {mx:9.2f} {t:10.1f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
