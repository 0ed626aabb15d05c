#!/usr/bin/env python3
"""Per-kernel PMC counter averages from rocprofv3 rocpd databases (`rocprofv3 --kernel-trace --pmc ...`).
Counter values are summed over their instances per dispatch, then averaged over the dispatches of a kernel.

usage: rocpd_pmc.py results.db [results2.db ...] [--filter substring]
"""
import sqlite3
import sys


def load(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    q = """select s.display_name, d.id, d.end - d.start, p.name, sum(e.value)
           from rocpd_kernel_dispatch d
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           join rocpd_pmc_event e on e.event_id = d.event_id
           join rocpd_info_pmc p on e.pmc_id = p.id
           group by d.id, p.name"""
    per = {}
    for name, did, dur, cname, val in cur.execute(q):
        k = per.setdefault(name, {})
        k.setdefault("_dur_us", {})[did] = dur / 1e3
        k.setdefault(cname, {})[did] = val
    return per


def main(argv):
    flt = None
    if "--filter" in argv:
        i = argv.index("--filter")
        flt = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    merged = {}
    for path in argv:
        for name, cs in load(path).items():
            m = merged.setdefault(name, {})
            for c, vals in cs.items():
                v = list(vals.values())
                m[c] = (sum(v) / len(v), len(v))
    for name in sorted(merged, key=lambda n: -merged[n]["_dur_us"][0] * merged[n]["_dur_us"][1]):
        if flt and flt not in name:
            continue
        cs = merged[name]
        print(f"== {name[:110]}  (calls {cs['_dur_us'][1]}, avg {cs['_dur_us'][0]:.1f} us under counters)")
        for c in sorted(cs):
            if c != "_dur_us":
                print(f"     {c:28s} {cs[c][0]:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
