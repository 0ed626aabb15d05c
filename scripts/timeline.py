#!/usr/bin/env python3
"""Per-dispatch timeline of the last steps of a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap to the previous
kernel's end (what the dependent launch chain of one compress call looks like on the device).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 6 --warmup 2 --prewarm-ms 5 --no-cpu-baseline
    python scripts/timeline.py DIR > profiles/rNN_timeline_<workload>.txt
"""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    assert files, "no kernel_trace.csv under " + sys.argv[1]
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last three steps: a step ends with the gather
    ends = [i for i, r in enumerate(rows) if "gather_vec_kernel" in r[2]]
    first = ends[-4] + 1 if len(ends) >= 4 else 0
    t0 = rows[first][0]
    prev_end = None
    print(f"# {files[0].split('/')[-1]}: last three steps; times in us relative to the first listed kernel's start")
    print(f"{'start':>9} {'dur':>8} {'gap':>7}  kernel")
    for s, e, name in rows[first:]:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {short}")
        prev_end = e
        if "gather_vec_kernel" in name:
            print()


if __name__ == "__main__":
    main()
