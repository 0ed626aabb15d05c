#!/usr/bin/env python3
"""Lab build of the cluster select with phase time stamps (tools/select_lab.py --stamps): writes a copy of
kvpress_amd/csrc/topk_cluster.hip with TC_STAMP(i) calls inserted at its phase boundaries.  The production source carries no lab
code; this script is the patch (anchored on source lines, it fails loudly when they move).

    python tools/make_tc_timing.py /tmp/topk_cluster_timing.hip
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MACRO = '''#define TC_STAMP(i) do { if (threadIdx.x == 0 && cluster == 0) TCW(bar)[TC_MAXC * 32 + 32 + slot * 32 + (i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
    TC_STAMP(0);
'''
# (anchor line fragment, stamp index, "before" | "after"); the two-hop form (round 5).  Stamp 11 = 1 if the two-hop form finished the
# select (0: it declined -- threshold outside the sample bracket or a full candidate record -- and the three rounds ran).
ANCHORS = [
    ("        const bool full = kmin != 0u;", 1, "after"),                                       # keys loaded
    ("            uint32_t* whist = fh + 2 * TC_NS;   // (zeroed by window_setup)", 2, "before"),  # (the window is set up inside the key phase)
    ("            // complete when the histogram's total is the row's S keys", 3, "before"),     # window histogram flushed
    ("            d1 = tc_uni(d1);", 4, "before"),                                               # window digit found (poll complete)
    ("                cluster_barrier(cs, 2, &s_fail[0], true);", 5, "before"),                  # candidate record written
    ("                cluster_barrier(cs, 2, &s_fail[0], true);", 6, "after"),
    ("                const bool ovf = ", 7, "before"),                                          # the row's records in LDS
    ("                    T = Tpre;", 8, "before"),                                              # threshold known (local rounds)
    ("        // ---- ordered compaction: keys > T", 9, "before"),                               # offsets of the earlier slots
]


def main(out):
    lines = open(os.path.join(ROOT, "kvpress_amd", "csrc", "topk_cluster.hip")).read().split("\n")
    res, used = [], set()
    for ln in lines:
        hits = [(i, pos) for frag, i, pos in ANCHORS if ln.startswith(frag)]
        for i, pos in hits:
            if pos == "before":
                res.append(f"        TC_STAMP({i});")
                used.add(i)
        res.append(ln)
        if ln.startswith("    if (row >= a.R) return;"):
            res.append(MACRO.rstrip("\n"))
            used.add(0)
        for i, pos in hits:
            if pos == "after":
                res.append(f"        TC_STAMP({i});")
                used.add(i)
    # stamp 12: the end of the kernel body = the closing of the compaction's final store loop
    text = "\n".join(res)
    tail = "            if (rank0 + i < k) out[rank0 + i] = ob[i];\n"
    assert text.count(tail) == 1, "anchor for the final stamp moved"
    text = text.replace(tail, tail + "        TC_STAMP(10);\n        if (threadIdx.x == 0 && cluster == 0) TCW(bar)[TC_MAXC * 32 + 32 + slot * 32 + 11] = done ? 1u : 0u;\n")
    used.add(10)
    missing = sorted(set(range(11)) - used)
    assert not missing, f"anchors moved: stamps {missing} not placed"
    open(out, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/topk_cluster_timing.hip")
