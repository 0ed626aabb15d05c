#!/usr/bin/env python3
"""Lab build of the cluster select with phase time stamps (tools/select_lab.py --stamps): writes a copy of
kvpress_amd/csrc/topk_cluster.hip with TC_STAMP(i) calls inserted at its phase boundaries.  The production source carries no lab
code; this script is the patch (anchored on source lines, it fails loudly when they move).

    python tools/make_tc_timing.py /tmp/topk_cluster_timing.hip
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MACRO = '''#define TC_STAMP(i) do { if (threadIdx.x == 0 && cluster == 0) a.w.bar[TC_CLUSTERS * 32 + 32 + slot * 16 + (i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
    TC_STAMP(0);
'''
# (anchor line fragment, stamp index, "before" | "after")
ANCHORS = [
    ("        const bool full = kmin != 0u;", 1, "after"),                                   # keys loaded
    ("            if (!poll) cluster_barrier(cs, 1, &s_fail[0], true);", 2, "before"),      # first histogram flushed
    ("            if (!poll) cluster_barrier(cs, 1, &s_fail[0], true);", 3, "after"),       # (counter barrier 1 passed)
    ("        // ---- digit 2: (key >> 8) & 0xFFF among key >> 20 == b1", 4, "before"),      # first digit found
    ("        cluster_barrier(cs, 2, &s_fail[0], HIST1 || poll);", 5, "before"),          # second histogram flushed
    ("        cluster_barrier(cs, 2, &s_fail[0], HIST1 || poll);", 6, "after"),
    ("        const uint32_t prefix = (b1 << 12) | b2;", 7, "after"),                        # second digit found
    ("        cluster_barrier(cs, 3, &s_fail[0], false);", 8, "before"),          # third histogram + suffix table
    ("        cluster_barrier(cs, 3, &s_fail[0], false);", 9, "after"),
    ("        const uint32_t T = (prefix << 8) | b3;", 10, "after"),                          # threshold known
    ("        // ---- ordered compaction: keys > T", 11, "before"),                           # offsets of the earlier slots, histograms zeroed
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
    text = text.replace(tail, tail + "        TC_STAMP(12);\n")
    used.add(12)
    missing = sorted(set(range(13)) - used)
    assert not missing, f"anchors moved: stamps {missing} not placed"
    open(out, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/topk_cluster_timing.hip")
