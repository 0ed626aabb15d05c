"""Compile-time resource audit (no GPU needed): no kernel may spill or use scratch memory.
(A by-reference staging struct once ended up in scratch and silently serialised the K pipeline.)"""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

from kvpress_amd import build as B


def _remarks(src):
    cmd = [B._hipcc(), *B.flags_for(src), "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


def test_no_scratch_no_spills():
    with ThreadPoolExecutor(max_workers=8) as ex:
        outs = list(ex.map(_remarks, B.sources()))
    nkern = 0
    for src, text in zip(B.sources(), outs):
        name = None
        for line in text.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                nkern += 1
            for key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m:   # every kernel of the library is hand-written (no vendor template kernels since round 5): no exemptions
                    assert int(m.group(1)) == 0, f"{os.path.basename(src)}: {name}: {key} = {m.group(1)}"
    assert nkern >= 20


def test_no_vendor_sort_or_scan_library():
    """the score-order sort is a hand-written kernel -- no rocPRIM / hipCUB / thrust include anywhere in csrc/."""
    csrc = os.path.dirname(B.sources()[0])
    for f in sorted(os.listdir(csrc)):
        if not os.path.isfile(os.path.join(csrc, f)):
            continue
        text = open(os.path.join(csrc, f)).read()
        for lib in ("rocprim/", "hipcub/", "thrust/", "<cub/"):
            assert "#include <" + lib not in text and '#include "' + lib not in text, f"{f} includes {lib}"
