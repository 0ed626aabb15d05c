#!/usr/bin/env python3
"""Table rows of profiles/rNN_clock_power.txt from the raw record of tools/power_clock_lab.py (JSON lines):
    python tools/format_clock_power.py profiles/r06_clock_power_raw.txt
Measurement aid, not part of the product."""
import json
import sys


def main():
    for line in open(sys.argv[1]):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        smi = d.get("smi") or {}
        name = d.get("phase", "?")
        lib = d.get("lib", "")
        us = f"{d['us_per_iter']:7.1f} us/iter" if "us_per_iter" in d else " " * 15
        pw = f"{smi.get('power_w_mean', float('nan')):6.1f} / {smi.get('power_w_max', 0):.0f} W"
        clk = f"smi {smi.get('smi_gfxclk_mhz_mean', float('nan')):6.1f} MHz"
        umc = f"umc {smi.get('umc_activity', float('nan')):4.1f} %"
        extra = []
        for k in ("p1_asm", "p2_asm", "gather"):
            if k in d:
                s = d[k]
                per = s.get("mhz_per_xcc", {})
                extra.append(f"{k} {s['mhz_mean']:.0f} MHz, workgroup {s['wg_us_mean']:.1f} us, launch span {s.get('launch_span_us', float('nan')):.1f} us, per XCC {[round(per[x]) for x in sorted(per, key=int)]}")
        if "in_kernel_mhz" in d:
            extra.append(f"in-kernel {d['in_kernel_mhz']:.1f} MHz (sustained), {d['ns_per_stage_sustained']:.1f} ns per stage sustained")
        print(f"{name:44s} [{lib:10s}] {us} | {pw} | {clk} | {umc} | " + "; ".join(extra))


if __name__ == "__main__":
    main()
