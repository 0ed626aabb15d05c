#!/usr/bin/env python3
"""Does the matrix pipe draw less when consecutive MFMAs share an operand?  (round 5, LAB R5.5)

The SnapKV passes are power-limited (LAB R5.1): time = energy / budget, and the MFMAs are the largest term (324 of ~620 uJ per
stage).  In the production loops consecutive matrix instructions never share an operand (k-step inner: both the K and the Q
fragment change every instruction).  This lab runs MFMA-only loops with the same 96 instructions per iteration and different
reuse patterns (tools/gen_stage_asm.py, `mfma_reuse_body`) on 256 workgroups with random operands for PCL_LOOP_S seconds each,
with tools/smi_sampler.py beside them: if `bquad` / `aquad` run measurably faster than `mfma_only` at the same socket power, a
k-step-outer / sub-tile-inner order of the stage is worth building.

    python tools/operand_reuse_lab.py > gpurun_out/r05_operand_reuse.txt
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import power_clock_lab as pcl  # noqa: E402

VARIANTS = os.environ.get("ORL_VARIANTS", "mfma_only,mfma_bfix,mfma_afix,mfma_abfix,mfma_bpair,mfma_bquad,mfma_apair,mfma_aquad,mfma_only").split(",")


def main():
    sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "smi_sampler.py"), pcl.SAMPLES, "0.02"])
    time.sleep(1.0)
    recs = []
    try:
        loop_ms = int(pcl.LOOP_S * 1000)
        for mode in (1, 0):
            for v in VARIANTS:
                time.sleep(0.4)
                try:
                    recs.append(pcl.ubench(v, mode, loop_ms if mode else 300))
                except Exception as e:  # noqa: BLE001
                    recs.append({"phase": f"ubench:{v}", "error": repr(e)})
    finally:
        time.sleep(0.3)
        sampler.terminate()
        sampler.wait()
    samples = []
    for line in open(pcl.SAMPLES):
        try:
            s = json.loads(line)
        except Exception:  # noqa: BLE001
            continue
        if not s.get("first"):
            samples.append(s)
    print("# tools/operand_reuse_lab.py -- MFMA-only loops, 96 v_mfma_f32_32x32x16_bf16 per iteration, operand reuse between consecutive instructions")
    for rec in recs:
        if "t0" in rec:
            rec["smi"] = pcl.join(rec, samples)
        rec.pop("t0", None)
        rec.pop("t1", None)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
