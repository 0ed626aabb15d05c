#!/bin/bash
# round 6, call 17 (lab): pass 1 with contiguous tile ranges and a STATIC share skew between even and odd XCDs (KVP_SK_P1_SKEW = 1 + per mille)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
: > gpurun_out/r06_ab_p1_skew.txt
for rep in 1 2; do
  for sk in 0 1 21 31 41 51 61; do
    KVP_SK_P1_SKEW=$sk timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --live-pmc off > gpurun_out/ab_sk_${sk}_$rep.log 2>&1
    echo "KVP_SK_P1_SKEW=$sk #$rep $(grep '^{' gpurun_out/ab_sk_${sk}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "event median", d["step_events"]["median_ms"], "parity", d["parity"]["max_rel_err_scores"], d["parity"]["set_differences_inside_band"], {k:round(v,1) for k,v in d["roofline"]["path"]["kernels_us"].items() if k.startswith("snapkv")})' 2>&1 | cut -c1-300)" | tee -a gpurun_out/r06_ab_p1_skew.txt
  done
done
