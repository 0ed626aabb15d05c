#!/bin/bash
# round 5 LAB R5.7 (apply tools/lab_patches/p2_fused_combine.diff first: the product has no KVP_SK_FUSE_COMBINE): pass 2 folds pass 1's partials itself (no softmax_combine launch): parity + A/B through the knob on one box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "snapkv or finch or pyramid or pipeline" > gpurun_out/fu_tests.log 2>&1; echo "snapkv tests rc=$? $(tail -1 gpurun_out/fu_tests.log)"
for rep in 1 2 3; do
  for var in 0 1; do
    KVP_SK_FUSE_COMBINE=$var timeout 300 python bench.py --workload snapkv128k --steps 200 --warmup 20 --no-cpu-baseline --live-pmc off > gpurun_out/fu_ab_${var}_$rep.log 2>&1
    echo "ab[fuse=$var #$rep] rc=$? $(grep '^{' gpurun_out/fu_ab_${var}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["step_events"]["median_ms"], d["parity"]["max_rel_err_scores"], {k:round(v,1) for k,v in d["roofline"]["path"]["kernels_us"].items()})' 2>&1 | cut -c1-300)"
  done
done
