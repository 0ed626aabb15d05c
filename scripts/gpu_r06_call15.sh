#!/bin/bash
# round 6, call 15: ExpectedAttention's U fragments packed once per head (ea_tri_pack_kernel) -- A/B on one box, EA tests, fixture parity
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --no-header -x -k "ea_ or expected or config4" 2>&1 | tail -2
: > gpurun_out/r06_ab_ea_pack.txt
for rep in 1 2 3; do
  for pk in 0 1; do
    KVP_EA_PACK_LAB=$pk timeout 600 python bench.py --workload ea128k --steps 30 --warmup 5 --no-cpu-baseline --no-extra --live-pmc off > gpurun_out/ab_pack_${pk}_$rep.log 2>&1
    echo "KVP_EA_PACK_LAB=$pk #$rep $(grep '^{' gpurun_out/ab_pack_${pk}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "parity", d["parity"]["max_rel_err_scores"], d["parity"]["set_differences_inside_band"], {k:round(v,1) for k,v in d["roofline"]["path"]["kernels_us"].items()}, d["roofline"].get("timing_source","")[:40])' 2>&1 | cut -c1-420)" | tee -a gpurun_out/r06_ab_ea_pack.txt
  done
done
