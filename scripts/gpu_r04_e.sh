#!/bin/bash
# Round 4, GPU session E: rotated tile walk in the window projection (A/B inside the bench loop) + its parity tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k "qproj or hidden_path" > gpurun_out/r04_qproj_tests.log 2>&1
echo "qproj tests rc=$? $(tail -1 gpurun_out/r04_qproj_tests.log)"
ab() {  # ab <tag> <workload> <env...>
  tag=$1; wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-extra --live-pmc off --profile-json gpurun_out/ab_$tag.json > gpurun_out/ab_$tag.log 2>&1
  echo "ab[$tag] rc=$? $(python - <<PY
import json
try:
    d=json.load(open('gpurun_out/ab_$tag.json'))
    print(round(d['ms_per_step']*1e3,1),'us/step', {k:round(v*1e3*d['launches_per_step'][k],1) for k,v in d['kernels_avg_ms'].items()})
except Exception as e:
    print('no table', e)
PY
)"
}
for rep in 1 2 3; do
ab rot1_$rep snapkv128k KVP_QP_ROTATE=1
ab rot0_$rep snapkv128k KVP_QP_ROTATE=0
done
