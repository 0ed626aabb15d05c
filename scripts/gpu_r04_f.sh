#!/bin/bash
# Round 4, GPU session F: the cluster select's polling protocol (a round is complete when its histogram's total says so) against
# the counter barriers: parity, failure path in both, phase stamps, A/B inside the bench loops.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster_failure.py -m gpu -q --no-header -x -k "topk or fused or cluster or select or knorm or timeout or waits or mask" > gpurun_out/r04_select_tests.log 2>&1
echo "select tests rc=$? $(tail -1 gpurun_out/r04_select_tests.log)"
bash tools/build_variants.sh tc_timing > gpurun_out/variants.log 2>&1; echo "variants rc=$?"
( echo "# polling protocol (default)"; KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_timing.so timeout 300 python tools/select_lab.py --stamps; echo "# counter barriers (KVP_TC_POLL=0)"; KVP_TC_POLL=0 KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_timing.so timeout 300 python tools/select_lab.py --stamps ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_select_poll_stamps.txt
echo "stamps rc=$?"
( echo "# polling (default)"; timeout 600 python tools/select_lab.py --reps 300; echo "# KVP_TC_POLL=0"; KVP_TC_POLL=0 timeout 600 python tools/select_lab.py --reps 300 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_select_poll_lab.txt; echo "lab rc=$?"
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
for rep in 1 2; do
ab sk_poll_$rep snapkv128k KVP_TC_POLL=1
ab sk_bar_$rep snapkv128k KVP_TC_POLL=0
ab kn_poll_$rep knorm32k KVP_TC_POLL=1
ab kn_bar_$rep knorm32k KVP_TC_POLL=0
done
ab kn128_poll knorm128k KVP_TC_POLL=1
ab kn128_bar knorm128k KVP_TC_POLL=0
