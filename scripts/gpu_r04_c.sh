#!/bin/bash
# Round 4, GPU session C: the speculative second digit of the cluster select -- parity, phase stamps, A/B inside the bench loops.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster_failure.py -m gpu -q --no-header -x -k "topk or fused or cluster or select or knorm" > gpurun_out/r04_select_tests.log 2>&1
echo "select tests rc=$? $(tail -1 gpurun_out/r04_select_tests.log)"
bash tools/build_variants.sh tc_timing > gpurun_out/variants.log 2>&1; echo "variants rc=$?"
( KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_timing.so timeout 300 python tools/select_lab.py --stamps; KVP_TC_SPEC=0 KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_timing.so timeout 300 python tools/select_lab.py --stamps ) > gpurun_out/r04_select_stamps.txt 2>&1
echo "stamps rc=$?"
timeout 600 python tools/select_lab.py --reps 300 > gpurun_out/r04_select_lab.txt 2>&1; echo "lab rc=$?"
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
ab sk_spec_$rep snapkv128k KVP_TC_SPEC=1
ab sk_nospec_$rep snapkv128k KVP_TC_SPEC=0
ab kn_spec_$rep knorm32k KVP_TC_SPEC=1
ab kn_nospec_$rep knorm32k KVP_TC_SPEC=0
done
ab kn128_spec knorm128k KVP_TC_SPEC=1
ab kn128_nospec knorm128k KVP_TC_SPEC=0
