#!/bin/bash
# Round 4, GPU session G: round 1 of the cluster select with one weighted histogram step per thread -- parity + A/B against the previous
# commit's topk_cluster.hip (built here as kvpress_amd/lib/variants/tc_prev.so from gpurun_out/../tc_prev.hip shipped with the snapshot).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out kvpress_amd/lib/variants
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ikvpress_amd/csrc -c tools/lab_patches/tc_prev.hip -o /tmp/tc_prev.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/tc_prev.so $(ls kvpress_amd/build/*.o | grep -v topk_cluster.o) /tmp/tc_prev.o; echo "variant rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster_failure.py -m gpu -q --no-header -x -k "topk or fused or cluster or select or knorm or timeout or waits or mask" > gpurun_out/r04_select_tests.log 2>&1
echo "select tests rc=$? $(tail -1 gpurun_out/r04_select_tests.log)"
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
ab sk_new_$rep snapkv128k KVP_X=1
ab sk_prev_$rep snapkv128k KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_prev.so
done
ab kn128_new knorm128k KVP_X=1
ab kn128_prev knorm128k KVPRESS_HIP_LIB=kvpress_amd/lib/variants/tc_prev.so
