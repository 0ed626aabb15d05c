#!/bin/bash
# Round 4, GPU session B: qproj variants (1 = round-1 kernel, 2 = LDS-DMA loaders, 3 = register-staged loaders) against the model's
# own GEMM + RoPE launch; the device-wide sort of the score-order mode; the kept-order tests in both modes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_kept_order_reference.py tests/test_gpu_cluster_failure.py -m gpu -q --no-header -s > gpurun_out/r04_new_tests.log 2>&1
echo "new tests rc=$? $(tail -1 gpurun_out/r04_new_tests.log)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k "qproj or hidden_path or order" > gpurun_out/r04_qproj_tests.log 2>&1
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
for rep in 1 2; do
ab base_$rep snapkv128k KVP_LIBRARY_QPROJ=0
ab qp1_$rep snapkv128k KVP_LIBRARY_QPROJ=1 KVP_QP_VARIANT=1
ab qp3_$rep snapkv128k KVP_LIBRARY_QPROJ=1 KVP_QP_VARIANT=3
done
ab order snapkv128k_scoreorder KVP_LIBRARY_QPROJ=0
