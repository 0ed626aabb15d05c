#!/bin/bash
# Round 4, GPU session A: the new failure-path / kept-order / qproj tests, then A/B of the window projection and of the
# cluster select's block -> cluster mapping, then one full default bench line.  Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests/test_gpu_cluster_failure.py tests/test_kept_order_reference.py -m gpu -q --no-header -x -s > gpurun_out/r04_new_tests.log 2>&1
echo "new tests rc=$? $(tail -1 gpurun_out/r04_new_tests.log)"
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
ab base snapkv128k KVP_LIBRARY_QPROJ=0
ab qp2 snapkv128k KVP_LIBRARY_QPROJ=1 KVP_QP_VARIANT=2
ab qp1 snapkv128k KVP_LIBRARY_QPROJ=1 KVP_QP_VARIANT=1
ab base2 snapkv128k KVP_LIBRARY_QPROJ=0
ab qp2b snapkv128k KVP_LIBRARY_QPROJ=1 KVP_QP_VARIANT=2
ab il1 snapkv128k KVP_LIBRARY_QPROJ=0 KVP_TC_INTERLEAVE=1
ab kn_il0 knorm32k KVP_TC_INTERLEAVE=0
ab kn_il1 knorm32k KVP_TC_INTERLEAVE=1
ab kn_il0b knorm32k KVP_TC_INTERLEAVE=0
ab order snapkv128k_scoreorder KVP_LIBRARY_QPROJ=0
timeout 900 python -m pytest tests -m gpu -q --no-header -x > gpurun_out/r04_gpu_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r04_gpu_tests.log)"
timeout 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench default rc=$? $(tail -1 gpurun_out/bench_default.log | cut -c1-400)"
