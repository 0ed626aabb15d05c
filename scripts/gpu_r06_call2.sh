#!/bin/bash
# round 6, call 2: CU-mask bit mapping + overlap probe on the mapped layout; window projection for batches; GPU suite of the hygiene commit;
# the new bench line (rocprofv3-timed kernels) and the batch > 1 workloads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/cumask_map.hip -o /tmp/cumask_map 2>/dev/null; timeout 120 /tmp/cumask_map > gpurun_out/r06_cumask_map.txt 2>&1; echo "map rc=$?"; cat gpurun_out/r06_cumask_map.txt
timeout 600 python tools/cumask_overlap_lab.py > gpurun_out/r06_cumask_overlap.txt 2> gpurun_out/cumask.err; echo "cumask rc=$?"; cat gpurun_out/r06_cumask_overlap.txt; tail -3 gpurun_out/cumask.err
timeout 300 python tools/qproj_batch_lab.py > gpurun_out/r06_qproj_batch_lab.txt 2> gpurun_out/qpb.err; echo "qproj lab rc=$?"; cat gpurun_out/r06_qproj_batch_lab.txt; tail -3 gpurun_out/qpb.err
timeout 1500 python -m pytest tests -m gpu -q --no-header -x > gpurun_out/r06_gpu_tests_c2.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c2.log)"; grep -E "FAILED|Error" gpurun_out/r06_gpu_tests_c2.log | head -10
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/c2_bench_head.log 2>&1; echo "bench[head] rc=$? $(tail -1 gpurun_out/c2_bench_head.log | cut -c1-1800)"
for wl in snapkv128k_b2 knorm128k_b4 knorm128k; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off --profile-json gpurun_out/c2_kernels_$wl.json > gpurun_out/c2_bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/c2_bench_$wl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["path_frac"], d["roofline"]["path"]["kernels_us"])' 2>&1 | cut -c1-600)"
done
