#!/bin/bash
# round 6, call 1: CU-masked overlap probe + baseline lines of this round's box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
timeout 600 python tools/cumask_overlap_lab.py > gpurun_out/r06_cumask_overlap.txt 2> gpurun_out/cumask.err; echo "cumask rc=$?"; cat gpurun_out/r06_cumask_overlap.txt; tail -3 gpurun_out/cumask.err
for wl in snapkv128k knorm32k; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off --profile-json gpurun_out/c1_kernels_$wl.json > gpurun_out/c1_bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/c1_bench_$wl.log | cut -c1-400)"
done
timeout 1200 python -m pytest tests -m gpu -q --no-header -x > gpurun_out/r06_gpu_tests_c1.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c1.log)"
