#!/bin/bash
# round 5, call 1: clock / power evidence + baseline bench lines of this round's box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
(rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -60) > gpurun_out/smi_idle.txt
(amd-smi static --limit 2>&1 | head -60) >> gpurun_out/smi_idle.txt
timeout 900 python tools/power_clock_lab.py > gpurun_out/r05_clock_power.txt 2> gpurun_out/pcl.err; echo "pcl rc=$? lines=$(wc -l < gpurun_out/r05_clock_power.txt)"
cp /tmp/pcl_samples.jsonl gpurun_out/pcl_samples.jsonl 2>/dev/null
for wl in snapkv128k snapkv128k_scoreorder knorm32k; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off --profile-json gpurun_out/c1_kernels_$wl.json > gpurun_out/c1_bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/c1_bench_$wl.log | cut -c1-260)"
done
