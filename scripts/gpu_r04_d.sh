#!/bin/bash
# Round 4, GPU session D: the whole GPU suite on the regenerated fixtures (every column pinned), host overhead of the decode regime,
# f-row workloads with the library window projection.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --no-header -x > gpurun_out/r04_gpu_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r04_gpu_tests.log)"
timeout 300 python tools/host_overhead_probe.py --workload decode_snapkv2k > gpurun_out/r04_host_probe_decode.txt 2>&1; echo "probe rc=$?"
timeout 300 python tools/host_overhead_probe.py --workload knorm32k --calls 1000 > gpurun_out/r04_host_probe_knorm32k.txt 2>&1; echo "probe rc=$?"
for wl in decode_snapkv2k chunk_snapkv128k finch128k; do
  timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-extra --live-pmc off > gpurun_out/bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/bench_$wl.log | cut -c1-200)"
  tail -1 gpurun_out/bench_$wl.log > gpurun_out/r04_bench_$wl.json
done
