#!/bin/bash
# round 6, call 4: ExpectedAttention logits kernel after the start-up amortisation / scalar tile requests / packed row-dot: parity tests + bench A/B against HEAD~ is not possible in one tree, so: tests, then ea128k bench with per-kernel events, then rocprofv3 stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests -m gpu -q --no-header -k "ea or qproj or hidden_path or expected" > gpurun_out/r06_gpu_tests_c4.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c4.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c4.log | head -10
for rep in 1 2; do
timeout 600 python bench.py --workload ea128k --steps 20 --warmup 3 --no-cpu-baseline --live-pmc off --profile-json gpurun_out/c4_kernels_ea128k_$rep.json > gpurun_out/c4_bench_ea128k_$rep.log 2>&1
echo "bench[ea128k #$rep] rc=$? $(tail -1 gpurun_out/c4_bench_ea128k_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["parity"], d["roofline"]["path"]["kernels_us"])' 2>&1 | cut -c1-700)"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_ea" -o ea128k -- python "$GRAFT_REPO_ROOT/bench.py" --workload ea128k --steps 20 --warmup 3 --no-cpu-baseline --live-pmc off > "$GRAFT_REPO_ROOT/gpurun_out/prof_ea.log" 2>&1
echo "prof rc=$?"
find "$GRAFT_REPO_ROOT/gpurun_out/prof_ea" -name "*kernel_stats.csv" -exec cp {} "$GRAFT_REPO_ROOT/gpurun_out/c4_rocprofv3_kernel_stats_ea128k.csv" \;
head -8 "$GRAFT_REPO_ROOT/gpurun_out/c4_rocprofv3_kernel_stats_ea128k.csv" | cut -c1-160
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_ea"
