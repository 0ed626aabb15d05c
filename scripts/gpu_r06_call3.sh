#!/bin/bash
# round 6, call 3: GPU suite; A/B of pass 2's tile shares from pass 1's clock (KVP_SK_BALANCE) on one box; batch > 1 workloads after the
# gather-grid fix and the library projection for two elements
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests_c3.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c3.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c3.log | head -10
: > gpurun_out/r06_ab_balance.txt
for rep in 1 2 3; do
  for bal in 0 1; do
    KVP_SK_BALANCE=$bal timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --live-pmc off > gpurun_out/ab_bal_${bal}_$rep.log 2>&1
    echo "KVP_SK_BALANCE=$bal #$rep $(grep '^{' gpurun_out/ab_bal_${bal}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "event median", d["step_events"]["median_ms"], "parity", d["parity"]["max_rel_err_scores"], d["parity"]["set_differences_inside_band"], {k:round(v,1) for k,v in d["roofline"]["path"]["kernels_us_events"].items()})' 2>&1 | cut -c1-400)" | tee -a gpurun_out/r06_ab_balance.txt
  done
done
for wl in snapkv128k_b2 knorm128k_b4 knorm32k; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off --profile-json gpurun_out/c3_kernels_$wl.json > gpurun_out/c3_bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/c3_bench_$wl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["path_frac"], d["roofline"]["path"]["kernels_us"])' 2>&1 | cut -c1-600)"
done
