#!/bin/bash
# round 6, call 7: tile shares computed in pass 2's prologue (combine back to its plain kernel) + head size 96: window / balance / fused tests,
# A/B of KVP_SK_BALANCE on one box, chunk workload after dropping the segmented select's zero-filled workspace, shape sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests_c7.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c7.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c7.log | head -10
: > gpurun_out/r06_ab_balance2.txt
for rep in 1 2 3; do
  for bal in 0 1; do
    KVP_SK_BALANCE=$bal timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --live-pmc off > gpurun_out/ab2_bal_${bal}_$rep.log 2>&1
    echo "KVP_SK_BALANCE=$bal #$rep $(grep '^{' gpurun_out/ab2_bal_${bal}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "event median", d["step_events"]["median_ms"], "parity", d["parity"]["max_rel_err_scores"], d["parity"]["set_differences_inside_band"], {k:round(v,1) for k,v in d["roofline"]["path"]["kernels_us_events"].items()})' 2>&1 | cut -c1-400)" | tee -a gpurun_out/r06_ab_balance2.txt
  done
done
for wl in chunk_snapkv128k snapkv128k_b2; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off > gpurun_out/c7_bench_$wl.log 2>&1
  echo "bench[$wl] rc=$? $(tail -1 gpurun_out/c7_bench_$wl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["path_frac"], d["roofline"]["path"]["kernels_us"])' 2>&1 | cut -c1-600)"
done
timeout 900 python tools/shape_sweep.py > gpurun_out/r06_shape_sweep.txt 2> gpurun_out/sweep.err; echo "sweep rc=$?"; grep "D= 96\|D=256\|D= 64" gpurun_out/r06_shape_sweep.txt | cut -c1-170
