#!/bin/bash
# round 5, two-hop cluster select: GPU suite + bench lines of the workloads whose select it is
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
WHAT="${1:-tests bench}"
if [[ "$WHAT" == *alltests* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/th_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/th_tests.log)"
fi
if [[ "$WHAT" == *selonly* ]]; then
  timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "topk or select or cluster or knorm or compress" > gpurun_out/th_seltests.log 2>&1; echo "seltests rc=$? $(tail -1 gpurun_out/th_seltests.log)"
fi
if [[ "$WHAT" == *bench* ]]; then
  for wl in ${TH_WL:-snapkv128k knorm32k knorm128k}; do
    for rep in 1 2; do
      timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --live-pmc off > gpurun_out/th_bench_${wl}_$rep.log 2>&1
      echo "bench[$wl#$rep] rc=$? $(grep '^{' gpurun_out/th_bench_${wl}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"].get("path_frac"), d["roofline"]["path"]["kernels_us"] if "path" in d["roofline"] else "")' 2>&1 | cut -c1-300)"
    done
  done
fi
if [[ "$WHAT" == *ab* ]]; then
  # A/B on ONE box: the committed select (variants/base.so = HEAD's sources) against this tree's, alternating
  for wl in ${TH_WL:-snapkv128k knorm32k knorm128k}; do
    for rep in 1 2 3; do
      for var in base new; do
        if [[ $var == base ]]; then export KVPRESS_HIP_LIB=$PWD/kvpress_amd/lib/variants/base.so; else unset KVPRESS_HIP_LIB; fi
        timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --live-pmc off > gpurun_out/th_ab_${wl}_${var}_$rep.log 2>&1
        echo "ab[$wl $var #$rep] rc=$? $(grep '^{' gpurun_out/th_ab_${wl}_${var}_$rep.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["step_events"]["median_ms"], d["roofline"]["path"]["kernels_us"].get("topk_cluster_kernel"))' 2>&1 | cut -c1-200)"
      done
    done
  done
  unset KVPRESS_HIP_LIB
fi
