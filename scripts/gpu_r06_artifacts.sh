#!/bin/bash
# round 6: the committed evidence of the final sources -- GPU suite, bench lines (BASELINE configs + f-rows + batch > 1), rocprofv3 kernel
# statistics, PMC summaries, clock / power record (incl. the ExpectedAttention kernels), end-to-end prefill
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ROUND_TAG=r06 PROF_WL="snapkv128k_b2 chunk_snapkv128k" PMC_WL="snapkv128k knorm32k ea128k"
bash scripts/gpu_check.sh tests bench frows prof pmc e2e power
# shapes outside the benchmark's + the fuzzers, on the same build
timeout 900 python tools/shape_sweep.py > gpurun_out/r06_shape_sweep.txt 2> gpurun_out/sweep.err; echo "sweep rc=$?"
{
  echo "# csrc_digest $(python -c 'import bench; print(bench.csrc_digest())')"
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 6"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 6 2>&1 | tail -125
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 61"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 61 2>&1 | tail -125
  echo "== tools/select_fuzz.py"; timeout 1500 python tools/select_fuzz.py 2>&1 | tail -64
  echo "== tools/gpu_fuzz.py"; timeout 1500 python tools/gpu_fuzz.py 2>&1 | tail -30
} > gpurun_out/r06_gpu_fuzz.txt 2>&1
echo "fuzz: $(grep -c '^round' gpurun_out/r06_gpu_fuzz.txt) rounds; $(grep -cE 'MISMATCH|Traceback|Error' gpurun_out/r06_gpu_fuzz.txt) problems"
