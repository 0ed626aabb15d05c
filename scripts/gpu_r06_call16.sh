#!/bin/bash
# round 6, call 16: a longer fuzz campaign on the final sources (new seeds)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
{
  echo "# csrc_digest $(python -c 'import bench; print(bench.csrc_digest())')"
  for seed in ${SHAPE_SEEDS:-101 102 103 104 105 106}; do echo "== tools/snapkv_shape_fuzz.py --rounds 200 --seed $seed"; timeout 1200 python tools/snapkv_shape_fuzz.py --rounds 200 --seed $seed 2>&1 | grep -E "MISMATCH|fuzz|Traceback|Error" | tail -8; done
  for seed in ${SELECT_SEEDS:-11 12 13}; do echo "== tools/select_fuzz.py --rounds 80 --seed $seed"; timeout 1500 python tools/select_fuzz.py --rounds 80 --seed $seed 2>&1 | grep -E "MISMATCH|fuzz ok|Traceback|Error|assert" | tail -5; done
  for seed in ${SCORE_SEEDS:-21 22 23 24}; do echo "== tools/gpu_fuzz.py --rounds 40 --seed $seed"; timeout 1500 python tools/gpu_fuzz.py --rounds 40 --seed $seed 2>&1 | grep -E "fuzz ok|Traceback|Error|assert" | tail -5; done
} > gpurun_out/r06_gpu_fuzz_long.txt 2>&1
cat gpurun_out/r06_gpu_fuzz_long.txt | cut -c1-200
