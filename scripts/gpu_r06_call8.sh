#!/bin/bash
# round 6, call 8: after the signed causal limit (S < padded window): the shape fuzz that found it (more rounds, two seeds), the select / score
# fuzzers, the full GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
{
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 6"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 6 2>&1 | tail -125
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 61"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 61 2>&1 | tail -125
  echo "== tools/select_fuzz.py"; timeout 900 python tools/select_fuzz.py 2>&1 | tail -8
  echo "== tools/gpu_fuzz.py"; timeout 1500 python tools/gpu_fuzz.py 2>&1 | tail -30
} > gpurun_out/r06_gpu_fuzz.txt 2>&1
grep -c "^round" gpurun_out/r06_gpu_fuzz.txt; grep -E "MISMATCH|Traceback|Error|fuzz ok|rounds ok|ok$" gpurun_out/r06_gpu_fuzz.txt | head -12
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests_c8.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c8.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c8.log | head -10
