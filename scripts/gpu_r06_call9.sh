#!/bin/bash
# round 6, call 9: ExpectedAttention for head sizes 64 / 96 on the matrix cores: new tests, EA tests, extended fuzzers, shape sweep, full suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -x -k "ea_" > gpurun_out/c9_ea_tests.log 2>&1; echo "ea tests rc=$? $(tail -1 gpurun_out/c9_ea_tests.log)"; grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/c9_ea_tests.log | head -20
timeout 900 python tools/shape_sweep.py 2> gpurun_out/sweep.err | grep "^ea" | cut -c1-220
{
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 6"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 6 2>&1 | tail -125
  echo "== tools/snapkv_shape_fuzz.py --rounds 120 --seed 61"; timeout 900 python tools/snapkv_shape_fuzz.py --rounds 120 --seed 61 2>&1 | tail -125
  echo "== tools/select_fuzz.py"; timeout 1500 python tools/select_fuzz.py 2>&1 | tail -64
  echo "== tools/gpu_fuzz.py"; timeout 1500 python tools/gpu_fuzz.py 2>&1 | tail -30
} > gpurun_out/r06_gpu_fuzz.txt 2>&1
grep -c "^round" gpurun_out/r06_gpu_fuzz.txt; grep -E "MISMATCH|Traceback|Error|fuzz ok|rounds ok" gpurun_out/r06_gpu_fuzz.txt | head -12
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests_c9.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c9.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c9.log | head -10
