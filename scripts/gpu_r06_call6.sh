#!/bin/bash
# round 6, call 6: windows of any size and head size 64 on the MFMA passes: the new sweep test first, then the whole GPU suite, then the shape sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k "any_window" > gpurun_out/r06_gpu_tests_c6a.log 2>&1; echo "window tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c6a.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c6a.log | head -30; grep -E "max rel err" gpurun_out/r06_gpu_tests_c6a.log | head -12
timeout 1500 python -m pytest tests -m gpu -q --no-header --deselect tests/test_gpu_parity.py::test_snapkv_any_window_on_the_mfma_path > gpurun_out/r06_gpu_tests_c6.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c6.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c6.log | head -20
timeout 900 python tools/shape_sweep.py > gpurun_out/r06_shape_sweep.txt 2> gpurun_out/sweep.err; echo "sweep rc=$?"; cat gpurun_out/r06_shape_sweep.txt | cut -c1-150; tail -3 gpurun_out/sweep.err
