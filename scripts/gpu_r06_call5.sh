#!/bin/bash
# round 6, call 5: full GPU suite of the tree + shape sweep (performance evidence outside the benchmark's shape)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1500 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests_c5.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests_c5.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests_c5.log | head -10
timeout 900 python tools/shape_sweep.py > gpurun_out/r06_shape_sweep.txt 2> gpurun_out/sweep.err; echo "sweep rc=$?"; cat gpurun_out/r06_shape_sweep.txt; tail -3 gpurun_out/sweep.err
