#!/bin/bash
# round 6, call 14: the GPU suite of the final sources (after the shared-GPU bench mode moved to the multi-launch select) + that test five more times + the widened gpu_fuzz
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -q --no-header > gpurun_out/r06_gpu_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r06_gpu_tests.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r06_gpu_tests.log | head -10
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --no-header -k "two_ranks_real_kernels" 2>&1 | tail -1; done
timeout 1500 python tools/gpu_fuzz.py --rounds 32 --seed 3 > gpurun_out/c14_gpu_fuzz.txt 2>&1; echo "gpu_fuzz rc=$? $(tail -1 gpurun_out/c14_gpu_fuzz.txt)"; grep -c qstats gpurun_out/c14_gpu_fuzz.txt
