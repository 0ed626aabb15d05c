#!/bin/bash
# round 6: the committed evidence of the final sources -- GPU suite, bench lines (BASELINE configs + f-rows + batch > 1), rocprofv3 kernel
# statistics, PMC summaries, clock / power record (incl. the ExpectedAttention kernels), end-to-end prefill
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ROUND_TAG=r06 PROF_WL="snapkv128k_b2 chunk_snapkv128k" PMC_WL="snapkv128k knorm32k ea128k"
bash scripts/gpu_check.sh tests bench frows prof pmc e2e power
