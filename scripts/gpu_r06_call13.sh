#!/bin/bash
# round 6, call 13: counters of the D = 256 ExpectedAttention quadratic-form kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
cat > /tmp/ea_big_prof.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from kvpress_amd import _native as N
dev = torch.device("cuda", 0)
D, Hkv, S, Hq = 256, 8, 32768, 32
k = torch.randn((1, Hkv, S, D), device=dev).bfloat16(); v = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
mu = torch.randn((1, Hq, D), device=dev) * 0.3
a = torch.randn((1, Hq, D, D), device=dev) * 0.04
cov = a @ a.transpose(-1, -2)
for _ in range(4):
    sc = N.ea_score(k, v, mu, cov, 4, True, 0.0)
torch.cuda.synchronize()
PY
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_big_$i -o pmc -- python /tmp/ea_big_prof.py > /tmp/pmc_big_$i.log 2>&1; echo "pmc $i rc=$?"
done
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python scripts/rocpd_pmc.py $(find /tmp/pmc_big_[0-9] -name '*.db' | sort) 2>&1 | grep -A14 "big_kernel" | head -40
