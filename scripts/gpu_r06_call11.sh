#!/bin/bash
# round 6, call 11: narrow-head ExpectedAttention at size (streaming statistics) + its per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --no-header -x -s -k "narrow_heads" > gpurun_out/c11_ea_tests.log 2>&1; echo "ea tests rc=$? $(tail -1 gpurun_out/c11_ea_tests.log)"; grep -E "^ea D=|^FAILED|^ERROR|Error|assert" gpurun_out/c11_ea_tests.log | head -20
cat > /tmp/ea_small_prof.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from kvpress_amd import _native as N
dev = torch.device("cuda", 0)
for D, Hkv in ((64, 8), (96, 32)):
    S, Hq = 32768, 32
    k = torch.randn((1, Hkv, S, D), device=dev).bfloat16(); v = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
    q = torch.randn((1, S - 4, Hq * D), device=dev).bfloat16().view(1, S - 4, Hq, D).transpose(1, 2)
    for _ in range(12):
        mu, cov = N.ea_qstats(q, True)
        sc = N.ea_score(k, v, mu, cov, 4, True, 0.0)
    torch.cuda.synchronize()
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ea_small -o ea_small -- python /tmp/ea_small_prof.py > /tmp/prof_ea_small.log 2>&1; echo "prof rc=$?"; cd "${GRAFT_REPO_ROOT:-/root/repo}"
f=$(find /tmp/prof_ea_small -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" gpurun_out/r06_kernel_stats_ea_small_heads.csv; cut -d'"' -f2,3,4,5,6,7,8 "$f" | cut -c1-60,100-400 | head -14; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4s}  {r["Name"][:110]}')
PY
else tail -5 /tmp/prof_ea_small.log; fi
