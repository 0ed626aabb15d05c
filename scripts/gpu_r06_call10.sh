#!/bin/bash
# round 6, call 10: statistics of narrow heads (D = 96, D = 64 in other layouts) as zero-padded heads of 128: EA tests, sweep, per-kernel times of D = 64 / 96
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --no-header -x -k "ea_ or expected or config4" > gpurun_out/c10_ea_tests.log 2>&1; echo "ea tests rc=$? $(tail -1 gpurun_out/c10_ea_tests.log)"; grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/c10_ea_tests.log | head -20
timeout 900 python tools/shape_sweep.py 2> gpurun_out/sweep.err > gpurun_out/r06_shape_sweep.txt; grep "^ea" gpurun_out/r06_shape_sweep.txt | cut -c1-200
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
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ea_small -o ea_small -- python /tmp/ea_small_prof.py > /dev/null 2>&1; cd - > /dev/null
f=$(find /tmp/prof_ea_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_kernel_stats_ea_small_heads.csv && cut -d, -f1-4 "$f" | cut -c1-150 | head -14
