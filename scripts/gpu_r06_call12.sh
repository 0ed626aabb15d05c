cat > /tmp/ea_big_prof.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from kvpress_amd import _native as N
dev = torch.device("cuda", 0)
D, Hkv, S, Hq = 256, 8, 32768, 32
k = torch.randn((1, Hkv, S, D), device=dev).bfloat16(); v = torch.randn((1, Hkv, S, D), device=dev).bfloat16()
q = torch.randn((1, S - 4, Hq * D), device=dev).bfloat16().view(1, S - 4, Hq, D).transpose(1, 2)
for _ in range(12):
    mu, cov = N.ea_qstats(q, True)
    sc = N.ea_score(k, v, mu, cov, 4, True, 0.0)
torch.cuda.synchronize()
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ea_big -o ea_big -- python /tmp/ea_big_prof.py > /tmp/prof_ea_big.log 2>&1; echo "prof rc=$?"
f=$(find /tmp/prof_ea_big -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {r["Calls"]:>4s}  {r["Name"][:100]}')
PY
cp "$f" /root/repo/gpurun_out/r06_kernel_stats_ea_d256.csv
