#!/bin/bash
# A/B of tuning knobs INSIDE the bench loop (the state the kernels really meet: what the previous step left in the caches).
# usage: scripts/ab_bench.sh <workload> "<ENV=..> <ENV=..>" "<other config>" ...   -> one line per config: ms/step + per-kernel us
cd "${GRAFT_REPO_ROOT:-/root/repo}"
wl=$1; shift
for cfg in "$@"; do
  out=$(env $cfg timeout 300 python bench.py --workload $wl --steps ${AB_STEPS:-50} --warmup 5 --no-cpu-baseline --live-pmc off 2>/dev/null | tail -1)
  python - "$wl" "$cfg" "$out" <<'PY'
import json, sys
wl, cfg, out = sys.argv[1:4]
try:
    d = json.loads(out)
    k = d["roofline"]["path"]["kernels_us"]
    print(f"{wl:18s} [{cfg:40s}] {d['ms_per_step']*1e3:7.1f} us  " + " ".join(f"{n.replace('_kernel','')}={v:.1f}" for n, v in k.items()))
except Exception as e:
    print(wl, cfg, "FAILED", e, out[-300:])
PY
done
