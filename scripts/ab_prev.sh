#!/bin/bash
# A/B on ONE box: this tree against the tree of the previous commit (ab_prev/, untracked), alternating runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
for wl in ${AB_WL:-snapkv128k decode_snapkv2k}; do
  for rep in 1 2 3; do
    for side in new prev; do
      d=.; [ $side = prev ] && d=ab_prev
      r=$(cd $d && timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-extra --live-pmc off 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "ab[$wl] rep $rep $side $r"
    done
  done
done
