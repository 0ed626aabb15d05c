#!/bin/bash
# One GPU-box session: build check, parity tests by group, smoke, bench lines.  Logs -> gpurun_out/.
# usage: scripts/gpu_check.sh [tests] [bench] [frows] [prof] [timeline] [pmc] [e2e] [power]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
WHAT="${*:-tests bench}"
R="${ROUND_TAG:-r06}"
python __graft_entry__.py > gpurun_out/build.log 2>&1; echo "build rc=$?"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
if [[ "$WHAT" == *tests* ]]; then
  timeout 2400 python -m pytest tests -m gpu -q --no-header > gpurun_out/${R}_gpu_tests.log 2>&1
  echo "tests rc=$? $(tail -1 gpurun_out/${R}_gpu_tests.log)"
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log)"
fi
if [[ "$WHAT" == *bench* ]]; then
  for wl in ${BENCH_WL:-knorm32k knorm128k snapkv128k ea128k}; do
    # (the kernel-statistics CSV is written by the bench run itself from the trace pass its per-kernel numbers come from)
    timeout 900 python bench.py --workload $wl --profile-json gpurun_out/${R}_kernels_$wl.json --trace-stats-csv gpurun_out/${R}_rocprofv3_kernel_stats_$wl.csv > gpurun_out/bench_$wl.log 2>&1
    echo "bench[$wl] rc=$? $(tail -1 gpurun_out/bench_$wl.log | cut -c1-300)"
    tail -1 gpurun_out/bench_$wl.log > gpurun_out/${R}_bench_$wl.json
  done
fi
if [[ "$WHAT" == *frows* ]]; then
  # SURVEY section 8(f) workloads (no CPU baseline: the four BASELINE configs above carry it)
  for wl in snapkv128k_scoreorder snapkv128k_b2 knorm128k_b4 keydiff128k cur128k finch128k chunk_snapkv128k rerotate128k decode_snapkv2k; do
    timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --live-pmc off > gpurun_out/bench_$wl.log 2>&1
    echo "bench[$wl] rc=$? $(tail -1 gpurun_out/bench_$wl.log | cut -c1-200)"
    tail -1 gpurun_out/bench_$wl.log > gpurun_out/${R}_bench_$wl.json
  done
fi
if [[ "$WHAT" == *pmc* ]]; then
  # HBM traffic and issue counters, per MI355X_MICROARCH.md: separate --pmc passes, kernel-trace only (no other tracing domains)
  cd /tmp
  for wl in ${PMC_WL:-snapkv128k knorm32k}; do
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
               "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${wl}_$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --steps 3 --warmup 1 --prewarm-ms 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_${wl}_$i.log" 2>&1
      echo "pmc[$wl $i: $set] rc=$?"
    done
    ( cd "$GRAFT_REPO_ROOT"; echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline (four separate passes)";
      echo "# csrc_digest $(python -c 'import bench; print(bench.csrc_digest())')";
      python scripts/rocpd_pmc.py $(find gpurun_out/pmc_${wl}_[0-9] -name '*.db' | sort) ) > "$GRAFT_REPO_ROOT/gpurun_out/${R}_pmc_summary_$wl.txt" 2>&1
    rm -rf "$GRAFT_REPO_ROOT"/gpurun_out/pmc_${wl}_[0-9]
  done
  cd "$GRAFT_REPO_ROOT"
fi
if [[ "$WHAT" == *e2e* ]]; then
  # random-init Llama-3.1-8B body, prefill with / without the press (tools/e2e_prefill.py)
  : > gpurun_out/${R}_e2e_prefill.jsonl
  timeout 900 python tools/e2e_prefill.py --seq-len 131072 --press snapkv --reps 3 2> gpurun_out/e2e.err | tail -1 >> gpurun_out/${R}_e2e_prefill.jsonl; echo "e2e[128k snapkv] rc=$?"
  timeout 600 python tools/e2e_prefill.py --seq-len 32768 --press knorm --reps 5 2>> gpurun_out/e2e.err | tail -1 >> gpurun_out/${R}_e2e_prefill.jsonl; echo "e2e[32k knorm] rc=$?"
fi
if [[ "$WHAT" == *timeline* ]]; then
  cd /tmp
  for wl in ${TL_WL:-snapkv128k knorm32k}; do
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tl_$wl" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --steps 6 --warmup 2 --prewarm-ms 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/tl_$wl.log" 2>&1
    python "$GRAFT_REPO_ROOT/scripts/timeline.py" "$GRAFT_REPO_ROOT/gpurun_out/tl_$wl" > "$GRAFT_REPO_ROOT/gpurun_out/${R}_timeline_$wl.txt" 2>&1
    echo "timeline[$wl] rc=$?"
    rm -rf "$GRAFT_REPO_ROOT/gpurun_out/tl_$wl"
  done
  cd "$GRAFT_REPO_ROOT"
fi
if [[ "$WHAT" == *prof* ]]; then
  cd /tmp
  for wl in ${PROF_WL:-snapkv128k knorm32k knorm128k ea128k}; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$wl" -o $wl -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_$wl.log" 2>&1
    echo "prof[$wl] rc=$?"
    find "$GRAFT_REPO_ROOT/gpurun_out/prof_$wl" -name "*kernel_stats.csv" -exec cp {} "$GRAFT_REPO_ROOT/gpurun_out/${R}_rocprofv3_kernel_stats_$wl.csv" \;
    grep "^{" "$GRAFT_REPO_ROOT/gpurun_out/prof_$wl.log" | tail -1 > "$GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof_$wl.json"
    rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_$wl"
  done
  cd "$GRAFT_REPO_ROOT"
fi
if [[ "$WHAT" == *power* ]]; then
  # socket power / clocks per kernel (amdsmi + in-kernel stamps of a lab build): profiles/rNN_clock_power.txt
  python tools/gen_stage_asm.py ubench > /dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o tools/ubench_stage 2>/dev/null
  python tools/make_clock_lab.py > /dev/null 2>&1; echo "clocklab rc=$?"
  timeout 900 python tools/power_clock_lab.py > gpurun_out/${R}_clock_power_raw.txt 2> gpurun_out/pcl.err; echo "power rc=$?"
fi
