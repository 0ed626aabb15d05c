#!/bin/bash
# Lab builds of libkvpress_hip.so with ablated asm loops (GEN_ABL of tools/gen_stage_asm.py) -> kvpress_amd/lib/variants/<name>.so
# usage: tools/build_variants.sh "name1=GEN_ABL=abl,abl" "name2=GEN_PF_AHEAD=3 GEN_ABL=..." ...   (run tools/sk_lab.py with KVPRESS_HIP_LIB=kvpress_amd/lib/variants/<name>.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p kvpress_amd/lib/variants
for spec in "$@"; do
  [[ "$spec" == tc_* ]] && continue
  name=${spec%%=*}; abl=${spec#*=}
  env $abl python tools/gen_stage_asm.py kernel > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $(env $abl bash -c 'echo ${KVP_VARIANT_CFLAGS:-}') -c kvpress_amd/csrc/snapkv_mfma.hip -o /tmp/snapkv_mfma_$name.o 2>/tmp/snapkv_mfma_$name.err || { cat /tmp/snapkv_mfma_$name.err; exit 1; }
  objs=$(ls kvpress_amd/build/*.o | grep -v snapkv_mfma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/$name.so $objs /tmp/snapkv_mfma_$name.o
  echo "built $name ($abl)"
done
python tools/gen_stage_asm.py kernel > /dev/null   # restore the production loops
# tc_timing: the cluster select with phase time stamps (tools/select_lab.py --stamps)
if [[ " $* " == *" tc_timing "* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DKVP_TC_TIMING -c kvpress_amd/csrc/topk_cluster.hip -o /tmp/topk_cluster_timing.o
  objs=$(ls kvpress_amd/build/*.o | grep -v topk_cluster.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/tc_timing.so $objs /tmp/topk_cluster_timing.o
  echo "built tc_timing"
fi
# tc_l2 / tc_l2_timing: the cluster select on XCD-local (placement-dependent) traffic -- lab measurement only
for v in tc_l2 tc_l2_timing; do
  if [[ " $* " == *" $v "* ]]; then
    fl="-DKVP_TC_L2LOCAL"; [[ $v == *timing ]] && fl="$fl -DKVP_TC_TIMING"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl -c kvpress_amd/csrc/topk_cluster.hip -o /tmp/topk_cluster_$v.o
    objs=$(ls kvpress_amd/build/*.o | grep -v topk_cluster.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/$v.so $objs /tmp/topk_cluster_$v.o
    echo "built $v"
  fi
done
# tc_kn8: eight instead of four 64-row steps of the cluster select's Knorm stream in flight
if [[ " $* " == *" tc_kn8 "* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTC_KN_UNROLL=8 -c kvpress_amd/csrc/topk_cluster.hip -o /tmp/topk_cluster_kn8.o
  objs=$(ls kvpress_amd/build/*.o | grep -v topk_cluster.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/tc_kn8.so $objs /tmp/topk_cluster_kn8.o
  echo "built tc_kn8"
fi
# tc_knnt: non-temporal loads in the cluster select's Knorm stream
if [[ " $* " == *" tc_knnt "* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTC_KN_NT -c kvpress_amd/csrc/topk_cluster.hip -o /tmp/topk_cluster_knnt.o
  objs=$(ls kvpress_amd/build/*.o | grep -v topk_cluster.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/tc_knnt.so $objs /tmp/topk_cluster_knnt.o
  echo "built tc_knnt"
fi
