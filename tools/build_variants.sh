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
  objs=$(ls kvpress_amd/build/*.o | grep -v 'snapkv_mfma.o\|/contrib_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/$name.so $objs /tmp/snapkv_mfma_$name.o
  echo "built $name ($abl)"
done
python tools/gen_stage_asm.py kernel > /dev/null   # restore the production loops
# tc_timing: the cluster select with phase time stamps (tools/select_lab.py --stamps): the stamps are a PATCH on the production source
# (inserted by tools/make_tc_timing.py), not #ifdefs inside it
if [[ " $* " == *" tc_timing "* ]]; then
  python tools/make_tc_timing.py /tmp/topk_cluster_timing.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ikvpress_amd/csrc -c /tmp/topk_cluster_timing.hip -o /tmp/topk_cluster_timing.o
  objs=$(ls kvpress_amd/build/*.o | grep -v 'topk_cluster.o\|/contrib_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/tc_timing.so $objs /tmp/topk_cluster_timing.o
  echo "built tc_timing"
fi
# (round 3's other cluster-select lab variants -- XCD-local traffic, 8 steps in flight / non-temporal loads in the Knorm stream -- were
# measured and dropped: profiles/r03_select_cluster_lab.txt, r03_stream_lab.txt; round 4's: tools/lab_patches/tc_speculation.diff)
