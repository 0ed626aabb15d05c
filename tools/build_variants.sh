#!/bin/bash
# Lab builds of libkvpress_hip.so with ablated asm loops (GEN_ABL of tools/gen_stage_asm.py) -> kvpress_amd/lib/variants/<name>.so
# usage: tools/build_variants.sh "name1=GEN_ABL=abl,abl" "name2=GEN_PF_AHEAD=3 GEN_ABL=..." ...   (run tools/sk_lab.py with KVPRESS_HIP_LIB=kvpress_amd/lib/variants/<name>.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p kvpress_amd/lib/variants
for spec in "$@"; do
  name=${spec%%=*}; abl=${spec#*=}
  env $abl python tools/gen_stage_asm.py kernel > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -c kvpress_amd/csrc/snapkv_mfma.hip -o /tmp/snapkv_mfma_$name.o 2>/dev/null
  objs=$(ls kvpress_amd/build/*.o | grep -v snapkv_mfma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kvpress_amd/lib/variants/$name.so $objs /tmp/snapkv_mfma_$name.o
  echo "built $name ($abl)"
done
python tools/gen_stage_asm.py kernel > /dev/null   # restore the production loops
