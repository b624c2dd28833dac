#!/bin/bash
# A/B builds of the FP64 strip layer kernel for C2 (KS = 15): tools/variants64.sh NAME "FLAGS" ...
# -> vsmartmom.jl_amd/lib_dbg/libw_NAME.so (vsm_strip.hip, KS = 15 object, rebuilt with FLAGS)
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
mkdir -p ../lib_dbg
OTHER=$(ls *.o | grep -v '^vsm_strip_15\.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $flags \
      -DVSM_STRIP_KS=15 -Rpass-analysis=kernel-resource-usage -c vsm_strip.hip -o ../lib_dbg/w_$name.o 2> ../lib_dbg/w_$name.log
    hipcc --offload-arch=gfx950 -shared -fPIC $OTHER ../lib_dbg/w_$name.o -o ../lib_dbg/libw_$name.so
    echo "$name: $(grep -A12 'k_layer_stripILi15ELb0' ../lib_dbg/w_$name.log | grep -E 'ScratchSize|VGPRs Spill' | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ')" ) &
done
wait
