#!/bin/bash
# A/B builds of the FP32 strip kernels: tools/variants32.sh NAME "FLAGS" [NAME "FLAGS" ...]
# -> vsmartmom.jl_amd/lib_dbg/libv_NAME.so (vsm_strip32.hip rebuilt with FLAGS, every other object as in the main build)
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
mkdir -p ../lib_dbg
OTHER=$(ls *.o | grep -v '^vsm_strip32\.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $flags \
      -Rpass-analysis=kernel-resource-usage -c vsm_strip32.hip -o ../lib_dbg/v_$name.o 2> ../lib_dbg/v_$name.log
    hipcc --offload-arch=gfx950 -shared -fPIC $OTHER ../lib_dbg/v_$name.o -o ../lib_dbg/libv_$name.so
    echo "$name: $(grep -A12 'k_layer_strip32ILi6ELb0ELb1' ../lib_dbg/v_$name.log | grep -E 'ScratchSize|VGPRs Spill' | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ')" ) &
done
wait
