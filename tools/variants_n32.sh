#!/bin/bash
# A/B builds of the FP32 native layer kernels (vsm_native32.hip, every KS object): tools/variants_n32.sh NAME "FLAGS" [NAME "FLAGS" ...]
# -> vsmartmom.jl_amd/lib_dbg/libn32_NAME.so; run with VSM_LIB_PATH=vsmartmom.jl_amd/lib_dbg/libn32_NAME.so
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
mkdir -p ../lib_dbg
OTHER=$(ls *.o | grep -v '^vsm_native32_[0-9]*\.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p ../lib_dbg/n32_$name
  for k in $(seq 1 24); do
    echo "hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $flags -DVSM_NATIVE32_KS=$k -c vsm_native32.hip -o ../lib_dbg/n32_$name/k$k.o"
  done | xargs -P 8 -I{} sh -c "{}"
  hipcc --offload-arch=gfx950 -shared -fPIC $OTHER ../lib_dbg/n32_$name/k*.o -o ../lib_dbg/libn32_$name.so
  echo "$name: built"
done
