#!/bin/bash
# A/B builds of the linearized strip kernels (KS = 15): tools/variantslin.sh NAME "FLAGS" ... -> lib_dbg/libl_NAME.so
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
mkdir -p ../lib_dbg
OTHER=$(ls *.o | grep -v '^vsm_striplin_15\.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function $flags -DVSM_STRIP_KS=15 \
      -Rpass-analysis=kernel-resource-usage -c vsm_striplin.hip -o ../lib_dbg/l_$name.o 2> ../lib_dbg/l_$name.log
    hipcc --offload-arch=gfx950 -shared -fPIC $OTHER ../lib_dbg/l_$name.o -o ../lib_dbg/libl_$name.so
    echo "$name: $(grep -A12 'k_dbl_lin_step' ../lib_dbg/l_$name.log | grep -E 'ScratchSize|VGPRs:|AGPRs' | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ' | cut -c1-200)" ) &
done
wait
