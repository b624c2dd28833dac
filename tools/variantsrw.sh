#!/bin/bash
# A/B builds of the wave-per-line Raman kernels: tools/variantsrw.sh NAME "FLAGS" [NAME "FLAGS" ...]
# -> vsmartmom.jl_amd/lib_dbg/libv_NAME.so (vsm_raman_wave.hip rebuilt with FLAGS, every other object as in the main build)
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
mkdir -p ../lib_dbg
OTHER=$(ls *.o | grep -v '^vsm_raman_wave')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function $flags \
      -c vsm_raman_wave.hip -o ../lib_dbg/v_$name.o 2> ../lib_dbg/v_$name.log
    hipcc --offload-arch=gfx950 -shared -fPIC $OTHER ../lib_dbg/v_$name.o -o ../lib_dbg/libv_$name.so
    echo "$name built" ) &
done
wait
