#!/bin/bash
# A/B build: the linearized doubling kernel of vsm_strip128lin.hip taking over from N > DBL_MIN instead of 60 and up to
# N <= SMALL_MAX instead of 48 -> lib_ab/lin128_DBLMIN_SMALLMAX.so  (vsm_striplin.hip's k_dbl_lin_multi takes the rest of 8 <= N <= 60)
#   tools/variants_lin128.sh 60 48
set -e
cd "$(dirname "$0")/../vsmartmom.jl_amd/csrc"
DMIN=${1:-60}
SMAX=${2:-48}
MIN=${DMIN}_${SMAX}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DVSM_LIN128_DBL_MIN=$DMIN -DVSM_LIN128_SMALL_MAX=$SMAX"
mkdir -p ../lib_ab/obj
hipcc $FL -c vsm_strip128lin.hip -o ../lib_ab/obj/vsm_strip128lin.o &
hipcc $FL -c vsm_lin.hip -o ../lib_ab/obj/vsm_lin.o &
wait
OBJS=$(ls *.o | grep -v -e '^vsm_strip128lin.o$' -e '^vsm_lin.o$')
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib_ab/obj/vsm_strip128lin.o ../lib_ab/obj/vsm_lin.o -o ../lib_ab/lin128_$MIN.so
ls -la ../lib_ab/lin128_$MIN.so
