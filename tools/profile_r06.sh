#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box through gpurun): stats + PMC passes, one tag per directory under gpurun_out/;
# copy the condensed summary.json / kernel_stats.csv to profiles/r06/<tag>/.  Every summary names the library build it profiled
# (library.source_hash = vsm_build_id(), tools/source_hash.sh <commit> recomputes it from a tree) and the hash of the sources that
# decide what ITS tag measures (tag_sources_hash): a tag whose committed summary carries the current hash is skipped, so an edit of
# the Raman kernels does not re-take C2 / C4 / the linearized tags.
# usage: tools/profile_r06.sh [c2 c2pmc10k c2full c2aer c4 c4full lin lin112 fwd112 ia ialong c5 ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C="vsmartmom.jl_amd/csrc"
COMMON="$C/vsm_internal.h,$C/vsm_common.h,$C/vsm_inverse.h,$C/vsm_lds.h,$C/vsm_elemental.h,$C/vsm_api.hip,$C/vsm_generic.hip,$C/vsm_gemm_lds.h,$C/vsm_optics.hip,$C/vsm_surface.hip,$C/Makefile,include/vsmartmom_hip.h,vsmartmom.jl_amd/*.py,bench.py"
FWD="$COMMON,$C/vsm_native.hip,$C/vsm_native_dev.h,$C/vsm_native_run.h,$C/vsm_strip.hip,$C/vsm_strip_dev.h,$C/vsm_fused.hip"
F32="$FWD,$C/vsm_native32.hip,$C/vsm_native32_dev.h,$C/vsm_strip32.hip"
LIN="$FWD,$C/vsm_lin.hip,$C/vsm_striplin.hip,$C/vsm_strip128lin.hip,$C/vsm_strip128_dev.h,tools/lin_timing.py,tools/shape_cliff_timing.py"
BIG="$COMMON,$C/vsm_native.hip,$C/vsm_native_dev.h,$C/vsm_strip128.hip,$C/vsm_strip128_dev.h,tools/shape_cliff_timing.py"
RAM="$COMMON,$C/vsm_raman.hip,$C/vsm_raman_quad.hip,$C/vsm_raman_chain.hip,$C/vsm_raman_wave.hip,$C/vsm_fused.hip,$C/vsm_native.hip,$C/vsm_native_dev.h"
prof() {   # tag sources [profile_any options ...] -- command
  local tag=$1 src=$2; shift 2
  python tools/profile_any.py --out gpurun_out/prof_r06_$tag --sources "$src" --skip-if-unchanged profiles/r06/$tag/summary.json "$@"
}
for w in "$@"; do
  case $w in
    c2)     prof c2 "$FWD" --dtype f64 -- python bench.py --points 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ;;
    c2pmc10k) prof c2_10k "$FWD" --dtype f64 -- python bench.py --points 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ;;
    c2full) prof c2_default "$FWD" --skip-pmc -- python bench.py --no-cpu-baseline --no-secondary ;;
    c2aer)  prof c2_aer "$FWD" --dtype f64 -- python bench.py --variant aerosol --points 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ;;
    c4)     prof c4 "$F32" --dtype f32 -- python bench.py --config C4 --points 4096 --steps 1 --warmup 0 --no-cpu-baseline ;;
    c4full) prof c4_default "$F32" --skip-pmc -- python bench.py --config C4 --no-cpu-baseline ;;
    lin)    prof lin "$LIN" --dtype f64 -- python tools/lin_timing.py --points 2048 ;;
    lin112) prof lin112 "$LIN" --dtype f64 -- python tools/shape_cliff_timing.py --cases IQUV:51 ;;
    lin64)  prof lin64 "$LIN" --dtype f64 -- python tools/shape_cliff_timing.py --cases IQUV:27 ;;
    fwd112) prof fwd112 "$BIG" --dtype f64 -- python tools/shape_cliff_timing.py --no-lin --cases IQUV:51 ;;
    ia)     prof ia "$FWD,tools/ia_timing.py" --dtype f64 -- python tools/ia_timing.py --points 4096 --refl 0.1 --dsym 3 ;;
    ialong) prof ia_long "$FWD,tools/ia_timing.py" --dtype f64 -- python tools/ia_timing.py --points 4096 --refl 0.4 --dsym 3 ;;
    f32n60) prof f32_n60 "$F32,tools/shape_cliff_timing.py" --dtype f32 -- python tools/shape_cliff_timing.py --dtype f32 --no-lin --points 8192 --cases IQU:35 ;;
    c5)     prof c5 "$RAM" --dtype f64 --points-per-run 4000 --runs 2 -- python bench.py --config C5 --total-points 4000 --steps 1 --warmup 1 ;;
  esac
done
