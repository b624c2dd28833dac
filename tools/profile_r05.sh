#!/bin/bash
# Round-5 rocprofv3 evidence (run on the GPU box through gpurun): stats + PMC passes, one tag per directory under gpurun_out/;
# copy the condensed summary.json / kernel_stats.csv to profiles/r05/<tag>/.  Every summary names the library build it profiled
# (library.source_hash = vsm_build_id(), tools/source_hash.sh <commit> recomputes it from a tree).
# usage: tools/profile_r05.sh [c2 c2full c4 c4full lin lin112 fwd112 ia c5 ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P="python tools/profile_any.py"
for w in "$@"; do
  case $w in
    c2)     $P --out gpurun_out/prof_r05_c2 --dtype f64 -- python bench.py --points 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ;;
    c2full) $P --skip-pmc --out gpurun_out/prof_r05_c2_default -- python bench.py --no-cpu-baseline --no-secondary ;;
    c4)     $P --out gpurun_out/prof_r05_c4 --dtype f32 -- python bench.py --config C4 --points 4096 --steps 1 --warmup 0 --no-cpu-baseline ;;
    c4full) $P --skip-pmc --out gpurun_out/prof_r05_c4_default -- python bench.py --config C4 --no-cpu-baseline ;;
    lin)    $P --out gpurun_out/prof_r05_lin --dtype f64 -- python tools/lin_timing.py --points 2048 ;;
    lin112) $P --out gpurun_out/prof_r05_lin112 --dtype f64 -- python tools/shape_cliff_timing.py --cases IQUV:51 ;;
    lin64)  $P --out gpurun_out/prof_r05_lin64 --dtype f64 -- python tools/shape_cliff_timing.py --cases IQUV:27 ;;
    lin30)  $P --out gpurun_out/prof_r05_lin30 --dtype f64 -- python tools/shape_cliff_timing.py --cases IQU:15 ;;
    fwd112) $P --out gpurun_out/prof_r05_fwd112 --dtype f64 -- python tools/shape_cliff_timing.py --no-lin --cases IQUV:51 ;;
    ia)     $P --out gpurun_out/prof_r05_ia --dtype f64 -- python tools/ia_timing.py --points 4096 --refl 0.1 --dsym 3 ;;
    c2aer)  $P --out gpurun_out/prof_r05_c2_aer --dtype f64 -- python bench.py --variant aerosol --points 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ;;
    c5)     $P --out gpurun_out/prof_r05_c5 --dtype f64 --points-per-run 4000 --runs 2 -- python bench.py --config C5 --total-points 4000 --steps 1 --warmup 1 ;;
  esac
done
