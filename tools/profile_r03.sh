#!/bin/bash
# Round-3 rocprofv3 evidence (run on the GPU box through gpurun): stats + PMC passes for the kernels VERDICT r02 names.
# usage: tools/profile_r03.sh [c2|c4|lin|raman|ia ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for w in "$@"; do
  case $w in
    c2)    python tools/profile_any.py --out gpurun_out/prof_r03_c2 --dtype f64 -- python bench.py --points 4096 --steps 1 --warmup 0 --no-cpu-baseline ;;
    c2full) python tools/profile_any.py --skip-pmc --out gpurun_out/prof_r03_c2_default -- python bench.py --no-cpu-baseline --no-secondary ;;
    c4)    python tools/profile_any.py --out gpurun_out/prof_r03_c4 --dtype f32 -- python bench.py --config C4 --points 4096 --steps 1 --warmup 0 --no-cpu-baseline ;;
    c4full) python tools/profile_any.py --skip-pmc --out gpurun_out/prof_r03_c4_default -- python bench.py --config C4 --no-cpu-baseline ;;
    lin)   python tools/profile_any.py --out gpurun_out/prof_r03_lin --dtype f64 -- python tools/lin_timing.py --points 2048 ;;
    ia)    python tools/profile_any.py --out gpurun_out/prof_r03_ia --dtype f64 -- python tools/ia_timing.py --points 4096 --refl 0.1 ;;
    raman) python tools/profile_any.py --out gpurun_out/prof_r03_raman --dtype f64 -- python tools/raman_timing.py --points 4000 ;;
  esac
done
