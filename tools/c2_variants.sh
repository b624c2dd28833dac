#!/bin/bash
# Bench every A/B build of tools/variants64.sh on the GPU box: tools/c2_variants.sh [bench.py args] -> one line per lib_dbg/libw_*.so
# (value, avg launch of the layer kernel, fraction of peak).  Diagnostic; the shipped library is vsmartmom.jl_amd/lib/.
cd "$(dirname "$0")/.."
for lib in vsmartmom.jl_amd/lib_dbg/libw_*.so; do
  out=$(VSM_LIB_PATH=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1)
  python - "$lib" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print("%-60s %8.0f pts/s  launch %.3f ms  frac %.4f" % (sys.argv[1].split("/")[-1], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", sys.argv[2][:200])
PY
done
