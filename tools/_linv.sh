for f in vsmartmom.jl_amd/lib_dbg/libl_*.so; do
  n=$(basename $f .so); echo "$n: $(VSM_LIB_PATH=$PWD/$f python tools/lin_timing.py --points 2000 2>&1 | grep 'forward' | head -1)"
done
