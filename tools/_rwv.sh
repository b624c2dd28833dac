for v in "" "$@"; do
  if [ -n "$v" ]; then export VSM_LIB_PATH=$PWD/vsmartmom.jl_amd/lib_dbg/libv_$v.so; fi
  echo "== ${v:-base}: $(timeout 200 python tools/raman_timing.py --points 4000 2>&1 | grep 'Raman RRS' | sed 's/.*second run//')"
done
