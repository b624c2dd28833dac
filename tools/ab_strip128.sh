cd /root/repo
for lib in vsmartmom.jl_amd/lib_ab/r03.so vsmartmom.jl_amd/lib/libvsmartmom_hip.so; do
  echo "== $lib"
  VSM_LIB_PATH=$PWD/$lib python tools/shape_cliff_timing.py --no-lin --cases IQUV:43,IQUV:51,IQUV:59 2>&1 | grep "N="
done
