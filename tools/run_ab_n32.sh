# timing of the FP32 native family (default library, or VSM_LIB_PATH variants given as arguments)
for v in default "$@"; do
  if [ $v = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=vsmartmom.jl_amd/lib_dbg/libn32_$v.so; fi
  echo "== $v"
  python tools/shape_cliff_timing.py --dtype f32 --no-lin --points 8000 --cases IQU:17,IQU:21,IQU:29,IQU:35,IQUV:27 2>/dev/null | grep "^N="
done
