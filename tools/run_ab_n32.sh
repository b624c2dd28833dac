# timing of the FP32 native family (default library, or VSM_LIB_PATH variants given as arguments)
for v in default "$@"; do
  if [ $v = default ]; then unset VSM_LIB_PATH; else export VSM_LIB_PATH=vsmartmom.jl_amd/lib_dbg/libn32_$v.so; fi
  echo "== $v"
  python bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4', round(d['value']), 'pts/s', d['ms_per_step'])"
  python tools/shape_cliff_timing.py --dtype f32 --no-lin --points 8000 --cases IQU:29,IQU:35,IQUV:27,IQU:43 2>/dev/null | grep "^N="
done
