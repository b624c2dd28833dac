for f in vsmartmom.jl_amd/lib_dbg/libw_*.so; do
  n=$(basename $f .so); 
  VSM_LIB_PATH=$PWD/$f python bench.py --no-cpu-baseline --steps 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_ms'],3))"
done
