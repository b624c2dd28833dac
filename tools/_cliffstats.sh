# rocprofv3 --stats of tools/shape_cliff_timing.py for one case: tools/_cliffstats.sh TAG <shape_cliff_timing args>
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag=$1; shift
rm -rf gpurun_out/cliffstats_$tag; mkdir -p gpurun_out/cliffstats_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cliffstats_$tag -o l -- python tools/shape_cliff_timing.py "$@" > gpurun_out/cliffstats_$tag/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/cliffstats_$tag/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
print("== $tag")
for r in rows[:14]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "N=" gpurun_out/cliffstats_$tag/log.txt | head -3
find gpurun_out/cliffstats_$tag -name "*kernel_trace.csv" -delete
