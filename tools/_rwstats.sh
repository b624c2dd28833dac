cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/rwstats; mkdir -p gpurun_out/rwstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rwstats -o rw -- python tools/raman_timing.py --points 4000 > gpurun_out/rwstats/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/rwstats/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "Raman RRS" gpurun_out/rwstats/log.txt
