# rocprofv3 --stats of the Raman run past the wave-per-line limit (N = 39): tools/_rw39.sh [raman_timing args]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/rw39; mkdir -p gpurun_out/rw39
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rw39 -o l -- python tools/raman_timing.py --points 1000 --l-trunc 21 "$@" > gpurun_out/rw39/log.txt 2>&1
grep -E "Raman RRS|algorithmic" gpurun_out/rw39/log.txt
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/rw39/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:12]: print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
find gpurun_out/rw39 -name "*kernel_trace.csv" -delete
