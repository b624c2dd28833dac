cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/linstats; mkdir -p gpurun_out/linstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/linstats -o l -- python tools/lin_timing.py --points 2048 "$@" > gpurun_out/linstats/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/linstats/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep "ratio" gpurun_out/linstats/log.txt | head -2
