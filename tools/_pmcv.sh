# PMC pass over the A/B builds (lib_dbg/libv_*.so): LDS activity, waits, duration of k_layer_strip32 on C4 at 4096 points
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for f in vsmartmom.jl_amd/lib_dbg/libv_*.so; do
  n=$(basename $f .so); d=/tmp/pmc_$n; rm -rf $d
  VSM_LIB_PATH=$PWD/$f rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $d -o p -- python bench.py --config C4 --points 4096 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc_$n.log 2>&1
  python - $d $n <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); disp = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_layer_strip32" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_layer_strip32" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
L = max(len(disp), 1)
busy = agg["SQ_BUSY_CYCLES"] / 32 / L
print("%-14s launches %d avg_ms %.3f | cycles/launch %.3gM | LDS active %.1f%% (conflict %.1f%% of it) | LDS instr/launch %.3g | waves/SIMD %.2f | wait_any %.3f wait_inst %.3f wait_lds %.3f"
      % (n, L, sum(dur) / max(len(dur), 1), busy / 1e6, 100 * agg["SQ_LDS_IDX_ACTIVE"] / L / 256 / busy, 100 * agg["SQ_LDS_BANK_CONFLICT"] / max(agg["SQ_LDS_IDX_ACTIVE"], 1),
         agg["SQ_INSTS_LDS"] / L, agg["SQ_WAVE_CYCLES"] * 4 / 1024 / L / busy, agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_INST_LDS"] / agg["SQ_WAVE_CYCLES"]))
PY
done
