# second PMC pass over lib_dbg/libv_*.so: latencies (LEVEL / count) of VMEM, LDS and instruction fetch for k_layer_strip32
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for f in vsmartmom.jl_amd/lib_dbg/libv_*.so; do
  n=$(basename $f .so); d=/tmp/pmc2_$n; rm -rf $d
  VSM_LIB_PATH=$PWD/$f rocprofv3 --output-format csv --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $d -o p -- python bench.py --config C4 --points 4096 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc2_$n.log 2>&1
  python - $d $n <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float); disp = set()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_layer_strip32" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
L = max(len(disp), 1)
print(n, {k: "%.4g" % (v / L) for k, v in agg.items()})
print("  VMEM latency (LEVEL/INSTS) %.0f ; LDS latency %.0f ; IFETCH latency %.0f ; ifetch per wave-cycle %.4f ; IFETCH_LEVEL/WAVE_CYCLES %.3f"
      % (agg["SQ_INST_LEVEL_VMEM"] / max(agg["SQ_INSTS_VMEM"], 1), agg["SQ_INST_LEVEL_LDS"] / max(agg["SQ_INSTS_LDS"], 1),
         agg["SQ_IFETCH_LEVEL"] / max(agg["SQ_IFETCH"], 1), agg["SQ_IFETCH"] / agg["SQ_WAVE_CYCLES"], agg["SQ_IFETCH_LEVEL"] / agg["SQ_WAVE_CYCLES"]))
PY
done
