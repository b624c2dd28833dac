# C3 (linearized ocean scene, 2 points) with / without the layer-parallel doubling: bash tools/c3_ab.sh
for par in False True; do
python - $par <<'PY' 2>&1 | grep -v "Extension\|amdgpu.ids" | tail -3
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import vsmartmom_jl_amd as vsm
import bench_secondary as BS
vsm.CoreRTLin.SceneLin.PARALLEL_LAYERS = sys.argv[1] == "True"
e = BS.c3_lin(vsm, torch, vsm.Architectures.GPU(0))
print("PARALLEL_LAYERS", sys.argv[1], "%.2f ms/step (graph replay); eager %.2f; device pass %.2f ms; replay equals eager: %s" % (
    e["ms_per_step"], e["ms_per_step_launch_by_launch"], e["device_pass_ms"], e["graph_replay_equals_launch_by_launch"]), flush=True)
PY
done
