# C3 linearized step vs the number of moment lanes: bash tools/c3_lanes.sh
for lanes in 3 4 6 8 12 16; do
python - $lanes <<'PY' 2>&1 | grep -v "Extension\|amdgpu.ids" | tail -1
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import vsmartmom_jl_amd as vsm
import bench_secondary as BS
vsm.CoreRTLin.SceneLin.LANES = int(sys.argv[1])
e = BS.c3_lin(vsm, torch, vsm.Architectures.GPU(0))
print("LANES", sys.argv[1], "%.2f ms/step (graph replay); eager %.2f" % (e["ms_per_step"], e["ms_per_step_launch_by_launch"]), flush=True)
PY
done
