# C3 (linearized ocean scene, 2 points): the step with / without the layer-parallel doubling and the interaction tree: bash tools/c3_ab.sh
for cfg in "False False" "True False" "True True"; do
set -- $cfg
python - $1 $2 <<'PY' 2>&1 | grep -v "Extension\|amdgpu.ids" | tail -3
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import vsmartmom_jl_amd as vsm
import bench_secondary as BS
SL = vsm.CoreRTLin.SceneLin
SL.PARALLEL_LAYERS, SL.TREE_INTERACTIONS = sys.argv[1] == "True", sys.argv[2] == "True"
e = BS.c3_lin(vsm, torch, vsm.Architectures.GPU(0))
print("PARALLEL_LAYERS %s TREE %s: %.2f ms/step (graph replay); eager %.2f; replay equals eager: %s" % (
    sys.argv[1], sys.argv[2], e["ms_per_step"], e["ms_per_step_launch_by_launch"], e["graph_replay_equals_launch_by_launch"]), flush=True)
PY
done
