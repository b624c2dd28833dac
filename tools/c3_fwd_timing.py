#!/usr/bin/env python3
"""Forward rt_run of the two-point ocean scene (config/ocean_coxmunk.yaml shape): wall-clock of a step (inputs, pass, D2H)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402

arch = vsm.Architectures.GPU(0)
with open(os.path.join(ROOT, "tests", "golden", "ocean_coxmunk_scene.json")) as f:
    d = json.load(f)
d.pop("source")
io = vsm.io_yaml
model = io.model_from_parameters(io.parameters_from_yaml(yaml.safe_dump(d)), arch)
S, L = model.tau_rayl.shape
model.tau_abs = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
scene = vsm.CoreRT.prepare_scene(model)


def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), r


for rep in range(4):
    a, _ = t(lambda: (scene.upload(), scene.prepare()))
    b, _ = t(lambda: scene.run())
    c, _ = t(lambda: scene.results_host())
print("N = %d, %d moments, %d layers, %d points: upload+prepare %.2f ms, run %.2f ms, D2H %.2f ms" % (
    scene.N, len(scene.moments), scene.Nz, scene.S, a, b, c))
t0 = time.perf_counter()
for _ in range(5):
    vsm.CoreRT.rt_run(model)
torch.cuda.synchronize()
print("whole rt_run(model) call: %.2f ms" % (1e3 * (time.perf_counter() - t0) / 5))
