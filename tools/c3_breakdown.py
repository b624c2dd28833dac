#!/usr/bin/env python3
"""Wall-clock breakdown of a linearized C3 step (2 points): inputs (H2D + device optics), pass (eager / graph replay), D2H."""
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
import bench_secondary as BS  # noqa: E402

arch = vsm.Architectures.GPU(0)
with open(os.path.join(ROOT, "tests", "golden", "ocean_coxmunk_scene.json")) as f:
    d = json.load(f)
d.pop("source")
io, H = vsm.io_yaml, vsm.host_model
model = io.model_from_parameters(io.parameters_from_yaml(yaml.safe_dump(d)), arch)
S, L = model.tau_rayl.shape
prof = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
model.tau_abs = prof * 1.0
scene = vsm.CoreRTLin.SceneLin(model, H.LinModel([prof * 1.0]), 0, 1, 1)


def t(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), r


for graph in ((False,) if "--eager-only" in sys.argv else (False, True)):
    scene.graph_replay = graph
    for rep in range(3):
        a, _ = t(lambda: (scene.fwd.upload(), scene.fwd.prepare()))
        b, _ = t(lambda: (scene.upload(), scene.prepare()))
        c, _ = t(lambda: scene.run())
        e, _ = t(lambda: scene.results_host())
    print("graph=%s: fwd upload+prepare %.2f ms, lin upload+prepare %.2f ms, run %.2f ms, D2H %.2f ms; total %.2f" % (graph, a, b, c, e, a + b + c + e))
if "--profile" in sys.argv:
    import cProfile
    import pstats
    scene.graph_replay = True
    pr = cProfile.Profile()
    pr.enable()
    for rep in range(5):
        scene.fwd.upload(); scene.fwd.prepare(); scene.upload(); scene.prepare(); scene.run(); scene.results_host()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
