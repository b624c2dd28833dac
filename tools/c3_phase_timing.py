#!/usr/bin/env python3
"""Where a linearized pass of the C3 scene (2 points, 22 moments, 33 layers) spends its time on the device: HIP-event intervals around
the elemental, doubling and interaction calls of an EAGER pass with one lane (no concurrency): python tools/c3_phase_timing.py"""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402

arch = vsm.Architectures.GPU(0)
with open(os.path.join(ROOT, "tests", "golden", "ocean_coxmunk_scene.json")) as f:
    d = json.load(f)
d.pop("source")
io, H, L_ = vsm.io_yaml, vsm.host_model, vsm.CoreRTLin
model = io.model_from_parameters(io.parameters_from_yaml(yaml.safe_dump(d)), arch)
S, L = model.tau_rayl.shape
prof = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
model.tau_abs = prof * 1.0
scene = L_.SceneLin(model, H.LinModel([prof * 1.0]), 0, 1, 1)
ev = defaultdict(list)


def wrap(name):
    f = getattr(L_, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        ev[name].append((e0, e1))
        return r
    setattr(L_, name, g)


for n in ("elemental_lin_", "doubling_allparams_", "interaction_lin_"):
    wrap(n)
for fold in (True, False):
    for rep in range(2):
        ev.clear()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        scene.run(lanes=2 if fold else 1, fold=fold, graph=False)
        t1.record()
        torch.cuda.synchronize()
    print("fold = %s: pass %.2f ms (eager)" % (fold, t0.elapsed_time(t1)))
    for n, l in ev.items():
        ms = [a.elapsed_time(b) for a, b in l]
        print("   %-22s calls %4d  total %7.2f ms  median %.4f ms  max %.4f" % (n, len(ms), sum(ms), float(np.median(ms)), max(ms)))
print("ndoubl per layer:", [ly["nd"] for ly in scene.fwd.moments[0]["layers"]])
