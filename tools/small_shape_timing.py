#!/usr/bin/env python3
"""Throughput on quickstart-shaped problems (BASELINE.json configs[0]: config/quickstart.yaml -- Stokes_I, 3 streams,
2..10 layers, ~100 spectral points) and on the reference's own benchmark scene shape (Phase1b: IQU, nstreams = 3 (N = 15),
12 layers, 103 points; BASELINE.md sec. 1: 340 points/s CPU, 553 points/s A100).  These are launch-bound."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402


def run(name, pol, l_trunc, S, L, FT, vza=(30.0,), reps=20):
    rng = np.random.default_rng(0)
    arch = vsm.Architectures.GPU(0)
    tau_rayl = np.tile(0.1 * np.full(L, 1.0 / L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0.5, (S, 1)) * np.full((1, L), 1.0 / L)
    model = vsm.host_model.model_from_arrays(arch, pol, l_trunc, 60.0, list(vza), [0.0] * len(vza), tau_rayl=tau_rayl,
                                             tau_abs=tau_abs, depol=0.03, albedo=0.1, m_max=2, float_type=FT)
    N = model.quad_points.Nquad * model.polarization_type.n
    scene = vsm.CoreRT.prepare_scene(model)
    scene.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scene.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ref = [x.clone() for x in scene.run()]
    scene.run_graph()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    for _ in range(reps):
        out = scene.run_graph()
    torch.cuda.synchronize()
    dt_graph = (time.perf_counter() - tg) / reps
    same = all(torch.equal(a, b) for a, b in zip(ref, out))
    t1 = time.perf_counter()
    for _ in range(5):
        vsm.CoreRT.rt_run(model)
    dt_full = (time.perf_counter() - t1) / 5
    print("%-28s N=%3d S=%5d L=%2d %s: device pass %.2f ms -> %.3g points/s ; HIP-graph replay %.2f ms -> %.3g points/s (bit-identical: %s) ; "
          "full rt_run(model) incl. host optics + H2D %.2f ms -> %.3g points/s"
          % (name, N, S, L, FT.__name__, 1e3 * dt, S / dt, 1e3 * dt_graph, S / dt_graph, same, 1e3 * dt_full, S / dt_full))


if __name__ == "__main__":
    run("quickstart (2 layers)", "I", 5, 100, 2, np.float64)
    run("quickstart-shaped (10 layers)", "I", 5, 100, 10, np.float64)
    run("quickstart-shaped, 10^4 pts", "I", 5, 10000, 10, np.float64)
    run("Phase1b noRS shape", "IQU", 5, 103, 12, np.float32)
    run("Phase1b noRS shape, 10^4 pts", "IQU", 5, 10000, 12, np.float32)
