#!/usr/bin/env python3
"""Does a whole Scene.run() / SceneLin.run() capture into one HIP graph (torch.cuda.CUDAGraph on the launch stream), and what does
a replay cost against the eager pass?  C3 scene (2 spectral points, 22 Fourier moments, 33 layers): host-call bound when eager."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402
import bench_secondary as BS  # noqa: E402
import json  # noqa: E402
import yaml  # noqa: E402


def main():
    arch = vsm.Architectures.GPU(0)
    with open(os.path.join(ROOT, "tests", "golden", "ocean_coxmunk_scene.json")) as f:
        d = json.load(f)
    d.pop("source")
    io, H = vsm.io_yaml, vsm.host_model
    model = io.model_from_parameters(io.parameters_from_yaml(yaml.safe_dump(d)), arch)
    S, L = model.tau_rayl.shape
    prof = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
    model.tau_abs = prof * 1.0
    for name, mk in (("forward", lambda: vsm.CoreRT.prepare_scene(model)),
                     ("linearized", lambda: vsm.CoreRTLin.SceneLin(model, H.LinModel([prof * 1.0]), 0, 1, 1))):
        scene = mk()
        if name == "linearized" and len(sys.argv) > 1:
            run0 = scene.run
            scene.run = lambda: run0(lanes=int(sys.argv[1]), fold=(len(sys.argv) > 2 and sys.argv[2] == "fold"))
        scene.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = scene.run()
        torch.cuda.synchronize()
        eager = time.perf_counter() - t0
        ref = [t.clone() for t in out]
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out_g = scene.run()
            torch.cuda.synchronize()
            for t in out_g:
                t.zero_()
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            rep = time.perf_counter() - t0
            same = all(torch.equal(a, b) for a, b in zip(ref, out_g))
            print("%s: eager %.1f ms, graph replay %.1f ms, results identical: %s" % (name, 1e3 * eager, 1e3 * rep, same))
        except Exception as ex:
            print("%s: eager %.1f ms, capture failed: %s: %s" % (name, 1e3 * eager, type(ex).__name__, str(ex)[:300]))


if __name__ == "__main__":
    main()
