#!/usr/bin/env python3
"""C2-shaped run with the `:thermal` per-source slot next to the solar-only run: what the slot costs with the fused thermal
layer launch (vsm_layer_forward_thermal) and at operator level (VSM_NO_THERMAL_FUSION=1)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402


def timed(f):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4000)
    ap.add_argument("--layers", type=int, default=40)
    a = ap.parse_args()
    S, L = a.points, a.layers
    arch = vsm.Architectures.GPU(0)
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    H = vsm.host_model
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=2)
    B = 0.1 * np.ones((L, S))
    solar = vsm.CoreRT.prepare_scene(H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], **kw))
    t_s = timed(lambda: solar.run())
    both = vsm.CoreRT.prepare_scene(H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0],
                                                        sources=(H.SolarBeam(), H.ThermalEmission(B_layer=B)), **kw))
    t_f = timed(lambda: both.run())
    vsm.CoreRT.THERMAL_FUSION = False
    t_o = timed(lambda: both.run())
    print("N=60 S=%d L=%d: solar only %.3f s (%.0f points/s); + thermal slot fused %.3f s (%.0f points/s, x%.2f); + thermal slot "
          "operator level %.3f s (%.0f points/s, x%.2f)" % (S, L, t_s, S / t_s, t_f, S / t_f, t_f / t_s, t_o, S / t_o, t_o / t_s))


if __name__ == "__main__":
    main()
