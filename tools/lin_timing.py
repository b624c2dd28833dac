#!/usr/bin/env python3
"""Timing of the linearized (Jacobian) pass rt_run(model, lin_model, 0, NGas, 1) on a C2-shaped problem next to the
forward pass on the same model.  Diagnostic (numbers quoted in DESIGN.md); not the bench contract."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--gases", type=int, default=1)
    a = ap.parse_args()
    S, L = a.points, a.layers
    arch = vsm.Architectures.GPU(0)
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    H = vsm.host_model
    model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                albedo=0.15, m_max=2)
    lin = H.LinModel([tau_abs / a.gases for _ in range(a.gases)])
    P = a.gases + 1
    N = model.quad_points.Nquad * 3

    def timed(f):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = f()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    fwd_scene = vsm.CoreRT.prepare_scene(model)
    t_fwd, _ = timed(lambda: fwd_scene.run())
    t0 = time.perf_counter()
    lin_scene = vsm.CoreRTLin.SceneLin(model, lin, 0, a.gases, 1)
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    t_lin, _ = timed(lambda: lin_scene.run())
    out = lin_scene.results_host()
    nd = 8
    n3, n2 = float(N) ** 3, float(N) ** 2
    f_fwd = 3 * (L * nd * (12 * n3 + 8 * n2) + L * (24 * n3 + 8 * n2))
    f_lin = 3 * (L * a.gases * nd * (24 * n3 + 16 * n2) + L * P * (48 * n3 + 16 * n2))
    print("N=%d S=%d L=%d P=%d (%d gas + albedo): forward %.3f s (%.0f points/s) ; linearized %.3f s (%.0f points/s) ; ratio %.1f"
          % (N, S, L, P, a.gases, t_fwd, S / t_fwd, t_lin, S / t_lin, t_lin / t_fwd))
    print("  algorithmic GFLOP/point: forward %.2f, linearized adds %.2f (SURVEY 8d) -> %.1f TFLOP/s in the linearized run"
          % (f_fwd / 1e9, f_lin / 1e9, (f_fwd + f_lin) * S / t_lin / 1e12))
    f_tot = lin_scene.flops_per_point()
    print("  device passes only (scene.run()); SceneLin build (host optics + H2D) %.3f s; algorithmic GFLOP/point of the "
          "linearized run %.2f -> %.1f TFLOP/s = %.3f of the FP64 MFMA peak; wall ratio %.2f vs flop ratio %.2f"
          % (t_prep, f_tot / 1e9, f_tot * S / t_lin / 1e12, f_tot * S / t_lin / 78.6e12, t_lin / t_fwd, f_tot / fwd_scene.flops_per_point()))
    print("  max |dR/dalbedo| = %.3e" % np.abs(out[2][..., -1]).max())


if __name__ == "__main__":
    main()
