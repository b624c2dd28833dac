#!/usr/bin/env python3
"""Forward and linearized rt_run over a sweep of matrix sizes N around the kernel-selection boundaries (N = 32|33, 60|61, 64|65,
96|97): points/s and algorithmic TFLOP/s per size, to show where a shape falls off a fused path.  Diagnostic, not the bench."""
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

NS = {"I": 1, "IQU": 3, "IQUV": 4}


def timed(f):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--no-lin", action="store_true")
    ap.add_argument("--cases", default="IQU:15,IQU:17,IQU:27,IQU:33,IQU:35,IQU:37,IQUV:27,I:127,IQU:41,IQUV:43")
    a = ap.parse_args()
    S, L = a.points, a.layers
    FT = np.float64 if a.dtype == "f64" else np.float32
    arch = vsm.Architectures.GPU(0)
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    H = vsm.host_model
    for case in a.cases.split(","):
        pol, lt = case.split(":")
        model = H.model_from_arrays(arch, pol, int(lt), 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                    albedo=0.15, m_max=2, float_type=FT)
        N = model.quad_points.Nquad * NS[pol]
        scene = vsm.CoreRT.prepare_scene(model)
        t_f = timed(lambda: scene.run())
        f_f = scene.flops_per_point()
        msg = "N=%3d (%s, Nquad=%d) S=%d L=%d: forward %7.0f points/s %5.1f TFLOP/s" % (
            N, pol, model.quad_points.Nquad, S, L, S / t_f, f_f * S / t_f / 1e12)
        del scene
        if not a.no_lin:
            lin = H.LinModel([tau_abs])
            ls = vsm.CoreRTLin.SceneLin(model, lin, 0, 1, 1)
            t_l = timed(lambda: ls.run())
            f_l = ls.flops_per_point()
            msg += " ; linearized (1 gas + albedo) %7.0f points/s %5.1f TFLOP/s (wall ratio %.2f, flop ratio %.2f)" % (
                S / t_l, f_l * S / t_l / 1e12, t_l / t_f, f_l / f_f)
            del ls
        print(msg, flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
