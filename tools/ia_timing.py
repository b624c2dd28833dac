#!/usr/bin/env python3
"""The interaction kernel in isolation (north_star: ">= 40 % MFMA utilisation in the interaction kernel"): repeated
interaction!(::ScatteringInterface_11) on random physical layers, N <= 64 FP64 (k_ia_native<RT, KS, DSYM>) or N = 96 FP32
(k_ia_strip32<6>).  Wrap in tools/profile_any.py for the PMC MFMA-busy fraction.  Diagnostic; not the bench contract."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--N", type=int, default=60)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--refl", type=float, default=0.4,
                    help="scale of the reflection operators (||r R|| ~ 0.2 refl^2: 0.4 = a clear-sky composite, series inverse; 1.5 forces Gauss-Jordan)")
    ap.add_argument("--dsym", type=int, default=0,
                    help="nStokes of a D-symmetric added layer (r+- = D r-+ D, t-- = D t++ D derived in the kernel, as doubling! leaves "
                         "a scattering layer); 0 = all four matrices of the added layer in memory (surface-layer form)")
    a = ap.parse_args()
    FT = np.float64 if a.dtype == "f64" else np.float32
    N, S = a.N, a.points
    rng = np.random.default_rng(1)
    arch = vsm.Architectures.GPU(0)
    CR = vsm.CoreRT
    pc = CR.make_composite_layer(FT, arch, (N, N), S)
    pa = CR.make_added_layer(FT, arch, (N, N), S, d_symmetric=a.dsym)

    def refl(scale):
        return CR.to_device_matrix((scale * rng.random((S, N, N)) / N).astype(FT), arch, FT)

    def trans():
        return CR.to_device_matrix((np.eye(N)[None] * rng.uniform(0.3, 0.95, (S, N, 1)) + 0.05 * rng.random((S, N, N)) / N).astype(FT),
                                   arch, FT)

    conv_v = vsm.Architectures.array_type(arch)
    init = dict(R_mp=refl(a.refl), R_pm=refl(a.refl), T_pp=trans(), T_mm=trans(), J0_p=conv_v(rng.random((S, N)).astype(FT)),
                J0_m=conv_v(rng.random((S, N)).astype(FT)))
    for k in ("r_mp", "r_pm")[:1 if a.dsym else 2]:
        getattr(pa, k).copy_(refl(0.75 * a.refl))
    for k in ("t_pp", "t_mm")[:1 if a.dsym else 2]:
        getattr(pa, k).copy_(trans())
    pa.j0_p.copy_(conv_v(rng.random((S, N)).astype(FT)))
    pa.j0_m.copy_(conv_v(rng.random((S, N)).astype(FT)))

    def reset():
        for k, v in init.items():
            getattr(pc, k).copy_(v)

    reset()
    CR.interaction_("11", pc, pa)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    tot = 0.0
    for _ in range(a.reps):
        reset()   # (the composite is updated in place: start every repetition from the same operators)
        torch.cuda.synchronize()
        ev[0].record()
        CR.interaction_("11", pc, pa)
        ev[1].record()
        torch.cuda.synchronize()
        tot += ev[0].elapsed_time(ev[1])
    ms = tot / a.reps
    flop = S * (24.0 * N ** 3 + 8.0 * N ** 2)
    peak = 78.6 if a.dtype == "f64" else 157.3
    print("interaction!(_11) N=%d S=%d %s: %.4f ms per launch -> %.1f TFLOP/s algorithmic = %.3f of the %s MFMA peak (%.1f TF)"
          % (N, S, a.dtype, ms, flop / ms / 1e9, flop / ms / 1e9 / peak, a.dtype.upper(), peak))


if __name__ == "__main__":
    main()
