"""rt_run(RRS) with the moment m = 0 as a Stokes_IQ run (core_rt_raman.REDUCE_M0) against the full-Stokes walk, and timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402


def scene(S, K=40, L=12, pol="IQU", lt=9):
    rng = np.random.default_rng(20260929)
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S, 1))
    tau_abs = (10.0 ** rng.uniform(-4, 0, (S, 1))) * dp[None, :]
    H = vsm.host_model
    model = H.model_from_arrays(vsm.Architectures.GPU(0), pol, lt, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075,
                                albedo=0.05, m_max=2)
    model.varpi_Cabannes = 0.96
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 7)
    rs = vsm.CoreRTRaman.RRS(shifts, np.full(len(shifts), 0.04 / len(shifts)), H.get_greek_rayleigh(0.75))
    return rs, model


def main():
    R = vsm.CoreRTRaman
    rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
    for pol, lt in (("IQU", 9), ("IQUV", 7), ("IQU", 5)):
        rs, model = scene(300, pol=pol, lt=lt)
        R.REDUCE_M0 = True
        a = R.rt_run(rs, model, 1)
        R.REDUCE_M0 = False
        b = R.rt_run(rs, model, 1)
        print(pol, lt, "reduced vs full:", ["%.2e" % rel(x, y) for x, y in zip(a, b)], flush=True)
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    rs, model = scene(S)
    for flag in (True, False, True):
        R.REDUCE_M0 = flag
        R.rt_run(rs, model, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R.rt_run(rs, model, 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("C5 shape S=%d reduce_m0=%s: %.3f s, %.0f points/s, peak mem %.1f GB" % (S, flag, dt, S / dt, torch.cuda.max_memory_allocated() / 1e9),
              flush=True)
    R.REDUCE_M0 = True


if __name__ == "__main__":
    main()
