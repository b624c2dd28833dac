"""Native-layout run (vsm_run_*) against the reference-layout layer loop and the oracle, shape by shape; optional timing.
    python tools/native_check.py [--time S]"""
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
from oracle import vsm_oracle as O  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", type=int, default=0)
    ap.add_argument("--oracle", type=int, default=1)
    args = ap.parse_args()
    arch = vsm.Architectures.GPU(0)
    H = vsm.host_model
    rng = np.random.default_rng(5)
    bad = 0
    shapes = [("IQU", 35, 3), ("IQU", 21, 3), ("I", 9, 3), ("I", 33, 3), ("IQ", 21, 3), ("IQUV", 21, 3), ("IQU", 11, 3),
              ("I", 3, 2), ("IQUV", 11, 4), ("I", 55, 3), ("IQU", 5, 4)]
    for pol, lt, L in shapes:
        S = 9
        tau_rayl = np.tile(np.linspace(0.02, 0.3, L), (S, 1))
        tau_abs = 10.0 ** rng.uniform(-3, 0.5, (S, L))
        kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=2)
        model = H.model_from_arrays(arch, pol, lt, 40.0, [30.0, 10.0], [0.0, 75.0], **kw)
        N = model.quad_points.Nquad * model.polarization_type.n
        vsm.CoreRT.NATIVE_RUN = True
        sc = vsm.CoreRT.prepare_scene(model)
        nat = sorted(sc._native_moments())
        sc.run()
        torch.cuda.synchronize()
        vsm._lib.check_device_status("native")
        Rn, Tn = sc.results_host()
        vsm.CoreRT.NATIVE_RUN = False
        sc2 = vsm.CoreRT.prepare_scene(model)
        sc2.run()
        torch.cuda.synchronize()
        Rl, Tl = sc2.results_host()
        vsm.CoreRT.NATIVE_RUN = True
        msg = "pol=%-4s N=%3d L=%d coupling=%s native=%s  native-vs-legacy R %.2e T %.2e" % (
            pol, N, L, [hex(c) for c in (sc.coupling or [])], nat, rel(Rn, Rl), rel(Tn, Tl))
        if args.oracle:
            Ro, To = O.rt_run(O.build_model(pol, lt, 40.0, [30.0, 10.0], [0.0, 75.0], **kw))
            msg += "  vs oracle R %.2e T %.2e (legacy %.2e %.2e)" % (rel(Rn, Ro), rel(Tn, To), rel(Rl, Ro), rel(Tl, To))
            if not (rel(Rn, Ro) < 1e-8 and rel(Tn, To) < 1e-8):
                bad += 1
        if not np.all(np.isfinite(Rn)) or rel(Rn, Rl) > 1e-9:
            bad += 1
        print(msg, flush=True)
    # thick, near-conservative layers: long series orders and the pivoted inverse
    for pol, lt in (("IQU", 35), ("I", 21)):
        S, L = 5, 3
        tau_rayl = np.tile(np.array([0.5, 4.0, 30.0]), (S, 1))
        tau_abs = np.tile(np.array([1e-4, 1e-5, 1e-6]), (S, 1)) * (1 + rng.random((S, 1)))
        kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.6, m_max=2)
        model = H.model_from_arrays(arch, pol, lt, 40.0, [30.0], [0.0], **kw)
        vsm.CoreRT.NATIVE_RUN = True
        Rn, Tn = vsm.CoreRT.rt_run(model)
        st = list(vsm._lib.last_device_status)
        vsm.CoreRT.NATIVE_RUN = False
        Rl, Tl = vsm.CoreRT.rt_run(model)
        vsm.CoreRT.NATIVE_RUN = True
        Ro, To = O.rt_run(O.build_model(pol, lt, 40.0, [30.0], [0.0], **kw))
        print("thick %s: native-vs-legacy R %.2e T %.2e; vs oracle R %.2e T %.2e; device status %s" % (
            pol, rel(Rn, Rl), rel(Tn, Tl), rel(Rn, Ro), rel(Tn, To), st), flush=True)
        if not (rel(Rn, Ro) < 1e-8 and rel(Tn, To) < 1e-8):
            bad += 1
    if args.time:
        S, L = args.time, 40
        tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
        model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                    albedo=0.15, m_max=2)
        for flag in (True, False, True):
            vsm.CoreRT.NATIVE_RUN = flag
            sc = vsm.CoreRT.prepare_scene(model)
            sc.compute_hdrf = False
            sc.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sc.run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print("C2 shape S=%d native=%s: %.1f ms per pass, %.0f points/s" % (S, flag, dt * 1e3, S / dt), flush=True)
        vsm.CoreRT.NATIVE_RUN = True
    print("native_check: %s" % ("OK" if bad == 0 else "%d FAILURES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
