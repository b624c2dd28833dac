#!/usr/bin/env python3
"""elemental! + doubling! alone (vsm_elemental_doubling -> k_ed_strip<KS> for FP64 32 < N <= 60) on the C2 layer optics:
time per launch and per doubling step against the MFMA time of its products, at 1 and 2 workgroups per CU and at full
occupancy.  Diagnostic; not the bench contract."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[256, 512, 10240])
    ap.add_argument("--layer", type=int, default=20)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    arch = vsm.Architectures.GPU(0)
    CR = vsm.CoreRT
    for S in a.points:
        tau_rayl, tau_abs = bench.o2a_atmosphere(S, 40)
        model = vsm.host_model.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                                                 depol=0.0279, albedo=0.15, m_max=0)
        scene = CR.Scene(model, full_added_layer=True)
        ly = scene.moments[0]["layers"][a.layer]
        N = scene.N
        res = []
        for nd in (0, ly["nd"]):
            def go():
                CR.elemental_doubling_(scene.pol, ly["tau_sum"], ly["dtau"], scene.F0, ly["props"], 0, nd, scene.dq, scene.added)
            go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / a.reps)
        nd = ly["nd"]
        rounds = -(-S // 512)
        step_ms = (res[1] - res[0]) / nd
        per_pair = step_ms * 1e-3 * a.clock_ghz * 1e9 / rounds          # cycles per doubling step of the workgroups sharing a CU
        wg = min(2, -(-S // 256))
        ks = (N + 3) // 4
        flops = S * nd * (12.0 * N ** 3 + 8.0 * N ** 2)
        print("S=%6d N=%d nd=%d: launch %.3f ms (nd=0: %.3f ms), doubling step %.4f ms = %.0f cycles per CU-round (%d workgroup(s) per CU); "
              "MFMA issue of 6 products: %d cycles per workgroup; algorithmic %.1f TFLOP/s = %.3f of peak"
              % (S, N, nd, res[1], res[0], step_ms, per_pair, wg, 6 * 4 * ks * 64, flops / ((res[1] - res[0]) * 1e-3) / 1e12,
                 flops / ((res[1] - res[0]) * 1e-3) / 78.6e12))


if __name__ == "__main__":
    main()
