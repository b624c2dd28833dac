#!/usr/bin/env python3
"""Timing of the rotational-Raman pass on a C5-shaped problem (BASELINE.json configs[4]): nStokes = 3, spectral points S,
K Raman offsets, L layers.  Diagnostic (numbers quoted in DESIGN.md); not the bench contract."""
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
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--lines", type=int, default=40)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--l-trunc", type=int, default=9)
    ap.add_argument("--oracle-points", type=int, default=0)
    ap.add_argument("--json", action="store_true", help="also print one JSON line in the shape of the bench.py contract")
    a = ap.parse_args()
    S, K, L = a.points, a.lines, a.layers
    rng = np.random.default_rng(20260929)
    arch = vsm.Architectures.GPU(0)
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S, 1))
    tau_abs = (10.0 ** rng.uniform(-4, 0, (S, 1))) * dp[None, :]
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.05, m_max=2)
    model = vsm.host_model.model_from_arrays(arch, "IQU", a.l_trunc, 40.0, [30.0], [0.0], **kw)
    model.varpi_Cabannes = 0.96
    N = model.quad_points.Nquad * 3
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 7)
    w_ie = np.full(len(shifts), 0.04 / len(shifts))
    rs = vsm.CoreRTRaman.RRS(shifts, w_ie, vsm.host_model.get_greek_rayleigh(0.75))
    t0 = time.perf_counter()
    out = vsm.CoreRTRaman.rt_run(rs, model, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = vsm.CoreRTRaman.rt_run(rs, model, 1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("Raman RRS: N=%d S=%d K=%d L=%d m=0..2: first run %.2f s, second run %.2f s -> %.0f spectral-points/s; peak device memory %.1f GB"
          % (N, S, len(shifts), L, t1 - t0, t2 - t1, S / (t2 - t1), torch.cuda.max_memory_allocated() / 1e9))
    print("  max |ieR| = %.3e, max |R| = %.3e" % (np.abs(out[2]).max(), np.abs(out[0]).max()))
    # ALGORITHMIC flops (counted from the reference's statements): per doubling step the elastic 12N^3+8N^2 plus,
    # per in-band Raman line, 16 products + 10 mat-vecs (doubling_inelastic.jl:13-164); per interaction the elastic 24N^3+8N^2
    # plus, per in-band line, 18 products + 8 mat-vecs (interaction_inelastic.jl:319-521)
    H = vsm.host_model
    n3, n2 = float(N) ** 3, float(N) ** 2
    lods = H.constructCoreOpticalProperties(model, 0)
    nds = [H.get_dtau_ndoubl(np.atleast_1d(lo.tau), np.broadcast_to(np.asarray(lo.varpi), np.atleast_1d(lo.tau).shape),
                             model.quad_points, np.float64, model.numerics)[1] for lo in lods]
    n1 = np.arange(S)
    kin = sum(((n1 + int(sh) >= 0) & (n1 + int(sh) < S)).astype(float) for sh in shifts).mean()     # in-band lines per recipient
    per_m = sum(nd * (12 * n3 + 8 * n2 + kin * (32 * n3 + 20 * n2)) for nd in nds) + L * (24 * n3 + 8 * n2 + kin * (36 * n3 + 16 * n2))
    flops_pt = 3 * per_m
    tf = flops_pt * S / (t2 - t1) / 1e12
    print("  ndoubl per layer %s, %.1f in-band lines per recipient: algorithmic %.2f GFLOP/point -> %.1f TFLOP/s = %.3f of the FP64 MFMA "
          "peak (78.6 TF)" % (nds, kin, flops_pt / 1e9, tf, tf / 78.6))
    if a.json:
        import json
        print(json.dumps({
            "metric": "spectral-points/s for rt_run(RRS), C5 shape (BASELINE.json configs[4])", "value": S / (t2 - t1),
            "unit": "spectral-points/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * (t2 - t1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C5: rotational Raman, nStokes=3, N=%d, %d layers, %d spectral points, %d Raman lines, m=0..2"
                                   % (N, L, S, len(shifts)),
                       "timed_step": "rt_run(RS_type::RRS, model, 1) incl. host optics + H2D + D2H (second call)",
                       "algorithmic_gflop_per_point": flops_pt / 1e9, "in_band_lines_per_point": kin},
            "roofline": {"bound": "mfma", "kernel": "whole run (k_raman_doubling_wave_sp 72 %, k_raman_interaction_wave 16 %)",
                         "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": None}}))
    if a.oracle_points:
        from oracle import vsm_oracle as O
        from oracle import vsm_oracle_raman as OR
        n = a.oracle_points
        om = O.build_model("IQU", a.l_trunc, 40.0, [30.0], [0.0], tau_rayl=tau_rayl[:n], tau_abs=tau_abs[:n], depol=0.0075,
                           albedo=0.05, m_max=2)
        om.varpi_cabannes = 0.96
        t0 = time.perf_counter()
        OR.rt_run_rrs(om, OR.RRS(i_shift=shifts, varpi_ie=w_ie, greek_raman=O.get_greek_rayleigh(0.75)))
        dt = time.perf_counter() - t0
        print("  numpy oracle on the first %d points: %.2f s -> %.1f spectral-points/s (1 process)" % (n, dt, n / dt))


if __name__ == "__main__":
    main()
