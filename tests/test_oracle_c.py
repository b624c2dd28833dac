"""The C + OpenMP restatement (oracle/vsm_oracle_c.c: CPU baseline of bench.py) against the pinned numpy oracle and, through
it, against the reference's golden tables.  CPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import vsm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def OC():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import vsm_oracle_c
    vsm_oracle_c.lib()
    return vsm_oracle_c


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("pol,l_trunc", [("I", 5), ("IQU", 9), ("IQUV", 7)])
def test_c_restatement_equals_numpy_oracle(OC, pol, l_trunc):
    """Multi-layer Rayleigh + absorption scene with thin and thick layers (ndoubl 0 .. ~12), Lambertian surface, m = 0..2,
    1 and several threads: same result as the numpy oracle to rounding (different LU / summation order)."""
    rng = np.random.default_rng(7)
    S, L = 7, 4
    tau_rayl = np.tile(np.array([1e-7, 0.02, 0.2, 0.05]), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-4, 0.5, (S, L))
    tau_abs[:, 0] = 0.0
    F0 = np.zeros((O.polarization(pol).n, S))
    F0[0] = 0.7 + 0.6 * rng.random(S)
    m = O.build_model(pol, l_trunc, 40.0, [30.0, 0.0, 55.0], [0.0, 10.0, 170.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03,
                      albedo=0.25, m_max=2)
    m.F0 = F0
    tr = []
    Ro, To = O.rt_run(m, trace=tr)
    assert min(t["ndoubl"] for t in tr) == 0 and max(t["ndoubl"] for t in tr) >= 8
    for nt in (1, 3):
        R, T = OC.rt_run(m, nthreads=nt)
        assert _rel(R, Ro) < 1e-11 and _rel(T, To) < 1e-11, (nt, _rel(R, Ro), _rel(T, To))


def test_c_restatement_meets_the_6sv1_gate(OC, golden_dir):
    """The reference's 6SV1 known-answer table (test/test_CoreRT.jl:7-43: pi R / mu0 within 6e-3) through the C path: Rayleigh +
    Lambertian surface (albedo 0 and 0.25), the surface-interaction leg -- a pin of the C code that does not go through numpy."""
    fx = json.load(open(os.path.join(golden_dir, "sixsv1.json")))
    p = fx["procedure"]
    Rt = np.array(fx["R_trues"])
    n = 0
    for ci, c in enumerate(p["cases"]):
        for si, sza in enumerate(c["sza_deg"]):
            for ai, az in enumerate(p["azimuths_deg"]):
                mdl = O.build_model("IQUV", 2 * p["nstreams"] - 1, sza, p["vza_deg"], [az] * 16, tau_rayl=[[c["tau"]]],
                                    depol=p["depol"], albedo=c["albedo"], m_max=2)
                R, _ = OC.rt_run(mdl, nthreads=2)
                mod = np.pi * R[:, 0, 0] / mdl.quad_points.mu0
                assert np.max(np.abs(Rt[ci, si, ai] - mod) / Rt[ci, si, ai]) < p["rtol"], (ci, sza, az)
                n += 1
    assert n == len(p["cases"]) * 3 * len(p["azimuths_deg"])
