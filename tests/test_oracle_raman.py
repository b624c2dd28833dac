"""Validation of the Raman (RRS) oracle, CPU only.

Two legs: (1) the reference's own quantitative RRS regression (test/reference/phase1b_RRS_sanghavi_q0.jld2 ->
tests/golden/phase1b_rrs_sanghavi_q0.json) at the reference's gate, which PINS the oracle (bottom of this file);
(2) an exact property of the restated equations: the inelastic recurrences are the first-order perturbation of the elastic ones in the
single-scattering albedo.  On a spectrally uniform atmosphere, with the Raman phase matrix equal to the
elastic one and fScattRayleigh = tau_rayl / tau, the Raman field at spectral point n1 must equal
    d(elastic field)/d(varpi_Cabannes) * Sum_{dn : n1 + shift[dn] in band} varpi_ie[dn].
"""
import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_raman as OR


def _uniform_model(S, L, pol="IQU", l_trunc=7, varpi_cab=0.96, albedo=0.1, FT=np.float64, m_max=2):
    tau_rayl = np.tile(np.linspace(0.02, 0.05, L), (S, 1))
    tau_abs = np.tile(np.linspace(0.03, 0.01, L), (S, 1))
    model = O.build_model(pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 60.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03,
                          albedo=albedo, m_max=m_max, FT=FT)
    model.varpi_cabannes = varpi_cab
    return model


def test_get_n0_n1_matches_reference_ranges():
    # inelastic_helper.jl:19-26 with nSpec = 10 (1-based there, 0-based slices here)
    assert OR.get_n0_n1(10, 3) == (slice(3, 10), slice(0, 7))
    assert OR.get_n0_n1(10, -2) == (slice(0, 8), slice(2, 10))
    assert OR.get_n0_n1(10, 0) == (slice(0, 10), slice(0, 10))
    n0, n1 = OR.get_n0_n1(4, 7)
    assert n1.stop <= n1.start


@pytest.mark.parametrize("pol", ["I", "IQU"])
def test_raman_is_first_order_perturbation_of_elastic(pol):
    S, L = 12, 3
    model = _uniform_model(S, L, pol=pol)
    shifts = np.array([-3, -1, 2, 5, -20])          # -20: never in band
    w_ie = np.array([0.011, 0.02, 0.007, 0.013, 0.5])
    rs = OR.RRS(i_shift=shifts, varpi_ie=w_ie, greek_raman=model.greek_rayleigh)
    fscatt = model.tau_rayl / (model.tau_rayl + model.tau_abs)
    R, T, ieR, ieT = OR.rt_run_rrs(model, rs, fscatt=fscatt)
    # the elastic part is untouched by the Raman bookkeeping
    R0, T0 = O.rt_run(model)
    assert np.max(np.abs(R - R0)) <= 1e-14 * np.max(np.abs(R0))
    assert np.max(np.abs(T - T0)) <= 1e-14 * np.max(np.abs(T0))
    # central finite difference in varpi_Cabannes
    h = 1e-5
    wc = model.varpi_cabannes
    model.varpi_cabannes = wc + h
    Rp, Tp = O.rt_run(model)
    model.varpi_cabannes = wc - h
    Rm, Tm = O.rt_run(model)
    model.varpi_cabannes = wc
    dR, dT = (Rp - Rm) / (2 * h), (Tp - Tm) / (2 * h)
    wsum = np.array([sum(w for s, w in zip(shifts, w_ie) if 0 <= n1 + s < S) for n1 in range(S)])
    expR, expT = dR * wsum[None, None, :], dT * wsum[None, None, :]
    assert np.max(np.abs(ieR - expR)) <= 2e-7 * np.max(np.abs(expR))
    assert np.max(np.abs(ieT - expT)) <= 2e-7 * np.max(np.abs(expT))
    assert np.max(np.abs(ieR)) > 1e-4  # not vacuous


def test_raman_shift_moves_spectral_structure():
    """A spectrally varying source F0: the Raman field at n1 scales with F0 at the donor point n0 = n1 + shift."""
    S, L = 10, 2
    model = _uniform_model(S, L, pol="I", m_max=0, albedo=0.0)   # (the Lambertian source uses I0, not F0)
    F0 = np.zeros((1, S))
    F0[0, :] = 1.0
    rs = OR.RRS(i_shift=np.array([2]), varpi_ie=np.array([0.01]), greek_raman=model.greek_rayleigh)
    fscatt = model.tau_rayl / (model.tau_rayl + model.tau_abs)
    model.F0 = F0
    _, _, ieR_a, _ = OR.rt_run_rrs(model, rs, fscatt=fscatt)
    F0b = F0.copy()
    F0b[0, 7] = 3.0
    model.F0 = F0b
    _, _, ieR_b, _ = OR.rt_run_rrs(model, rs, fscatt=fscatt)
    # only the recipient n1 = 5 (donor 7) changes, by the factor 3
    chg = np.abs(ieR_b - ieR_a).max(axis=(0, 1))
    assert np.all(chg[[0, 1, 2, 3, 4, 6, 7]] <= 1e-15)
    assert np.allclose(ieR_b[:, :, 5], 3.0 * ieR_a[:, :, 5], rtol=1e-12)
    assert np.all(ieR_a[:, :, 8:] == 0)  # donors out of band


# ---------------------------------------------------------------------------
# The reference's quantitative RRS regression (test/test_forward_raman_phase1b.jl): pins the Raman oracle
# ---------------------------------------------------------------------------
def phase1b_scene(golden_dir):
    """Scene of test_parameters/Phase1b_RRS_761-764nm.yaml, built by the host-side input producers
    (vsmartmom.jl_amd/raman_inputs.py: N2/O2 constants -> Raman lines on the band grid, Cabannes optics, Bodhaine tau)."""
    import json
    import os
    import vsmartmom_jl_amd as V
    with open(os.path.join(golden_dir, "phase1b_rrs_sanghavi_q0.json")) as f:
        fx = json.load(f)
    nu = fx["nu_start"] + fx["nu_step"] * np.arange(int(np.floor((fx["nu_stop"] - fx["nu_start"]) / fx["nu_step"])) + 1)
    bs = V.raman_inputs.rrs_band_setup(nu, fx["T"], fx["p"], fx["q"], fx["profile_reduction"], fx["depol"])
    return fx, bs


def check_phase1b(fx, got, tight=None):
    """isapprox(got, ref; atol, rtol) per pixel as test_forward_raman_phase1b.jl:84-100; `tight` adds our own bound."""
    worst = {}
    for name, a in zip(("R_rrs", "T_rrs", "ieR", "ieT"), got):
        ref = np.asarray(fx[name])
        a = np.asarray(a, dtype=np.float64)
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        for ip in range(ref.shape[1]):
            d = np.abs(a[:, ip] - ref[:, ip])
            rtol = fx["rtol"] if ip < 2 else 0.0
            tol = np.maximum(fx["atol"], rtol * np.maximum(np.abs(a[:, ip]), np.abs(ref[:, ip])))
            assert np.all(d <= tol), (name, ip, float(d.max()))
        worst[name] = float(np.max(np.abs(a[:, :2] - ref[:, :2]) / np.abs(ref[:, :2])))
    if tight:
        for k, v in tight.items():
            assert worst[k] <= v, (k, worst[k])
    return worst


def test_phase1b_rrs_inputs(golden_dir):
    fx, bs = phase1b_scene(golden_dir)
    assert len(bs.nu) == np.asarray(fx["ieR"]).shape[2] == 103
    assert bs.tau_rayl.shape == (103, 12)
    # N2 J=0,1 and O2 J=1 Stokes/anti-Stokes pairs fall inside the 51 cm^-1 band; each is split over two grid points
    assert list(bs.i_shift) == [-40, -39, -29, -28, -24, -23, 23, 24, 28, 29, 39, 40]
    assert abs(bs.varpi_ie.sum() - (1 - bs.varpi_cabannes_model)) < 1e-15
    assert 0.96 < bs.varpi_cabannes_rs < 0.97 and 0.007 < bs.depol_cabannes < 0.0075 and 0.028 < bs.depol_rayleigh < 0.0285
    assert abs(bs.greek_raman["beta"][2] - 0.5 * (1 - 6 / 7) / (1 + 3 / 7)) < 1e-15


def test_phase1b_rrs_golden_pins_the_oracle(golden_dir):
    """rt_run(RRS) of the oracle against the reference's stored R, T, ieR, ieT at the reference's own gate
    (atol 1e-6, rtol 0.02); observed: R 2.4e-4, ieR 9e-4, ieT 1e-2, T 1.4e-2 (the stored T carries FP32 noise)."""
    fx, bs = phase1b_scene(golden_dir)
    model = O.build_model("IQU", 2 * fx["nstreams"] - 1, fx["sza"], fx["vza"], fx["vaz"], bs.tau_rayl,
                          depol=bs.depol_cabannes, albedo=fx["albedo"], m_max=2)
    model.varpi_cabannes = bs.varpi_cabannes_rs
    rs = OR.RRS(i_shift=bs.i_shift, varpi_ie=bs.varpi_ie, greek_raman=O.greek_from_dict(bs.greek_raman))
    got = OR.rt_run_rrs(model, rs)
    check_phase1b(fx, got, tight=dict(R_rrs=1e-3, ieR=3e-3))
