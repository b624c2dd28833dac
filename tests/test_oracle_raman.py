"""Validation of the Raman (RRS) oracle, CPU only.

PARITY UNPINNED for Raman (see oracle/vsm_oracle_raman.py).  The check used here is an exact property of the
restated equations: the inelastic recurrences are the first-order perturbation of the elastic ones in the
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
