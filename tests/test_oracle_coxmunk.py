"""Pins oracle/vsm_oracle_coxmunk.py with the reference's own Cox-Munk checks (test/test_coxmunk.jl, testsets 1-10):
analytic known answers + properties at the reference's tolerances.  CPU only."""
import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_coxmunk as CM

NW = complex(1.33, 0.0)


def test_fresnel_coefficients():                      # test_coxmunk.jl:30-59
    rs, rp = CM.fresnel_coefficients(NW, np.float64(1.0))
    assert abs(abs(rs) ** 2 - ((1.33 - 1) / (1.33 + 1)) ** 2) < 1e-10
    assert abs(abs(rs) ** 2 - abs(rp) ** 2) < 1e-12
    rs, rp = CM.fresnel_coefficients(NW, np.float64(np.cos(np.arctan(1.33))))
    assert abs(rp) < 1e-10                            # Brewster
    rs, rp = CM.fresnel_coefficients(NW, np.float64(0.001))
    assert abs(rs) ** 2 > 0.99 and abs(rp) ** 2 > 0.99
    rs, rp = CM.fresnel_coefficients(complex(1.33, 0.01), np.float64(0.5))
    assert np.isfinite(abs(rs)) and np.isfinite(abs(rp))


def test_fresnel_mueller():                           # :65-107
    rs, rp = CM.fresnel_coefficients(NW, np.float64(0.7))
    M = CM.fresnel_mueller(rs, rp, 4)
    for (i, j) in [(0, 2), (0, 3), (1, 2), (1, 3), (2, 0), (2, 1), (3, 0), (3, 1)]:
        assert abs(M[i, j]) < 1e-15
    assert M[0, 1] == M[1, 0] and M[2, 2] == M[3, 3] and M[2, 3] == -M[3, 2]
    assert M[0, 0] >= abs(M[0, 1]) and M[0, 0] <= 1.0
    for n in (1, 2, 3):
        Mn = CM.fresnel_mueller(rs, rp, n)
        assert Mn.shape == (n, n)
        np.testing.assert_allclose(Mn, M[:n, :n], atol=1e-15)
    # complex index: the U-V coupling has the reference's sign pattern M[3,4] = Im(rs rp*), M[4,3] = -Im
    rs, rp = CM.fresnel_coefficients(complex(1.33, 0.2), np.float64(0.4))
    M = CM.fresnel_mueller(rs, rp, 4)
    assert M[2, 3] == (rs * np.conj(rp)).imag and M[3, 2] == -M[2, 3] and M[2, 3] != 0


def test_stokes_rotation():                           # :113-142
    I4 = np.eye(4)
    np.testing.assert_allclose(CM.stokes_rotation_matrix(np.float64(0.0), 4), I4, atol=1e-15)
    np.testing.assert_allclose(CM.stokes_rotation_matrix(np.float64(0.0), 2), np.eye(2), atol=1e-15)
    a, b = 0.3, 0.7
    L = lambda x: CM.stokes_rotation_matrix(np.float64(x), 4)
    np.testing.assert_allclose(L(a) @ L(b), L(a + b), atol=1e-12)
    np.testing.assert_allclose(L(1.2) @ L(-1.2), I4, atol=1e-12)
    np.testing.assert_allclose(L(0.8).T @ L(0.8), I4, atol=1e-12)
    assert L(0.4)[1, 2] > 0 and L(0.4)[2, 1] < 0      # docstring convention [c2 s2; -s2 c2]


def test_water_refractive_index():                    # :148-174
    n550 = CM.water_refractive_index(550.0)
    assert abs(n550.real - 1.333) < 0.005 and n550.imag < 1e-6
    assert n550.real == 1.328 and abs(n550.imag - 5.69e-9) < 1e-20      # a table node: exact
    assert CM.water_refractive_index(1500.0).imag > n550.imag
    assert CM.water_refractive_index(200.0).real > 1.35
    assert np.isfinite(CM.water_refractive_index(100.0).real) and np.isfinite(CM.water_refractive_index(1e4).real)
    mid = CM.water_refractive_index(555.0)
    assert 1.328 >= mid.real >= 1.328 and 5.69e-9 < mid.imag < 6.49e-9   # log-linear between two nodes


def test_helpers():                                   # :180-221
    assert abs(CM.wind_to_sigma2(0.0) - 0.003) < 1e-10
    assert abs(CM.wind_to_sigma2(10.0) - (0.003 + 0.0512)) < 1e-10
    assert CM.whitecap_fraction(0.0) == 0.0
    assert abs(CM.whitecap_fraction(10.0) / (2.95e-6 * 10.0 ** 3.52) - 1) < 1e-6
    assert 0.01 < CM.whitecap_fraction(15.0) < 0.10
    s2 = CM.wind_to_sigma2(5.0)
    Nn = 200
    zmax = 4 * np.sqrt(s2)
    dz = 2 * zmax / Nn
    z = -zmax + (np.arange(Nn) + 0.5) * dz
    assert abs(np.sum(CM.cox_munk_pdf(z[:, None], z[None, :], s2)) * dz * dz - 1.0) < 0.01
    Sz, Sg = CM.shadow_factor(0.95, 0.95, s2), CM.shadow_factor(0.1, 0.1, s2)
    assert Sz > 0.9 and 0 < Sg < Sz


def test_brdf_mueller():                              # :227-298
    surf = CM.CoxMunkSurface(5.0, NW, include_whitecaps=False)
    M = CM.coxmunk_brdf_mueller(surf, 4, 0.6, 0.8, 1.2, NW)
    assert M.shape == (4, 4) and np.all(np.isfinite(M))
    Mr = CM.coxmunk_brdf_mueller(surf, 4, 0.8, 0.6, 1.2, NW)
    assert abs(M[0, 0] / Mr[0, 0] - 1) < 1e-6                              # reciprocity
    lo = CM.coxmunk_brdf_mueller(CM.CoxMunkSurface(1.0, NW, include_whitecaps=False), 1, 0.7, 0.7, 0.0, NW)
    hi = CM.coxmunk_brdf_mueller(CM.CoxMunkSurface(20.0, NW, include_whitecaps=False), 1, 0.7, 0.7, 0.0, NW)
    assert lo[0, 0] > 5 * hi[0, 0]
    Nm, Nf = 30, 60                                                          # energy bound
    mu_out = (np.arange(Nm) + 0.5) / Nm
    ph = (np.arange(Nf) + 0.5) * 2 * np.pi / Nf
    Mh = CM.coxmunk_brdf_mueller(surf, 1, mu_out[:, None], 0.5, ph[None, :], NW)[..., 0, 0]
    assert np.sum(Mh * mu_out[:, None]) / Nm * 2 * np.pi / Nf <= 1.02
    wc = CM.coxmunk_brdf_mueller(CM.CoxMunkSurface(12.0, NW, 0.22, True), 1, 0.3, 0.8, 2.0, NW)
    no = CM.coxmunk_brdf_mueller(CM.CoxMunkSurface(12.0, NW, include_whitecaps=False), 1, 0.3, 0.8, 2.0, NW)
    assert wc[0, 0] != no[0, 0]


def test_fourier_decomposition():                     # :304-364
    surf = CM.CoxMunkSurface(5.0, NW)
    mu = np.array([0.3, 0.5, 0.7, 0.9])
    R0 = CM.reflectance(surf, 1, mu, 0, NW)
    assert R0.shape == (4, 4) and np.all(np.isfinite(R0)) and np.all(R0 >= -1e-10)
    assert np.max(np.abs(CM.reflectance(surf, 1, mu, 10, NW))) / 2 < np.max(np.abs(R0))
    R4 = CM.reflectance(surf, 4, mu, 0, NW)
    assert R4.shape == (16, 16) and np.all(np.isfinite(R4))
    assert np.max(np.abs(R4[1::4, 0::4])) > 1e-12                           # I -> Q coupling present
    for n, sz in ((3, 12), (2, 8)):
        Rn = CM.reflectance(surf, n, mu, 0, NW)
        assert Rn.shape == (sz, sz) and np.all(np.isfinite(Rn))
    np.testing.assert_allclose(R4[0::4, 0::4], R0, rtol=1e-13)               # the I-I block does not depend on n


def test_stokes_convention():                         # :370-393
    surf = CM.CoxMunkSurface(3.0, NW, include_whitecaps=False)
    M = CM.coxmunk_brdf_mueller(surf, 4, 0.7, 0.7, 0.0, NW)
    assert M[0, 0] > 0 and abs(M[0, 1] - M[1, 0]) < 1e-10 and abs(M[0, 1]) / M[0, 0] > 0.01


def test_jacobian_fd_nonzero():                       # :399-426
    mu = np.array([0.3, 0.5, 0.7])
    e = 1e-4
    Rp = CM.reflectance(CM.CoxMunkSurface(5.0 + e, NW), 1, mu, 0, NW)
    Rm = CM.reflectance(CM.CoxMunkSurface(5.0 - e, NW), 1, mu, 0, NW)
    assert np.max(np.abs((Rp - Rm) / (2 * e))) > 1e-6


@pytest.mark.parametrize("U,wc,geoms", [(5.0, True, [(0.6, 0.8, 1.2), (0.3, 0.9, 0.5), (0.7, 0.7, 0.01)]),
                                        (8.0, False, [(0.5, 0.6, 0.8)])])
def test_pointwise_derivative_vs_fd(U, wc, geoms):    # :434-483, rtol 1e-3
    e = 1e-5
    for (a, b, c) in geoms:
        s = lambda u: CM.CoxMunkSurface(u, NW, include_whitecaps=wc)
        M, dM = CM.coxmunk_brdf_mueller_and_deriv(s(U), 4, a, b, c, NW)
        np.testing.assert_allclose(M, CM.coxmunk_brdf_mueller(s(U), 4, a, b, c, NW), atol=1e-14)
        fd = (CM.coxmunk_brdf_mueller(s(U + e), 4, a, b, c, NW) - CM.coxmunk_brdf_mueller(s(U - e), 4, a, b, c, NW)) / (2 * e)
        for i in range(4):
            for j in range(4):
                if abs(fd[i, j]) > 1e-15:
                    assert abs(dM[i, j] / fd[i, j] - 1) < 1e-3
                else:
                    assert abs(dM[i, j]) < 1e-10


def test_fourier_derivative_vs_fd():                  # :485-509, 1 % of the maximum element
    mu = np.array([0.3, 0.5, 0.7])
    for (U, wc) in [(5.0, True), (10.0, False), (3.0, True)]:
        s = lambda u: CM.CoxMunkSurface(u, NW, include_whitecaps=wc)
        dU = max(1e-4, 1e-4 * U)
        for m in (0, 1, 3):
            for n in (1, 4):
                _, dR = CM.reflectance_and_deriv(s(U), n, mu, m, NW)
                fd = (CM.reflectance(s(U + dU), n, mu, m, NW) - CM.reflectance(s(U - dU), n, mu, m, NW)) / (2 * dU)
                mx = np.max(np.abs(fd))
                if mx > 1e-12:
                    assert np.max(np.abs(dR - fd)) / mx < 0.01
                else:
                    assert np.max(np.abs(dR)) < 1e-10


def test_derivative_helpers():                        # :511-533
    s2, e = 0.03, 1e-7
    fd = (CM.cox_munk_pdf(0.1, -0.05, s2 + e) - CM.cox_munk_pdf(0.1, -0.05, s2 - e)) / (2 * e)
    assert abs(CM.cox_munk_pdf_dsigma2(0.1, -0.05, s2) / fd - 1) < 1e-5
    # Julia's isapprox(a, b; rtol): |a-b| <= rtol max(|a|,|b|).  At the reference's own point (0.6, 0.7) its Lambda (written with
    # sqrt(2 pi) where Smith has sqrt(pi)) is clamped to 0, so both sides are 0 there; grazing streams exercise the formula
    for (a, b) in ((0.6, 0.7), (0.05, 0.08), (0.02, 0.3)):
        fd = (CM.shadow_factor(a, b, s2 + e) - CM.shadow_factor(a, b, s2 - e)) / (2 * e)
        an = CM.shadow_factor_dsigma2(a, b, s2)
        assert abs(an - fd) <= 1e-4 * max(abs(an), abs(fd))
    assert CM.shadow_factor_dsigma2(0.05, 0.08, s2) < 0
    fd = (CM.whitecap_fraction(7.0 + 1e-6) - CM.whitecap_fraction(7.0 - 1e-6)) / 2e-6
    assert abs(CM.whitecap_fraction_deriv(7.0) / fd - 1) < 1e-5


# ---- the C3 scene (config/ocean_coxmunk.yaml geometry) through the oracle drivers -----------------------------
def _c3_model(S=2, L=4, nstreams=5, m_max=None, with_abs=True):
    rng = np.random.default_rng(3)
    vza, vaz = [60, 45, 30, 15, 0, 15, 30, 45, 60], [180, 180, 180, 180, 0, 0, 0, 0, 0]
    tau_rayl = np.tile(np.full(L, 0.12 / L), (S, 1))
    tau_abs = rng.uniform(0.0, 0.05, (S, L)) if with_abs else None
    lt = 2 * nstreams - 1
    return O.build_model("IQUV", lt, 30.0, vza, vaz, tau_rayl, tau_abs=tau_abs, depol=0.03, m_max=lt if m_max is None else m_max)


def test_rt_run_coxmunk_properties():
    """Forward driver: finite, |Q|,|U|,|V| <= I, glint side brighter than the anti-glint side, TMS changes only R."""
    model = _c3_model()
    surf = CM.CoxMunkSurface(5.0)
    R, T = CM.rt_run(model, surf)
    R0, T0 = CM.rt_run(model, surf, ss_correction=False)
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(T))
    # physical bounds hold for the Fourier-summed field; the reference's TMS term (restated literally: weight 1/2 AND ff = 1 on
    # m = 0, coxmunk_surface.jl:507-517,563) is added on top and is not bounded that way
    assert np.all(R0[:, 0] > 0) and np.all(np.abs(R0[:, 1:]) <= R0[:, :1] + 1e-12)
    np.testing.assert_array_equal(T, T0)
    assert np.max(np.abs(R - R0)) > 0
    # vaz = 0 is the forward (specular) side in this convention: vza 30 at vaz 0 (index 6) sees the glint of sza 30
    assert R0[6, 0, 0] > R0[2, 0, 0] and R[6, 0, 0] > R[2, 0, 0]
    # the correction is  mu0 exp(-tau/mu0) c[iv, k]  with c independent of the spectral point
    c = CM.ss_correction_coefficients(surf, model.pol, model.vza, model.vaz, model.quad_points.mu0, model.m_max)
    tau = (model.tau_rayl + model.tau_abs).sum(axis=1)
    mu0 = model.quad_points.mu0
    np.testing.assert_allclose(R - R0, c[:, :, None] * (mu0 * np.exp(-tau / mu0))[None, None, :], rtol=0, atol=1e-15)


def test_rt_run_lin_coxmunk_vs_fd():
    """Linearized driver: wind-speed and gas Jacobians vs central differences of the forward lin driver (whose R has
    the t-- = 0 quirk, so the forward of the SAME driver is differenced), at the reference's albedo-style gate
    (test_jacobians_unit.jl:105-123: max 1e-3, mean 1e-4 relative)."""
    from oracle import vsm_oracle_lin as OL
    model = _c3_model(S=2, L=3, nstreams=4)
    S, L = model.tau_rayl.shape
    g = np.random.default_rng(5).uniform(0.5, 1.5, (S, L)) * 0.02
    lin = OL.LinModel([g])
    U = 5.0
    R, T, Rd, Td = CM.rt_run_lin(model, lin, CM.CoxMunkSurface(U))
    assert Rd.shape == R.shape + (2,)
    e = 1e-4
    Rp = CM.rt_run_lin(model, lin, CM.CoxMunkSurface(U + e))[:2]
    Rm = CM.rt_run_lin(model, lin, CM.CoxMunkSurface(U - e))[:2]
    for k, (an, p, m_) in enumerate(((Rd, Rp[0], Rm[0]), (Td, Rp[1], Rm[1]))):
        fd = (p - m_) / (2 * e)
        rel = np.abs(an[..., 1] - fd) / np.max(np.abs(fd))
        assert rel.max() < 1e-3 and rel.mean() < 1e-4, (k, rel.max(), rel.mean())
    import copy
    h = 1e-4
    mp, mm = copy.deepcopy(model), copy.deepcopy(model)
    mp.tau_abs = model.tau_abs + h * g
    mm.tau_abs = model.tau_abs - h * g
    Rp = CM.rt_run_lin(mp, lin, CM.CoxMunkSurface(U))[:2]
    Rm = CM.rt_run_lin(mm, lin, CM.CoxMunkSurface(U))[:2]
    for an, p, m_ in ((Rd, Rp[0], Rm[0]), (Td, Rp[1], Rm[1])):
        fd = (p - m_) / (2 * h)
        rel = np.abs(an[..., 0] - fd) / np.max(np.abs(fd))
        assert rel.max() < 1e-3 and rel.mean() < 1e-4, (rel.max(), rel.mean())
