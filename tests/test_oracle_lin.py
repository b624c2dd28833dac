"""Linearized oracle: analytic Jacobians vs central finite differences of the same forward model
(the reference checks its Jacobians the same way: test/test_jacobians_unit.jl:105-123, albedo max 1e-3 /
mean 1e-4 with a one-sided delta of 1e-4).  Parity of the linearized path is otherwise UNPINNED (no
committed reference numbers exist)."""
import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_lin as OL


def _model(pol, S=3, L=3, albedo=0.2, scale=(1.0, 1.0), seed=0):
    rng = np.random.default_rng(seed)
    tau_rayl = np.tile(0.03 * np.ones(L), (S, 1))
    base_a = 10.0 ** rng.uniform(-2.5, -0.5, (S, L))
    base_b = 10.0 ** rng.uniform(-2.5, -0.5, (S, L))
    base_b[:, 0] = 0.0                      # gas b is absent from the top layer
    tau_abs = scale[0] * base_a + scale[1] * base_b
    mdl = O.build_model(pol, 9, 40.0, [30.0, 5.0], [0.0, 60.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                        albedo=albedo, m_max=2)
    lin = OL.LinModel([base_a, base_b])     # x_g = multiplicative scale of gas g  (d tau_abs / d x_g = base_g)
    return mdl, lin


@pytest.mark.parametrize("pol", ["I", "IQU", "IQUV"])
def test_jacobians_match_finite_differences(pol):
    mdl, lin = _model(pol)
    R, T, Rd, Td = OL.rt_run_lin(mdl, lin)
    h = 1e-5
    for p, kw in enumerate([dict(scale=(1 + h, 1.0)), dict(scale=(1.0, 1 + h)), dict(albedo=0.2 + h)]):
        kwm = {k: (v if k != "scale" else tuple(2 - x for x in v)) for k, v in kw.items()}
        if "albedo" in kw:
            kwm = dict(albedo=0.2 - h)
        Rp, Tp, _, _ = OL.rt_run_lin(*_model(pol, **kw))
        Rm, Tm, _, _ = OL.rt_run_lin(*_model(pol, **kwm))
        fdR, fdT = (Rp - Rm) / (2 * h), (Tp - Tm) / (2 * h)
        for an, fd, name in ((Rd[..., p], fdR, "R"), (Td[..., p], fdT, "T")):
            scale = np.abs(fd).max()
            assert scale > 0
            assert np.abs(an - fd).max() <= 2e-6 * scale, (pol, p, name, np.abs(an - fd).max() / scale)


def test_lin_forward_part_equals_forward_run():
    """The forward R/T of the linearized driver equal rt_run's.  (The linearized Lambertian builder sets
    j0+ = 0 and t-- = 0, lambertian_surface_lin.jl:107,137 vs lambertian_surface.jl:75,85; this only changes
    J0+ at the SZA stream itself and composite operators that feed nothing afterwards.)"""
    mdl, lin = _model("IQU")
    R, T, _, _ = OL.rt_run_lin(mdl, lin)
    R0, T0 = O.rt_run(mdl)
    assert np.allclose(R, R0, rtol=1e-12, atol=1e-15)
    assert np.allclose(T, T0, rtol=1e-12, atol=1e-15)


# ---------------------------------------------------------------------------
# Aerosol Jacobian slots (7 per aerosol: tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p; parameter_layout.jl:28-56)
# ---------------------------------------------------------------------------
def _hg_greek(g, lmax, pol_frac=0.3):
    """A synthetic polarizing aerosol: Henyey-Greenstein beta with Rayleigh-like ratios in the other coefficients."""
    l = np.arange(lmax + 1)
    beta = (2 * l + 1) * g ** l
    alpha = np.where(l >= 2, 0.8 * beta, 0.0)
    gamma = np.where(l >= 2, -pol_frac * beta / (1 + 0.3 * l), 0.0)
    delta = 0.7 * beta * np.where(l >= 1, 1.0, 0.0)
    eps_ = np.where(l >= 2, 0.1 * beta / (1 + l), 0.0)
    zeta = np.where(l >= 2, 0.6 * beta, 0.0)
    return O.GreekCoefs(alpha, beta, gamma, delta, eps_, zeta)


def _aerosol_scene(pol, x=None, albedo=0.2, S=2, L=3, l_trunc=9):
    """A parametric scene: x[0:7] are perturbations of the aerosol's 7 sub-parameters.  The dependence of the aerosol inputs on x
    is LINEAR by construction (tau_aer, omega, f^t and the Greek coefficients move along fixed directions), so the analytic
    inputs `LinModel.tau_aer_dot / lin_aerosol_optics` are exact and central differences of the forward oracle test the chain
    rule of the layer optics (createAero, the `+` rules) and the RT propagation of Zdot."""
    x = np.zeros(7) if x is None else np.asarray(x, dtype=float)
    rng = np.random.default_rng(11)
    lmax = l_trunc
    tau_rayl = np.tile(np.array([0.01, 0.02, 0.04])[:L], (S, 1))
    tau_abs = 10.0 ** rng.uniform(-2.5, -0.7, (S, L))
    g0 = _hg_greek(0.65, lmax)
    gdot = [_hg_greek(0.5 + 0.1 * k, lmax, 0.2 + 0.05 * k) for k in range(4)]
    for k, gd in enumerate(gdot):                       # derivative directions: modest, beta_0 fixed (normalisation)
        for name in ("alpha", "beta", "gamma", "delta", "epsilon", "zeta"):
            setattr(gd, name, 0.05 * (k + 1) * (getattr(gd, name) - getattr(g0, name)))
    tau_aer0 = np.array([[0.0, 0.05, 0.15][:L]])        # absent from the top layer
    tau_aer_dot = np.array([[[0.0, 1.0, 3.0], [0.0, 0.02, -0.05], [0.0, -0.03, 0.02], [0.0, 0.1, 0.2], [0.0, 0.01, 0.04],
                             [0.0, 0.5, -0.5], [0.0, -0.2, 0.3]]])[:, :, :L]
    ssa0, f0 = 0.93, 0.12
    ssa_dot = np.array([0.02, -0.3, 0.05, 0.01])
    f_dot = np.array([0.01, 0.02, 0.2, -0.05])
    greek = O.GreekCoefs(*(getattr(g0, n) + sum(x[1 + k] * getattr(gdot[k], n) for k in range(4))
                           for n in ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")))
    ao = O.AerosolOptics(greek, ssa0 + float(ssa_dot @ x[1:5]), f0 + float(f_dot @ x[1:5]))
    tau_aer = tau_aer0 + np.einsum("k,akl->al", x, tau_aer_dot)
    mdl = O.build_model(pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 120.0], tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer,
                        aerosols=[ao], depol=0.03, albedo=albedo, m_max=4)
    lin = OL.LinModel([tau_abs.copy()], tau_aer_dot=tau_aer_dot, lin_aerosol_optics=[OL.LinAerosolOptics(gdot, ssa_dot, f_dot)])
    return mdl, lin


@pytest.mark.parametrize("pol", ["I", "IQU"])
def test_aerosol_jacobians_match_finite_differences(pol):
    mdl, lin = _aerosol_scene(pol)
    assert lin.n_layer_params == 8
    R, T, Rd, Td = OL.rt_run_lin(mdl, lin)
    R0, T0 = O.rt_run(mdl)
    assert np.allclose(R, R0, rtol=1e-12, atol=1e-15) and np.allclose(T, T0, rtol=1e-12, atol=1e-15)
    h = 1e-5
    for k in range(7):
        e = np.zeros(7)
        e[k] = h
        Rp, Tp = O.rt_run(_aerosol_scene(pol, e)[0])
        Rm, Tm = O.rt_run(_aerosol_scene(pol, -e)[0])
        for an, fd, name in ((Rd[..., k], (Rp - Rm) / (2 * h), "R"), (Td[..., k], (Tp - Tm) / (2 * h), "T")):
            scale = np.abs(fd).max()
            assert scale > 0
            assert np.abs(an - fd).max() <= 5e-6 * scale, (pol, k, name, np.abs(an - fd).max() / scale)
