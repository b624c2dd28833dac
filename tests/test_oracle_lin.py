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
