"""RPV / Ross-Li BRDF surfaces of the oracle: exact Lambertian limits (which tie them to the golden-pinned Lambertian path),
reciprocity, and hand-evaluated kernel values."""
import math

import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_brdf as OB


def _model(pol="I", albedo=0.0):
    tau_rayl = np.array([[0.02, 0.05, 0.1]] * 2)
    tau_abs = np.array([[0.0, 0.01, 0.02], [0.3, 0.1, 0.05]])
    return O.build_model(pol, 9, 40.0, [60.0, 30.0, 0.0], [180.0, 0.0, 0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03,
                         albedo=albedo, m_max=4)


@pytest.mark.parametrize("surf,alb", [(OB.RPVSurface(0.12, 1.0, 1.0, 0.0), 0.12), (OB.RossLiSurface(0.0, 0.0, 0.2), 0.2)])
def test_lambertian_limits(surf, alb):
    """rpvSurfaceScalar(rho0, 1, 1, 0) has M = F = H = 1 and RossLiSurfaceScalar(0, 0, fiso) keeps K_iso only: both are Lambertian
    surfaces of albedo rho0 / fiso, so the BRDF driver must reproduce rt_run with LambertianSurfaceScalar (moments m > 0 of the
    constant BRDF vanish to quadrature accuracy)."""
    Rb, Tb = OB.rt_run(_model(), surf)
    Rl, Tl = O.rt_run(_model(albedo=alb))
    assert np.max(np.abs(Rb - Rl)) < 1e-12 * np.max(np.abs(Rl)) and np.max(np.abs(Tb - Tl)) < 1e-12 * np.max(np.abs(Tl))


@pytest.mark.parametrize("surf", [OB.RPVSurface(0.12, 0.08, 0.75, -0.25), OB.RossLiSurface(0.05, 0.03, 0.10)])
def test_fourier_blocks_are_reciprocal_and_scalar(surf):
    """rho(mu_i, mu_r) = rho(mu_r, mu_i) for both models => symmetric Fourier blocks; only the I -> I element is filled."""
    mu = np.array([0.1, 0.35, 0.6, 0.9])
    for m in (0, 1, 3):
        R = OB.reflectance(surf, 3, mu, m)
        assert np.allclose(R[0::3, 0::3], R[0::3, 0::3].T, rtol=1e-12, atol=1e-15)
        assert np.all(R[1::3] == 0) and np.all(R[:, 1::3] == 0) and np.all(R[2::3] == 0)
    assert np.max(np.abs(OB.reflectance(surf, 1, mu, 0))) > 0


def test_kernel_values_by_hand():
    """Pointwise values against the formulas of rpv_surface.jl:113-150 / rossli_surface.jl:12-98 evaluated by hand."""
    mi, mr, dphi = 0.6, 0.8, 0.7
    rpv = OB.RPVSurface(0.12, 0.08, 0.75, -0.25)
    ti, tr = math.acos(mi), math.acos(mr)
    cosg = -mi * mr + math.sin(ti) * math.sin(tr) * math.cos(dphi)
    G = math.sqrt(math.tan(ti) ** 2 + math.tan(tr) ** 2 + 2 * math.tan(ti) * math.tan(tr) * math.cos(dphi))
    want = 0.12 * (mi * mr) ** (-0.25) / (mi + mr) ** 0.25 * (1 - 0.0625) / (1 + 0.0625 + 2 * 0.25 * cosg) ** 1.5 * (1 + 0.92 / (1 + G))
    assert math.isclose(float(OB.brdf_value(rpv, 1, mi, mr, dphi)), want, rel_tol=1e-14)
    rl = OB.RossLiSurface(0.05, 0.03, 0.10)
    d = math.pi - dphi
    xi = math.acos(mi * mr + math.sin(ti) * math.sin(tr) * math.cos(d))
    kvol = ((math.pi / 2 - xi) * math.cos(xi) + math.sin(xi)) / (mi + mr) - math.pi / 4
    D = math.sqrt(math.tan(ti) ** 2 + math.tan(tr) ** 2 - 2 * math.tan(ti) * math.tan(tr) * math.cos(d))
    ss = 1 / mi + 1 / mr
    ct = min(1.0, 2 * math.sqrt(D ** 2 + (math.tan(ti) * math.tan(tr) * math.sin(d)) ** 2) / ss)
    t = math.acos(ct)
    kgeo = (t - math.sin(t) * math.cos(t)) * ss / math.pi - ss + 0.5 * (1 + math.cos(xi)) / (mi * mr)
    assert math.isclose(float(OB.brdf_value(rl, 1, mi, mr, dphi)), 0.10 + 0.05 * kvol + 0.03 * kgeo, rel_tol=1e-13)
    assert float(OB.brdf_value(rl, 2, mi, mr, dphi)) == 0.0
