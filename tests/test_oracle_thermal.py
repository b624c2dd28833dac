"""The thermal-emission source slot of the oracle against the reference's own checks (test/test_thermal_emission.jl):
T-A1 elemental source formula, T-A6 independence of the solar zenith angle, T-A7 R / B = 1 for an opaque isothermal column."""
import numpy as np
import pytest

from oracle import vsm_oracle as O


def test_elemental_thermal_source_formula():
    """T-A1 (test_thermal_emission.jl:100-160): 2 pi (1 - varpi) B (1 - exp(-dtau/mu)) on the I rows, zero on Q/U/V."""
    pol = O.polarization("IQU")
    qp = O.rt_set_streams_gausslegquad(5, 30.0, [10.0], pol, np.float64)
    varpi = np.array([0.0, 0.2, 0.5, 0.8])
    dtau = np.array([0.1, 0.5, 1.0, 2.0])
    B = np.array([10.0, 20.0, 30.0, 40.0])
    j = O.thermal_source(pol, qp, dtau, varpi, B, np.float64)
    for i, mu in enumerate(qp.qp_mu):
        for n in range(4):
            assert np.isclose(j[n, 3 * i], 2 * np.pi * (1 - varpi[n]) * B[n] * (1 - np.exp(-dtau[n] / mu)), rtol=1e-14)
        assert np.all(j[:, 3 * i + 1:3 * i + 3] == 0)


def _scene(sza, tau_abs, pol="IQUV", L=4, S=3, albedo=0.0):
    tau_rayl = np.tile(np.array([0.02, 0.05, 0.1, 0.2])[:L], (S, 1))
    return O.build_model(pol, 9, sza, [0.0, 35.0], [0.0, 60.0], tau_rayl=tau_rayl, tau_abs=np.full((S, L), tau_abs), depol=0.03,
                         albedo=albedo, m_max=2)


def test_thermal_radiance_is_independent_of_the_solar_zenith_angle():
    """T-A6 (test_thermal_emission.jl:267-352): the slot doubles with its own expk = 1, so the emergent thermal radiance at
    fixed viewing angles cannot depend on the SZA the model was built with (rel. difference < 1e-6 in the reference; the
    quadrature node inserted for the SZA changes the streams slightly)."""
    B = 0.1 + 0.01 * np.arange(4)[:, None] * np.ones((1, 3))
    R30, T30 = O.rt_run_thermal(_scene(30.0, 0.05), B)
    R60, T60 = O.rt_run_thermal(_scene(60.0, 0.05), B)
    assert np.max(np.abs(R30[:, 0])) > 0
    assert np.max(np.abs(R30 - R60)) / np.max(np.abs(R30)) < 1e-6
    Rs, _ = O.rt_run(_scene(30.0, 0.05))
    assert not np.allclose(Rs, R30)


@pytest.mark.parametrize("m_max", [0, 1, 2])
def test_opaque_isothermal_column_emits_the_planck_radiance(m_max):
    """T-A7 (test_thermal_emission.jl:354-400): fully opaque isothermal column => R / B(T) = 1 (atol 1e-3), whatever the number
    of Fourier moments (thermal lives at m = 0 only)."""
    mdl = _scene(30.0, 100.0, S=2)
    mdl.m_max = m_max
    nu = np.array([900.0, 1100.0])
    Bp = O.planck_spectrum_wn(250.0, nu)
    R, _ = O.rt_run_thermal(mdl, np.tile(Bp, (4, 1)))
    assert np.allclose(R[0, 0, :] / Bp, 1.0, atol=1e-3)


def test_thermal_slot_state_of_the_reference_in_nonscattering_layers():
    """rt_kernel.jl:204-232: the `:thermal` slot is reset only in the scatter branch.  (1) Without a non-scattering layer the
    restated state and the corrected variant are the same run.  (2) A non-scattering layer below a scattering one re-uses that
    layer's doubled slot: the result equals a run in which the non-scattering layer is replaced by... nothing simpler -- so the
    check is structural: the variants differ, and the difference is confined to m = 0 unless the column BEGINS with a
    non-scattering layer, in which case moment m = 1 (azimuth dependent) carries thermal radiance as well."""
    S, L = 2, 3
    B = 0.1 + 0.05 * np.arange(L)[:, None] * np.ones((1, S))
    geo = ("I", 7, 30.0, [20.0, 20.0], [0.0, 90.0])          # the same viewing zenith at two azimuths
    tau_abs = np.full((S, L), 0.4)

    def run(rayl, reset):
        om = O.build_model(*geo, tau_rayl=np.tile(np.array(rayl), (S, 1)), tau_abs=tau_abs, depol=0.0, albedo=0.0, m_max=2)
        return O.rt_run_thermal(om, B, reset_slot_in_nonscattering_layers=reset)[0]
    a, b = run([0.05, 0.1, 0.2], False), run([0.05, 0.1, 0.2], True)
    assert np.array_equal(a, b)
    a, b = run([0.05, 0.0, 0.2], False), run([0.05, 0.0, 0.2], True)
    assert np.max(np.abs(a - b)) > 1e-4 * np.max(np.abs(b))
    assert np.max(np.abs(a[0] - a[1])) < 1e-14 and np.max(np.abs(b[0] - b[1])) < 1e-14     # m = 0 only: no azimuth dependence
    a, b = run([0.0, 0.05, 0.2], False), run([0.0, 0.05, 0.2], True)
    assert np.max(np.abs(b[0] - b[1])) < 1e-14
    assert np.max(np.abs(a[0] - a[1])) > 1e-6 * np.max(np.abs(a))                           # the slot of m = 0 rides into m = 1
