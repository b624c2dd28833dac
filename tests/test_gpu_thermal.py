"""Thermal-emission source slot on the device (vsm_thermal_source + the per-source slot pass of rt_run) against the oracle and
the reference's own physics checks (test/test_thermal_emission.jl T-A1, T-A6, T-A7)."""
import ctypes as C

import numpy as np
import pytest

from oracle import vsm_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsm():
    import vsmartmom_jl_amd as v
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    v._lib.lib()
    return v


@pytest.fixture(scope="module")
def arch(vsm):
    return vsm.Architectures.GPU()


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def _models(vsm, arch, sza, tau_abs, sources, pol="IQUV", S=3, L=4, albedo=0.1, FT=np.float64, m_max=2, l_trunc=9):
    tau_rayl = np.tile(np.array([0.02, 0.05, 0.1, 0.2])[:L], (S, 1))
    kw = dict(tau_rayl=tau_rayl, tau_abs=np.full((S, L), tau_abs) * (1 + 0.3 * np.arange(S))[:, None], depol=0.03, albedo=albedo,
              m_max=m_max)
    om = O.build_model(pol, l_trunc, sza, [0.0, 35.0], [0.0, 60.0], FT=FT, **kw)
    pm = vsm.host_model.model_from_arrays(arch, pol, l_trunc, sza, [0.0, 35.0], [0.0, 60.0], float_type=FT, sources=sources, **kw)
    return om, pm


@pytest.mark.parametrize("FT", [np.float64, np.float32])
def test_thermal_source_operator(vsm, arch, FT):
    """vsm_thermal_source_* == contribute!(::PreparedThermalEmission) (thermal_emission.jl:241-301; T-A1)."""
    CR = vsm.CoreRT
    pol = vsm.host_model.polarization_type("IQU")
    qp = vsm.host_model.rt_set_streams(5, 30.0, [10.0], pol, FT)
    opol, oqp = O.polarization("IQU"), O.rt_set_streams_gausslegquad(5, 30.0, [10.0], O.polarization("IQU"), FT)
    dq = CR.device_quad(qp, pol, arch, FT)
    S, N = 4, qp.Nquad * 3
    varpi, dtau, B = np.array([0.0, 0.2, 0.5, 0.8]), np.array([0.1, 0.5, 1.0, 2.0]), np.array([10.0, 20.0, 30.0, 40.0])
    added = CR.make_added_layer(FT, arch, (N, N), S)
    added.j0_p.fill_(7.0)
    conv = vsm.Architectures.array_type(arch)
    q, a = dq.cstruct(), added.cstruct()
    t = [conv(x.astype(FT)) for x in (dtau, varpi, B)]
    vsm._lib.call("vsm_thermal_source", added.dtype, C.byref(q), S, CR._ptr(t[0]), CR._ptr(t[1]), CR._ptr(t[2]), C.byref(a),
                  CR._stream_ptr())
    ref = O.thermal_source(opol, oqp, dtau.astype(FT), varpi.astype(FT), B.astype(FT), FT)
    tol = 1e-14 if FT == np.float64 else 1e-6
    assert _rel(vsm.Architectures.to_host(added.j0_p), ref) < tol and _rel(vsm.Architectures.to_host(added.j0_m), ref) < tol
    assert np.all(vsm.Architectures.to_host(added.j0_p)[:, 1::3] == 0)


@pytest.mark.parametrize("pol,l_trunc", [("I", 9), ("IQUV", 9),        # N = 8, 32: operator level
                                         ("IQUV", 19), ("IQU", 33),     # N = 52, 60: the slot rides in the fused strip layer kernel
                                         ("IQU", 35), ("IQUV", 25),     # N = 63, 64: ... without spare columns (mat-vecs)
                                         ("IQUV", 31)])                 # N = 76: operator-level slot on the strip128 kernels
def test_rt_run_thermal_slot_vs_oracle(vsm, arch, pol, l_trunc):
    """rt_run(model; sources = ThermalEmission) and sources = SolarBeam + ThermalEmission against the oracle's slot pass."""
    H = vsm.host_model
    B = 0.1 + 0.01 * np.arange(4)[:, None] * np.array([1.0, 2.0, 3.0])[None, :]
    om, pm = _models(vsm, arch, 30.0, 0.05, (H.ThermalEmission(B_layer=B),), pol=pol, l_trunc=l_trunc)
    Rt, Tt = O.rt_run_thermal(om, B)
    R, T = vsm.CoreRT.rt_run(pm)
    assert np.max(np.abs(Rt[:, 0])) > 0
    # without a SolarBeam the solar slot runs with F0 = 0; like the reference's, the Lambertian surface layer still carries its
    # beam term (lambertian_surface.jl:67-73 uses pol_type.I0, not F0), so the thermal-only total over a reflecting surface is
    # "solar slot at F0 = 0" + thermal slot
    om0 = O.build_model(pol, l_trunc, 30.0, [0.0, 35.0], [0.0, 60.0], tau_rayl=om.tau_rayl, tau_abs=om.tau_abs, depol=0.03, albedo=om.albedo,
                        m_max=om.m_max)
    om0.F0 = np.zeros((om.pol.n, om.tau_rayl.shape[0]))
    R0, T0 = O.rt_run(om0)
    assert _rel(R, R0 + Rt) < 1e-9 and _rel(T, T0 + Tt) < 1e-9
    _, pm2 = _models(vsm, arch, 30.0, 0.05, (H.SolarBeam(), H.ThermalEmission(B_layer=B)), pol=pol, l_trunc=l_trunc)
    Rs, Ts = O.rt_run(om)
    R2, T2 = vsm.CoreRT.rt_run(pm2)
    assert _rel(R2, Rs + Rt) < 1e-9 and _rel(T2, Ts + Tt) < 1e-9


def test_thermal_is_independent_of_sza_and_opaque_column_is_a_blackbody(vsm, arch):
    """T-A6: the emergent thermal radiance does not depend on the SZA (rel. 1e-6); T-A7: an opaque isothermal column emits
    B(T) (R / B = 1, atol 1e-3) for m_max = 0, 1, 2 (test_thermal_emission.jl:267-400)."""
    H = vsm.host_model
    B = 0.1 + 0.01 * np.arange(4)[:, None] * np.ones((1, 3))
    R30, _ = vsm.CoreRT.rt_run(_models(vsm, arch, 30.0, 0.05, (H.ThermalEmission(B_layer=B),), albedo=0.0)[1])
    R60, _ = vsm.CoreRT.rt_run(_models(vsm, arch, 60.0, 0.05, (H.ThermalEmission(B_layer=B),), albedo=0.0)[1])
    assert np.max(np.abs(R30[:, 0])) > 0 and np.max(np.abs(R30 - R60)) / np.max(np.abs(R30)) < 1e-6
    nu = np.array([800.0, 1000.0, 1200.0])
    for m_max in (0, 1, 2):
        src = (H.ThermalEmission(T_layers=[250.0] * 4, nu=nu),)
        R, _ = vsm.CoreRT.rt_run(_models(vsm, arch, 30.0, 100.0, src, albedo=0.0, m_max=m_max)[1])
        assert np.allclose(R[0, 0, :] / H.planck_spectrum_wn(250.0, nu), 1.0, atol=1e-3), m_max


@pytest.mark.parametrize("FT,geo,tol", [(np.float64, ("IQU", 29, 35.0, [0.0, 50.0], [0.0, 120.0]), 1e-9),     # N = 54
                                        (np.float64, ("IQU", 35, 35.0, [0.0, 50.0], [0.0, 120.0]), 1e-9),     # N = 63 (no spare columns)
                                        (np.float32, ("IQUV", 37, 35.0, [0.0, 50.0], [0.0, 120.0]), 2e-3),   # N = 88
                                        (np.float32, ("IQU", 51, 35.0, [0.0, 50.0], [0.0, 120.0]), 2e-3)])   # N = 87 (N % 4 != 0)
def test_thermal_slot_fused_equals_operator_level(vsm, arch, monkeypatch, FT, geo, tol):
    """The fused thermal slot (vsm_layer_forward_thermal: FP64 32 < N <= 60, FP32 64 < N <= 96) against the operator-level slot
    pass on the same scenes: Rayleigh + absorption with a non-scattering layer in the middle (that layer stays operator level
    inside the fused pass), thick layers (many doublings), and layers with two aerosol types (Z mixed per point inside the
    kernel)."""
    H = vsm.host_model
    rng = np.random.default_rng(3)
    S, L = 5, 4
    B = 0.05 + 0.02 * rng.random((L, S))
    tau_rayl = np.tile(np.array([0.02, 0.0, 0.3, 0.6]), (S, 1))        # layer 2 does not scatter
    tau_abs = 10.0 ** rng.uniform(-2, 0.5, (S, L))
    tau_aer = np.array([[0.0, 0.0, 0.2, 0.1], [0.03, 0.0, 0.1, 0.3]])
    aos = [H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.7, 12))), 0.95, 0.1),
           H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.5, 8))), 0.9, 0.0)]
    N = H.rt_set_streams(geo[1], geo[2], geo[3], H.polarization_type(geo[0]), FT).Nquad * H.polarization_type(geo[0]).n
    assert vsm._lib.lib().vsm_layer_thermal_fused(N, 1 if FT == np.float64 else 0) == 1
    for kw in (dict(tau_rayl=tau_rayl, tau_abs=tau_abs),
               dict(tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer, aerosol_optics=aos)):
        com = dict(depol=0.03, albedo=0.2, m_max=3, float_type=FT, **kw)
        # (the corrected slot variant: with the reference's slot state a column with a non-scattering layer keeps every doubled
        # slot in the AddedLayer and takes no fused step -- test_thermal_slot_state_in_nonscattering_layers)
        model = H.model_from_arrays(arch, *geo, sources=(H.SolarBeam(), H.ThermalEmission(B_layer=B, reset_slot_in_nonscattering_layers=True)),
                                    **com)
        monkeypatch.setattr(vsm.CoreRT, "THERMAL_FUSION", True)
        Rf, Tf = vsm.CoreRT.rt_run(model)
        monkeypatch.setattr(vsm.CoreRT, "THERMAL_FUSION", False)
        Ro, To = vsm.CoreRT.rt_run(model)
        monkeypatch.setattr(vsm.CoreRT, "THERMAL_FUSION", True)
        Rs, Ts = vsm.CoreRT.rt_run(H.model_from_arrays(arch, *geo, **com))
        assert np.max(np.abs(Ro - Rs)) > 1e-4                      # the slot contributes
        assert _rel(Rf - Rs, Ro - Rs) < tol and _rel(Tf - Ts, To - Ts) < tol


@pytest.mark.parametrize("pol,l_trunc", [("I", 9), ("IQUV", 19), ("IQUV", 31)])     # N = 8, 52 (fused-capable shape), 76
@pytest.mark.parametrize("column", ["noscat_in_the_middle", "noscat_on_top", "noscat_on_top_and_below"])
def test_thermal_slot_state_in_nonscattering_layers(vsm, arch, pol, l_trunc, column):
    """rt_kernel.jl:204-232 as written: the `:thermal` slot of the AddedLayer is reset only in the scatter branch, so a
    non-scattering layer interacts with the doubled slot of the last scattering layer before it, and a column that begins with
    non-scattering layers carries the slot of m = 0 into m = 1 -- the default here, against the oracle's restatement; and the
    corrected variant (ThermalEmission(reset_slot_in_nonscattering_layers=True)) against the oracle's."""
    H = vsm.host_model
    S, L = 3, 4
    rayl = {"noscat_in_the_middle": [0.05, 0.0, 0.1, 0.2], "noscat_on_top": [0.0, 0.05, 0.1, 0.2],
            "noscat_on_top_and_below": [0.0, 0.0, 0.1, 0.0]}[column]
    tau_rayl = np.tile(np.array(rayl), (S, 1))
    tau_abs = np.tile(np.array([0.3, 0.5, 0.2, 0.4]), (S, 1)) * (1 + 0.5 * np.arange(S))[:, None]
    B = 0.1 + 0.02 * np.arange(L)[:, None] * np.array([1.0, 2.0, 3.0])[None, :]
    geo = (pol, l_trunc, 30.0, [0.0, 35.0], [0.0, 60.0])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.1, m_max=2)
    om = O.build_model(*geo, **kw)
    Rs, Ts = O.rt_run(om)
    out = {}
    for reset in (False, True):
        Rt, Tt = O.rt_run_thermal(om, B, reset_slot_in_nonscattering_layers=reset)
        pm = H.model_from_arrays(arch, *geo, sources=(H.SolarBeam(), H.ThermalEmission(B_layer=B, reset_slot_in_nonscattering_layers=reset)),
                                 **kw)
        R, T = vsm.CoreRT.rt_run(pm)
        assert _rel(R - Rs, Rt) < 1e-8 and _rel(T - Ts, Tt) < 1e-8, reset
        out[reset] = Rt
    assert np.max(np.abs(out[False] - out[True])) > 1e-6 * np.max(np.abs(out[True]))   # the two variants do differ on these columns
