"""GPU parity of the Cox-Munk surface path (BASELINE config C3: config/ocean_coxmunk.yaml, forward + linearized) against
oracle/vsm_oracle_coxmunk.py (pinned by the reference's test_coxmunk.jl checks, tests/test_oracle_coxmunk.py).

Tolerances: FP64 1e-9 relative to each array's maximum for the surface kernels (the reference's rotation angles are
acos(c) with c -> +-1 near the principal plane, which amplifies the few-ulp differences between the device's and numpy's
transcendentals to ~1e-10), 1e-8 for rt_run end to end (the bar of the other rt_run tests), FP32 5e-4
for the reflectance blocks; analytic Jacobians vs central differences of the device's own forward at the reference's gate
(test/test_jacobians_unit.jl:105-123: max 1e-3, mean 1e-4 relative)."""
import copy
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_coxmunk as CM
from oracle import vsm_oracle_lin as OL

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


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
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _surfs(vsm, **kw):
    return vsm.host_model.CoxMunkSurface(**kw), CM.CoxMunkSurface(**kw)


def _quad(vsm, arch, pol_name, l_trunc, FT, sza=30.0, vza=(60, 45, 30, 15, 0)):
    H = vsm.host_model
    pol = H.polarization_type(pol_name)
    qp = H.rt_set_streams(l_trunc, sza, list(vza), pol, FT)
    return pol, qp, vsm.CoreRT.device_quad(qp, pol, arch, FT)


@pytest.mark.parametrize("FT,tol", [(np.float64, 1e-9), (np.float32, 5e-4)])
@pytest.mark.parametrize("pol_name", ["I", "IQ", "IQU", "IQUV"])
@pytest.mark.parametrize("kw", [dict(wind_speed=5.0), dict(wind_speed=12.0, n_water=complex(1.34, 0.01), whitecap_albedo=0.3),
                                dict(wind_speed=2.0, include_whitecaps=False, shadowing=False)])
def test_coxmunk_reflectance_and_deriv(vsm, arch, FT, tol, pol_name, kw):
    """vsm_coxmunk_reflectance == reflectance_and_deriv (coxmunk_surface.jl:381-460) for every polarization, with and without
    whitecaps / shadowing, absorbing water, m = 0, 1, 2, 7."""
    hs, os_ = _surfs(vsm, **kw)
    pol, qp, dq = _quad(vsm, arch, pol_name, 9, FT)
    for m in (0, 1, 2, 7):
        rho, drho = vsm.CoreRT.reflectance(hs, dq, m, arch, FT, deriv=True)
        r_o, d_o = CM.reflectance_and_deriv(os_, pol.n, qp.qp_mu.astype(np.float64), m)
        r_d, d_d = vsm.Architectures.to_host(rho).T, vsm.Architectures.to_host(drho).T
        assert _rel(r_d, r_o) < tol, (m, _rel(r_d, r_o))
        assert _rel(d_d, d_o) < tol * (20 if FT == np.float32 else 1), (m, _rel(d_d, d_o))
        rho2, none = vsm.CoreRT.reflectance(hs, dq, m, arch, FT)
        assert none is None and torch.equal(rho2, rho)


def test_reference_checks_on_device(vsm, arch):
    """The reference's own reflectance checks (test_coxmunk.jl:304-364, 485-509) run on the device kernel: scalar m = 0 block
    non-negative, higher moments decay, I->Q coupling present, the I-I block independent of nStokes, analytic derivative
    within 1 % of central differences."""
    FT = np.float64
    mk = lambda U, wc=True: vsm.host_model.CoxMunkSurface(wind_speed=U, n_water=complex(1.33, 0.0), include_whitecaps=wc)
    pol1, qp1, dq1 = _quad(vsm, arch, "I", 7, FT)
    pol4, qp4, dq4 = _quad(vsm, arch, "IQUV", 7, FT)
    get = lambda s, dq, m, d=False: [None if x is None else vsm.Architectures.to_host(x).T for x in
                                     vsm.CoreRT.reflectance(s, dq, m, arch, FT, deriv=d)]
    R0 = get(mk(5.0), dq1, 0)[0]
    assert np.all(np.isfinite(R0)) and np.all(R0 >= -1e-10)
    assert np.max(np.abs(get(mk(5.0), dq1, 10)[0])) / 2 < np.max(np.abs(R0))
    R4 = get(mk(5.0), dq4, 0)[0]
    assert np.max(np.abs(R4[1::4, 0::4])) > 1e-12
    assert _rel(R4[0::4, 0::4], R0) < 1e-12
    for (U, wc) in [(5.0, True), (10.0, False), (3.0, True)]:
        dU = max(1e-4, 1e-4 * U)
        for m in (0, 1, 3):
            for dq in (dq1, dq4):
                dR = get(mk(U, wc), dq, m, True)[1]
                fd = (get(mk(U + dU, wc), dq, m)[0] - get(mk(U - dU, wc), dq, m)[0]) / (2 * dU)
                assert np.max(np.abs(dR - fd)) / np.max(np.abs(fd)) < 0.01


def _added_host(vsm, a):
    f = lambda t: vsm.Architectures.to_host(t).transpose(0, 2, 1)
    h = vsm.Architectures.to_host
    return dict(r_mp=f(a.r_mp), t_pp=f(a.t_pp), r_pm=f(a.r_pm), t_mm=f(a.t_mm), j0_p=h(a.j0_p), j0_m=h(a.j0_m))


@pytest.mark.parametrize("pol_name", ["I", "IQU", "IQUV"])
def test_brdf_surface_layer_forward_and_lin(vsm, arch, pol_name):
    """vsm_brdf_surface / vsm_brdf_surface_lin == create_surface_layer! (rpv_surface.jl:51-97, coxmunk_surface_lin.jl:27-102),
    including the linearized builder's quirks (j0+ = 0, t-- = 0, F0 with all Stokes components)."""
    FT = np.float64
    CR, CL = vsm.CoreRT, vsm.CoreRTLin
    hs, os_ = _surfs(vsm, wind_speed=6.0)
    pol, qp, dq = _quad(vsm, arch, pol_name, 9, FT)
    opol = O.polarization(pol_name)
    oqp = O.rt_set_streams_gausslegquad(9, 30.0, [60, 45, 30, 15, 0], opol, FT)
    N, S, P, pl = qp.Nquad * pol.n, 5, 3, 2
    rng = np.random.default_rng(11)
    tau_sum = rng.uniform(0.05, 2.0, S)
    tsd = rng.standard_normal((S, pl))
    F0 = rng.uniform(0.2, 1.0, (pol.n, S))
    conv = vsm.Architectures.array_type(arch)
    for m in (0, 2):
        rho, drho = CR.reflectance(hs, dq, m, arch, FT, deriv=True)
        r_o, d_o = CM.reflectance_and_deriv(os_, pol.n, qp.qp_mu.astype(np.float64), m)
        # forward builder
        a = CR.AddedLayer(FT, arch, N, S, shared=True)
        for t in (a.r_mp, a.t_pp, a.r_pm, a.t_mm, a.j0_p, a.j0_m):
            t.fill_(float("nan"))
        CR.create_surface_layer_(hs, a, m, dq, conv(tau_sum), rho=rho)
        oa = O.make_added_layer(FT, N, S)
        CM.create_surface_layer_brdf(r_o, oa, m, opol, oqp, tau_sum, FT)
        d = _added_host(vsm, a)
        for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
            assert _rel(d[k][0], getattr(oa, k)[0]) < 1e-9 or np.max(np.abs(getattr(oa, k)[0])) == 0 and np.all(d[k][0] == 0), (m, k)
        # source vectors pick ONE column of rho (the SZA stream, whose specular pair carries the ill-conditioned rotation
        # angles): their error is 1e-9 of max|rho|, i.e. a larger fraction of the column's own maximum
        for k in ("j0_p", "j0_m"):
            assert _rel(d[k], getattr(oa, k)) < 1e-8, (m, k)
        # same through the dispatch without a precomputed block
        a2 = CR.AddedLayer(FT, arch, N, S, shared=True)
        CR.create_surface_layer_(hs, a2, m, dq, conv(tau_sum), arch=arch, FT=FT)
        assert torch.equal(a2.r_mp, a.r_mp) and torch.equal(a2.j0_m, a.j0_m)
        # linearized builder
        a = CR.AddedLayer(FT, arch, N, S, shared=True)
        al = CL.AddedLayerLin(FT, arch, P, N, S, shared=True)
        for t in (a.r_mp, a.t_pp, a.r_pm, a.t_mm, a.j0_p, a.j0_m, al.ap_r_mp, al.ap_t_pp, al.ap_r_pm, al.ap_t_mm, al.ap_J0_p,
                  al.ap_J0_m):
            t.fill_(float("nan"))
        q_, a_, al_ = dq.cstruct(), a.cstruct(), al.cstruct()
        d_ts, d_tsd, d_F0 = conv(tau_sum), CL.to_device_sp(tsd, arch, FT), conv(np.ascontiguousarray(F0.T))   # keep alive
        vsm._lib.call("vsm_brdf_surface_lin", torch.float64, C.byref(q_), S, m, CR._ptr(rho), CR._ptr(drho), P - 1,
                      CR._ptr(d_ts), CR._ptr(d_tsd), pl, CR._ptr(d_F0), C.byref(a_), C.byref(al_), CR._stream_ptr())
        torch.cuda.synchronize()
        oa, oal = O.make_added_layer(FT, N, S), OL.make_added_layer_lin(FT, P, N, S)
        CM.create_surface_layer_brdf_lin(r_o, d_o, oa, oal, P - 1, m, opol, oqp, tau_sum, tsd, F0, FT)
        d = _added_host(vsm, a)
        assert _rel(d["r_mp"][0], oa.r_mp[0]) < 1e-9 and _rel(d["t_pp"][0], oa.t_pp[0]) < 1e-14
        assert np.all(d["r_pm"] == 0) and np.all(d["t_mm"] == 0) and np.all(d["j0_p"] == 0)
        assert _rel(d["j0_m"], oa.j0_m) < 1e-8
        h = vsm.Architectures.to_host
        assert _rel(h(al.ap_r_mp).transpose(0, 1, 3, 2)[:, 0], oal.ap_r_mp[:, 0]) < 1e-9
        for t in (al.ap_r_pm, al.ap_t_pp, al.ap_t_mm, al.ap_J0_p):
            assert np.all(h(t) == 0)
        assert _rel(h(al.ap_J0_m), oal.ap_J0_m) < 1e-8


def test_ss_correction_coefficients_and_apply(vsm, arch):
    """vsm_coxmunk_ss_correction == apply_ss_correction! (coxmunk_surface.jl:481-545): the per-geometry coefficients and the
    in-place update of R_SFI."""
    FT = np.float64
    CR = vsm.CoreRT
    vza, vaz = [60, 45, 30, 15, 0, 15, 30, 45, 60], [180, 180, 180, 180, 0, 0, 0, 0, 0]
    for pol_name, m_max, kw in (("IQUV", 21, dict(wind_speed=5.0)), ("IQU", 9, dict(wind_speed=9.0, include_whitecaps=False)),
                                ("I", 0, dict(wind_speed=3.0, shadowing=False))):
        hs, os_ = _surfs(vsm, **kw)
        pol = vsm.host_model.polarization_type(pol_name)
        opol = O.polarization(pol_name)
        mu0 = O.cosd(30.0)
        S = 7
        tau = np.random.default_rng(2).uniform(0.05, 3.0, S)
        R0 = np.random.default_rng(3).standard_normal((S, pol.n, len(vza)))
        Rd = vsm.Architectures.array_type(arch)(R0.copy())
        coef = CR.apply_ss_correction_(Rd, hs, pol, vza, vaz, mu0, vsm.Architectures.array_type(arch)(tau), m_max, arch, FT)
        c_o = CM.ss_correction_coefficients(os_, opol, vza, vaz, mu0, m_max)
        assert _rel(vsm.Architectures.to_host(coef).T, c_o) < 1e-9
        Ro = R0.transpose(2, 1, 0).copy()
        CM.apply_ss_correction(Ro, os_, opol, vza, vaz, mu0, tau, m_max)
        assert _rel(vsm.Architectures.to_host(Rd).transpose(2, 1, 0), Ro) < 1e-12


# ---- BASELINE config C3: config/ocean_coxmunk.yaml ---------------------------------------------------------------------
def _c3_yaml_text():
    import yaml
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ocean_coxmunk_scene.json")) as f:
        d = json.load(f)
    d.pop("source")
    return yaml.safe_dump(d)


def _oracle_model(vsm, model, tau_abs=None):
    om = O.build_model("IQUV", 21, model.sza, list(model.vza), list(model.vaz), model.tau_rayl, tau_abs=tau_abs, depol=0.0,
                       m_max=model.m_max)
    om.greek_rayleigh = O.GreekCoefs(**{k: np.asarray(getattr(model.greek_rayleigh, k)) for k in
                                        ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")})
    return om


def test_c3_ocean_coxmunk_forward(vsm, arch):
    """parameters_from_yaml(config/ocean_coxmunk.yaml) -> model_from_parameters -> rt_run(model): IQUV, N = 60, 33 layers, 2
    spectral points, m = 0..21 (the Cox-Munk trait runs the Fourier loop to the stream cap), TMS correction on -- vs the oracle."""
    io = vsm.io_yaml
    params = io.parameters_from_yaml(_c3_yaml_text())
    model = io.model_from_parameters(params, arch)
    assert isinstance(model.surface, vsm.host_model.CoxMunkSurface) and model.surface.wind_speed == 5.0
    assert model.m_max == 21 and model.tau_rayl.shape == (2, 33) and model.quad_points.Nquad * 4 == 60
    trace = []
    scene = vsm.CoreRT.prepare_scene(model)
    scene.run(trace)
    torch.cuda.synchronize()
    R, T = scene.results_host()
    assert R.shape == (9, 4, 2)
    assert len(trace) == 22 * 33 and all(t["scatter"] and t["iface"] == "11" for t in trace)
    Ro, To = CM.rt_run(_oracle_model(vsm, model), CM.CoxMunkSurface(5.0))
    assert _rel(R, Ro) < 1e-8 and _rel(T, To) < 1e-8, (_rel(R, Ro), _rel(T, To))
    # without the TMS term: the Fourier-summed field, physical bounds
    scene.ss_correction = False
    scene.run()
    torch.cuda.synchronize()
    R0, T0 = scene.results_host()
    Ro0, _ = CM.rt_run(_oracle_model(vsm, model), CM.CoxMunkSurface(5.0), ss_correction=False)
    # (T is untouched by the TMS term; the traced run above walks the reference-layout layer loop, this one the native-layout run:
    # the same operations in another summation order)
    assert _rel(R0, Ro0) < 1e-8 and _rel(T0, T) < 1e-11
    assert np.all(R0[:, 0] > 0) and np.all(np.abs(R0[:, 1:]) <= R0[:, :1] + 1e-12)
    assert R0[6, 0, 0] > R0[2, 0, 0]      # vza 30 at vaz 0 looks into the glint of sza 30


def test_c3_ocean_coxmunk_linearized(vsm, arch):
    """rt_run(model, lin_model, NAer = 0, NGas = 1, NSurf = 1) on the ocean_coxmunk.yaml scene: R, T, dR/dx, dT/dx for x = (gas
    column scale, wind speed) vs the linearized oracle at 1e-8, and vs central differences of the device's own linearized
    forward at the reference's Jacobian gate (max 1e-3, mean 1e-4).  The gas slot is the scene's `q` slot
    (lin_model_from_parameters.jl:90) with a synthetic absorber so that its Jacobian is not identically zero."""
    io, H = vsm.io_yaml, vsm.host_model
    params = io.parameters_from_yaml(_c3_yaml_text())
    model = io.model_from_parameters(params, arch)
    S, L = model.tau_rayl.shape
    rng = np.random.default_rng(7)
    prof = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
    model.tau_abs = prof * 1.0
    g = prof * rng.uniform(0.8, 1.2, (S, L))
    lin = H.LinModel([g])
    R, T, Rd, Td = vsm.CoreRTLin.rt_run_lin(model, lin, 0, 1, 1)
    assert Rd.shape == (9, 4, 2, 2)
    om = _oracle_model(vsm, model, tau_abs=model.tau_abs)
    Ro, To, Rdo, Tdo = CM.rt_run_lin(om, OL.LinModel([g]), CM.CoxMunkSurface(5.0))
    for name, a, b in (("R", R, Ro), ("T", T, To), ("Rd", Rd, Rdo), ("Td", Td, Tdo)):
        assert _rel(a, b) < 1e-8, (name, _rel(a, b))
    for p in range(2):
        assert _rel(Rd[..., p], Rdo[..., p]) < 1e-8 and _rel(Td[..., p], Tdo[..., p]) < 1e-8
    # the zero-derivative q slot of the unmodified scene gives an exactly zero gas Jacobian
    m0 = copy.copy(model)
    m0.tau_abs = np.zeros((S, L))
    z = vsm.CoreRTLin.rt_run_lin(m0, H.LinModel([np.zeros((S, L))]), 0, 1, 1)
    assert np.all(z[2][..., 0] == 0) and np.all(z[3][..., 0] == 0) and np.max(np.abs(z[2][..., 1])) > 0
    # finite differences of the device's own forward
    def fwd(U=5.0, dabs=0.0):
        mm = copy.copy(model)
        mm.surface = H.CoxMunkSurface(wind_speed=U)
        mm.tau_abs = model.tau_abs + dabs * g
        return vsm.CoreRTLin.rt_run_lin(mm, lin, 0, 1, 1)[:2]
    e = 1e-4
    for p, (plus, minus) in enumerate(((fwd(dabs=e), fwd(dabs=-e)), (fwd(U=5.0 + e), fwd(U=5.0 - e)))):
        for an, a, b in ((Rd, plus[0], minus[0]), (Td, plus[1], minus[1])):
            fd = (a - b) / (2 * e)
            rel = np.abs(an[..., p] - fd) / np.max(np.abs(fd))
            assert rel.max() < 1e-3 and rel.mean() < 1e-4, (p, rel.max(), rel.mean())


def test_c3_linearized_sharded_equals_full(vsm, arch):
    """SceneLin with spec_slice: a rank's block of a linearized run equals the same rows of the full run bit for bit
    (ndoubl and the interface tags come from the full spectral axis)."""
    io, H = vsm.io_yaml, vsm.host_model
    params = io.parameters_from_yaml(_c3_yaml_text())
    params.spec_bands = [19417.0 + 0.5 * np.arange(5)]
    model = io.model_from_parameters(params, arch)
    S, L = model.tau_rayl.shape
    rng = np.random.default_rng(9)
    model.tau_abs = 10.0 ** rng.uniform(-4, -1, (S, 1)) * np.full((1, L), 1.0 / L)
    lin = H.LinModel([model.tau_abs * rng.uniform(0.5, 1.5, (S, L))])
    model.m_max = 3
    CL = vsm.CoreRTLin
    full = [t.clone() for t in CL.SceneLin(model, lin, 0, 1, 1).run()]
    for sl in (slice(0, 2), slice(2, 5), slice(5, 5)):
        part = CL.SceneLin(model, lin, 0, 1, 1, sl).run()
        torch.cuda.synchronize()
        assert torch.equal(part[0], full[0][sl]) and torch.equal(part[1], full[1][sl])
        assert torch.equal(part[2], full[2][:, sl]) and torch.equal(part[3], full[3][:, sl])
