"""GPU parity of the per-scene layer optics built on the device (vsm_compute_Z_moments,
vsm_layer_optics, vsm_layer_dtau) against the oracle's restatement of compute_Z_matrices.jl:26-110 and
compEffectiveLayerProperties.jl:11-93, and of a whole Scene prepared on the device against one prepared by the host
mirror.  FP64 tolerances: Z 1e-13 of max|Z| (different summation order over l), layer scalars bit-exact (the kernel runs the
same FP64 operations, contraction off), rt_run 1e-12."""
import json
import os

import numpy as np
import pytest

from oracle import vsm_oracle as O

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


def _greeks(golden_dir):
    with open(os.path.join(golden_dir, "siewert2000_IIA.json")) as f:
        sw = json.load(f)["greek"]
    rng = np.random.default_rng(4)
    L = 24
    dec = 0.85 ** np.arange(L)
    rnd = {k: rng.standard_normal(L) * dec for k in ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")}
    return {"rayleigh": O.get_greek_rayleigh(0.0279), "hg": O.hg_greek(0.7, 35), "siewert": O.greek_from_dict(sw),
            "random24": O.greek_from_dict(rnd)}


@pytest.mark.parametrize("FT,tol", [(np.float64, 1e-13), (np.float32, 3e-7)])
@pytest.mark.parametrize("pol_name", ["I", "IQ", "IQU", "IQUV"])
def test_compute_Z_moments(vsm, arch, golden_dir, FT, tol, pol_name):
    """Z++(m), Z-+(m) for every polarization, m in {0, 1, 2, 5, 15} and four Greek sets (Rayleigh: 3 terms -> zero for m > 2;
    Henyey-Greenstein 36 terms; Siewert-2000 IIA 12 terms with all six families; a random 24-term set)."""
    import ctypes as C
    H, CR = vsm.host_model, vsm.CoreRT
    pol = H.polarization_type(pol_name)
    qp = H.rt_set_streams(21, 35.0, [0.0, 47.0], pol, FT)
    dq = CR.device_quad(qp, pol, arch, FT)
    N = qp.Nquad * pol.n
    conv = vsm.Architectures.array_type(arch)
    opol = O.polarization(pol_name)
    for name, g in _greeks(golden_dir).items():
        tab = np.stack([np.asarray(getattr(g, k), dtype=np.float64) for k in ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")])
        gd = conv(np.ascontiguousarray(tab))
        for m in (0, 1, 2, 5, 15):
            Zpp = torch.full((N, N), float("nan"), dtype=dq.dtype, device=dq.mu.device)
            Zmp = torch.full_like(Zpp, float("nan"))
            q = dq.cstruct()
            vsm._lib.call("vsm_compute_Z_moments", dq.dtype, C.byref(q), m, tab.shape[1], CR._ptr(gd), CR._ptr(Zpp), CR._ptr(Zmp),
                          CR._stream_ptr())
            Zo_pp, Zo_mp = O.compute_Z_moments(opol, qp.qp_mu.astype(np.float64), g, m)
            scale = max(np.max(np.abs(Zo_pp)), np.max(np.abs(Zo_mp)))
            dpp, dmp = vsm.Architectures.to_host(Zpp).T, vsm.Architectures.to_host(Zmp).T
            if scale == 0:
                assert np.all(dpp == 0) and np.all(dmp == 0), (name, m)
                continue
            assert np.max(np.abs(dpp - Zo_pp)) / scale < tol and np.max(np.abs(dmp - Zo_mp)) / scale < tol, (name, m)


def _aerosol_model(vsm, arch, FT, S=37, L=7, seed=3, variant="mixed"):
    H = vsm.host_model
    rng = np.random.default_rng(seed)
    tau_rayl = np.tile(rng.uniform(0.002, 0.02, L), (S, 1)) * rng.uniform(0.9, 1.1, (S, 1))
    tau_abs = 10.0 ** rng.uniform(-4, 0.5, (S, L))
    aer = [H.AerosolOptics(H.henyey_greenstein_greek(0.7, 9), 0.95, 0.1), H.AerosolOptics(H.henyey_greenstein_greek(0.4, 5), 0.8, 0.0)]
    tau_aer = np.zeros((2, L))
    tau_aer[0, 2:5] = [0.05, 0.1, 0.02]
    tau_aer[1, 4:] = 0.03
    if variant == "rayleigh_free":          # a layer where ONLY the aerosol scatters (mode 1) and one where nothing does
        tau_rayl[:, 3] = 0.0
        tau_rayl[:, 0] = 0.0
    if variant == "nonscattering_aerosol":  # mode 2
        aer[1] = H.AerosolOptics(H.henyey_greenstein_greek(0.4, 5), 0.0, 0.0)
    return H.model_from_arrays(arch, "IQU", 9, 40.0, [30.0, 0.0], [0.0, 90.0], tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer,
                               aerosol_optics=aer, depol=0.03, albedo=0.2, m_max=3, float_type=FT)


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("variant", ["mixed", "rayleigh_free", "nonscattering_aerosol"])
def test_layer_optics_device_equals_host_mirror(vsm, arch, FT, variant):
    """tau, varpi, tau_sum, dtau, component weights, ndoubl and interface tags of a two-aerosol scene: device optics pass ==
    host mirror (constructLayerOpticsComponents / extractEffectiveProps / get_dtau_ndoubl), bit for bit."""
    model = _aerosol_model(vsm, arch, FT, variant=variant)
    dev = vsm.CoreRT.Scene(model)
    host = vsm.CoreRT.Scene(model, host_optics=True)
    torch.cuda.synchronize()
    for name in ("tau", "varpi", "tau_sum", "dtau"):
        assert torch.equal(getattr(dev, name), getattr(host, name)), name
    assert dev.zcomp == host.zcomp
    for iz, (mixed, k) in enumerate(dev.zcomp):
        if mixed:
            assert torch.equal(dev.fcomp[iz], host.fcomp[iz]), iz
    for md, mh in zip(dev.moments, host.moments):
        assert [ly["nd"] for ly in md["layers"]] == [ly["nd"] for ly in mh["layers"]]
        assert [ly["iface"] for ly in md["layers"]] == [ly["iface"] for ly in mh["layers"]]
        assert md["iface_surface"] == mh["iface_surface"]
    if variant == "rayleigh_free":
        assert dev.moments[0]["layers"][0]["iface"] == "00" and dev.zcomp[3] == (False, 1)
    tol = 1e-12 if FT == np.float64 else 2e-5
    Zd, Zh = dev.Zc, host.Zc
    for m in range(model.m_max + 1):
        for a, b in zip(Zd[m], Zh[m]):
            assert _rel(vsm.Architectures.to_host(a), vsm.Architectures.to_host(b)) < (1e-13 if FT == np.float64 else 3e-7)
    Rd, Td = [x.clone() for x in dev.run()]
    Rh, Th = host.run()
    torch.cuda.synchronize()
    assert _rel(vsm.Architectures.to_host(Rd), vsm.Architectures.to_host(Rh)) < tol
    assert _rel(vsm.Architectures.to_host(Td), vsm.Architectures.to_host(Th)) < tol


def test_scene_reprepare_after_new_inputs(vsm, arch):
    """upload() + prepare() on an existing scene pick up changed optical depths (the bench's timed step re-runs both): the
    result equals a fresh scene's, and a shard's rows equal the full run's."""
    model = _aerosol_model(vsm, arch, np.float64, S=24)
    scene = vsm.CoreRT.Scene(model)
    R0 = scene.run()[0].clone()
    model.tau_abs = model.tau_abs * 1.7
    scene.upload()
    scene.prepare()
    R1 = scene.run()[0].clone()
    fresh = vsm.CoreRT.Scene(model)
    R2 = fresh.run()[0]
    torch.cuda.synchronize()
    assert torch.equal(R1, R2) and not torch.equal(R0, R1)
    part = vsm.CoreRT.Scene(model, slice(5, 17))
    Rp = part.run()[0]
    torch.cuda.synchronize()
    assert torch.equal(Rp, R2[5:17])
    empty = vsm.CoreRT.Scene(model, slice(24, 24))
    assert empty.run()[0].shape[0] == 0


# ---- the optics kernels against the ORACLE (not the package's own host mirror) ---------------------------------------------
def _oracle_twin(model, FT=np.float64):
    """The oracle's RTModel of a host_model.RTModel (same arrays)."""
    aer = [O.AerosolOptics(O.GreekCoefs(**vars(ao.greek_coefs)), ao.ssa, ao.f_trunc) for ao in model.aerosol_optics]
    om = O.build_model(model.polarization_type.name.replace("Stokes_", ""), 9, model.sza, model.vza, model.vaz, tau_rayl=model.tau_rayl,
                       tau_abs=model.tau_abs, tau_aer=model.tau_aer, aerosols=aer, depol=0.03, albedo=model.albedo, m_max=model.m_max,
                       FT=FT)
    assert np.array_equal(om.quad_points.qp_mu, model.quad_points.qp_mu)
    return om


@pytest.mark.parametrize("FT,tol", [(np.float64, 1e-14), (np.float32, 3e-7)])
@pytest.mark.parametrize("variant", ["mixed", "rayleigh_free", "nonscattering_aerosol"])
def test_layer_optics_vs_oracle(vsm, arch, FT, tol, variant):
    """vsm_layer_optics + vsm_compute_Z_moments + vsm_layer_dtau against the oracle's restatement of
    constructCoreOpticalProperties / extractEffectiveProps / get_dtau_ndoubl (compEffectiveLayerProperties.jl:11-93,
    types.jl:1262-1308, rt_kernel.jl:266-287): tau, varpi, tau_sum, dtau, ndoubl, interface tags and the per-point mixed
    phase matrices Z = sum_c fcomp[c] Z_c of a two-aerosol scene incl. the batch-global branches of the mixing `+`."""
    model = _aerosol_model(vsm, arch, FT, variant=variant)
    om = _oracle_twin(model, FT)
    dev = vsm.CoreRT.Scene(model)
    torch.cuda.synchronize()
    th = vsm.Architectures.to_host
    S, L = model.tau_rayl.shape
    for m in range(model.m_max + 1):
        lods = O.construct_core_optical_properties(om, m)
        tags, tau_sum = O.extract_effective_props(lods, FT)
        assert [ly["iface"] for ly in dev.moments[m]["layers"]] == tags and dev.moments[m]["iface_surface"] == tags[-1]
        for iz, lo in enumerate(lods):
            e = O.expand_optical_properties(lo, FT)
            ly = dev.moments[m]["layers"][iz]
            assert _rel(th(ly["props"].tau), e.tau) < tol and _rel(th(ly["props"].varpi), e.varpi) < tol, (m, iz)
            assert _rel(th(ly["tau_sum"]), tau_sum[:, iz]) < tol, (m, iz)
            scatter = float(np.max(e.tau * e.varpi)) > 2 * np.finfo(FT).eps
            if scatter:
                dtau_o, nd_o = O.get_dtau_ndoubl(e.tau, e.varpi, om.quad_points, FT, om.numerics)
                assert ly["nd"] == nd_o and _rel(th(ly["dtau"]), dtau_o) < tol, (m, iz)
            else:
                assert ly["nd"] == 0
            Zp, Zm = th(ly["props"].materialize().Zpp), th(ly["props"].materialize().Zmp)     # (1|S, j, i) layout
            scale = max(float(np.max(np.abs(e.Zpp))), float(np.max(np.abs(e.Zmp))), 1e-300)   # (Z(m) = 0 for Rayleigh, m > 2)
            ztol = 2e-13 if FT == np.float64 else 2e-6
            assert float(np.max(np.abs(np.broadcast_to(Zp.transpose(0, 2, 1), e.Zpp.shape) - e.Zpp))) / scale < ztol, (m, iz)
            assert float(np.max(np.abs(np.broadcast_to(Zm.transpose(0, 2, 1), e.Zmp.shape) - e.Zmp))) / scale < ztol, (m, iz)
        assert _rel(th(dev.moments[m]["tau_sum_surface"]), tau_sum[:, -1]) < tol


def _lin_twin(vsm, arch, n_aer, n_gas, FT=np.float64, S=19, L=5, seed=5, pol="IQU", l_trunc=9):
    """(oracle model, oracle lin, host model, host lin) of a scene with n_aer aerosols carrying derivative tables (random but
    fixed) and n_gas absorbers; every layer scatters (the oracle's quotient rule divides by tau varpi)."""
    from oracle import vsm_oracle_lin as OL
    H = vsm.host_model
    rng = np.random.default_rng(seed)
    tau_rayl = np.tile(rng.uniform(0.002, 0.02, L), (S, 1)) * rng.uniform(0.9, 1.1, (S, 1))
    tau_abs = 10.0 ** rng.uniform(-4, 0.3, (S, L))
    lmax = [9, 6, 4][:n_aer]
    aer_o, aer_h, lin_o, lin_h = [], [], [], []
    for ia in range(n_aer):
        g0 = O.hg_greek(0.7 - 0.2 * ia, lmax[ia])
        for name in ("alpha", "gamma", "delta", "epsilon", "zeta"):        # a fully polarized coefficient set
            setattr(g0, name, 0.3 * rng.standard_normal(lmax[ia] + 1) * 0.8 ** np.arange(lmax[ia] + 1))
        gd = [O.GreekCoefs(*(0.1 * rng.standard_normal(lmax[ia] + 1) for _ in range(6))) for _ in range(4)]
        ssa, ft = 0.95 - 0.1 * ia, 0.1 * (1 - ia)
        sd, fd = rng.uniform(-0.3, 0.3, 4), rng.uniform(-0.2, 0.2, 4)
        aer_o.append(O.AerosolOptics(g0, ssa, ft))
        aer_h.append(H.AerosolOptics(H.GreekCoefs(**vars(g0)), ssa, ft))
        lin_o.append(OL.LinAerosolOptics(gd, sd, fd))
        lin_h.append(H.LinAerosolOptics([H.GreekCoefs(**vars(g)) for g in gd], sd, fd))
    tau_aer = rng.uniform(0.0, 0.1, (n_aer, L))
    if n_aer:
        tau_aer[0, 0] = 0.0                                                  # an aerosol absent from the top layer
    tau_aer_dot = rng.uniform(-1.0, 1.0, (n_aer, 7, L)) if n_aer else None
    tad = [10.0 ** rng.uniform(-4, 0.3, (S, L)) for _ in range(n_gas)]
    om = O.build_model(pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 120.0], tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer,
                       aerosols=aer_o, depol=0.03, albedo=0.2, m_max=3, FT=FT)
    ol = OL.LinModel([t.copy() for t in tad], tau_aer_dot=tau_aer_dot, lin_aerosol_optics=lin_o if n_aer else None)
    pm = H.model_from_arrays(arch, pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 120.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                             tau_aer=tau_aer, aerosol_optics=aer_h, depol=0.03, albedo=0.2, m_max=3, float_type=FT)
    pl = H.LinModel([t.copy() for t in tad], tau_aer_dot=tau_aer_dot, lin_aerosol_optics=lin_h if n_aer else None)
    return om, ol, pm, pl


@pytest.mark.parametrize("FT,tol", [(np.float64, 1e-12), (np.float32, 5e-6)])
@pytest.mark.parametrize("n_aer,n_gas", [(0, 2), (1, 1), (2, 1), (1, 0), (3, 2)])
def test_layer_optics_lin_vs_oracle(vsm, arch, FT, tol, n_aer, n_gas):
    """vsm_layer_optics_lin (+ vsm_compute_Z_moments on the Greek-coefficient derivatives) against the oracle's restatement of
    the reference's pairwise `+` chain with derivatives (oracle/vsm_oracle_lin.layer_optics_lin: types_lin.jl:196-380,
    compEffectiveLayerProperties_lin.jl:43-197,330-395): tau_dot / 2^ndoubl, varpi_dot, tau_sum_dot and Z_dot rebuilt from the
    device's coefficients over the component matrices -- gases only, one aerosol (7 slots), two and three aerosols, no gas.
    1e-12 of the array maximum in FP64.  (Two or more aerosols with derivatives: the reference's `+` cannot run them,
    types_lin.jl:268-277 -- the oracle chains its quotient rule, the device evaluates the closed form: DESIGN 6.)"""
    from oracle import vsm_oracle_lin as OL
    om, ol, pm, pl = _lin_twin(vsm, arch, n_aer, n_gas, FT)
    sc = vsm.CoreRTLin.SceneLin(pm, pl, n_aer, n_gas, 1)
    torch.cuda.synchronize()
    th = vsm.Architectures.to_host
    S, L = pm.tau_rayl.shape
    npl = 7 * n_aer + n_gas
    assert sc.pl == npl and sc.P == npl + 1
    for m in range(pm.m_max + 1):
        lods = O.construct_core_optical_properties(om, m)
        lins = OL.layer_optics_lin(om, ol, lods, m)
        tsd = np.zeros((S, npl))
        for iz in range(L):
            nd = sc.fwd.moments[m]["layers"][iz]["nd"]
            dd = th(sc.dtau_dot_all[iz])                      # (P, S)
            assert np.all(dd[npl:] == 0)
            if npl:
                assert _rel(dd[:npl].T, lins[iz].tau_dot / 2.0 ** nd) < tol, (m, iz)
                assert _rel(th(sc.varpi_dot[iz]).T, lins[iz].varpi_dot) < tol, (m, iz)
                if iz:
                    assert _rel(th(sc.tau_sum_dot[iz]).T, tsd) < tol, (m, iz)
                else:
                    assert np.all(th(sc.tau_sum_dot[0]) == 0)
            tsd = tsd + lins[iz].tau_dot
            if n_aer:
                Zp, Zm = th(sc.Zall[m][0]).transpose(0, 2, 1), th(sc.Zall[m][1]).transpose(0, 2, 1)      # [CT, i, j]
                cf = th(sc.zdcoef[iz])                                                                   # (S, pl, CT)
                Zpd, Zmd = np.einsum("spc,cij->psij", cf, Zp), np.einsum("spc,cij->psij", cf, Zm)
                scale = max(float(np.max(np.abs(lins[iz].Zpp_dot))), float(np.max(np.abs(lins[iz].Zmp_dot))))
                assert float(np.max(np.abs(Zpd - lins[iz].Zpp_dot))) <= 10 * tol * scale, (m, iz)
                assert float(np.max(np.abs(Zmd - lins[iz].Zmp_dot))) <= 10 * tol * scale, (m, iz)
                # the forward mix the linearized elemental step uses: Z = sum_c fz[c] Z_c
                e = O.expand_optical_properties(lods[iz], FT)
                fz = th(sc.fz[iz])                                                                       # (S, C)
                Zf = np.einsum("sc,cij->sij", fz, Zp[:1 + n_aer])
                assert float(np.max(np.abs(Zf - e.Zpp))) <= 10 * tol * float(np.max(np.abs(e.Zpp))), (m, iz)
        if npl:
            assert _rel(th(sc.tau_sum_dot[L]).T, tsd) < tol


@pytest.mark.parametrize("n_aer,n_gas,pol,l_trunc", [(1, 1, "IQU", 9), (2, 1, "I", 9), (1, 2, "IQU", 33)])
def test_scene_lin_device_optics_vs_oracle_and_host_mirror(vsm, arch, n_aer, n_gas, pol, l_trunc):
    """rt_run(model, lin_model, NAer, NGas, 1) with the linearized layer optics built on the device (coefficient form of Z_dot,
    vsm_elemental_lin_mix) against the oracle's rt_run_lin (1e-8) and against the same run fed by the host mirror with the
    materialised Z_dot (1e-11); N = 21 / 7 / 57 (the last one on the fused linearized strip kernels)."""
    from oracle import vsm_oracle_lin as OL
    om, ol, pm, pl = _lin_twin(vsm, arch, n_aer, n_gas, S=5, L=3, pol=pol, l_trunc=l_trunc)
    Ro, To, Rdo, Tdo = OL.rt_run_lin(om, ol)
    dev = vsm.CoreRTLin.SceneLin(pm, pl, n_aer, n_gas, 1)
    dev.run()
    R, T, Rd, Td = dev.results_host()
    host = vsm.CoreRTLin.SceneLin(pm, pl, n_aer, n_gas, 1, host_optics=True)
    host.run()
    Rh, Th, Rdh, Tdh = host.results_host()
    assert _rel(R, Ro) < 1e-9 and _rel(T, To) < 1e-9
    assert _rel(R, Rh) < 1e-12 and _rel(T, Th) < 1e-12
    for p in range(Rd.shape[-1]):
        assert _rel(Rd[..., p], Rdo[..., p]) < 1e-8 and _rel(Td[..., p], Tdo[..., p]) < 1e-8, p
        assert _rel(Rd[..., p], Rdh[..., p]) < 1e-11 and _rel(Td[..., p], Tdh[..., p]) < 1e-11, p


@pytest.mark.parametrize("host_optics", [False, True])
def test_scene_lin_shard_with_aerosol_equals_full_run(vsm, arch, host_optics):
    """A rank's spectral block of a linearized run WITH aerosol Jacobians equals the same rows of the full run bit for bit
    (Z_dot varies along the spectral axis through the mixing weights: the block must read its own points' coefficients / its
    own slice of a host-materialised Z_dot), for unequal blocks and an empty one."""
    om, ol, pm, pl = _lin_twin(vsm, arch, 1, 1, S=11, L=3)
    full = [t.clone() for t in vsm.CoreRTLin.SceneLin(pm, pl, 1, 1, 1, host_optics=host_optics).run()]
    for sl in (slice(0, 4), slice(4, 11), slice(11, 11)):
        part = vsm.CoreRTLin.SceneLin(pm, pl, 1, 1, 1, sl, host_optics=host_optics).run()
        torch.cuda.synchronize()
        for f, p_, sdim in zip(full, part, (0, 0, 1, 1)):
            ref = f[sl] if sdim == 0 else f[:, sl]
            assert torch.equal(ref, p_), sl
