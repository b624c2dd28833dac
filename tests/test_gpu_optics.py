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
