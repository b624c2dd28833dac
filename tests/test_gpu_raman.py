"""GPU parity tests of the rotational-Raman (RRS) pass: every vsm_*_inelastic_rrs_* entry point is called through
the C ABI and compared with oracle/vsm_oracle_raman.py on the same seeded inputs, then rt_run(RS_type, model)
end to end, then a size-independent property at a larger size.

Tolerances: FP64 operators 1e-10 relative to the array maximum (summation order only), FP64 end-to-end 1e-8,
FP32 operators 5e-4 / end-to-end 1e-2 (the reference's own FP32 gate, test/test_float32.jl:58-64).
"""
import sys

import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_raman as OR

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def vsm():
    import vsmartmom_jl_amd as v
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    v._lib.lib()
    return v


@pytest.fixture(scope="module")
def arch(vsm):
    return vsm.Architectures.GPU()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


TOL = {np.float64: 1e-10, np.float32: 5e-4}
SHIFTS = np.array([-5, -2, -1, 1, 3, 40, 0])      # 40: never in band for S <= 40
W_IE = np.array([0.004, 0.012, 0.02, 0.018, 0.007, 0.3, 0.009])


def _m4(vsm, arch, A, FT):   # [K,S,i,j] math order -> device [N,N,S,K] column-major
    return vsm.Architectures.array_type(arch)(np.ascontiguousarray(np.asarray(A, dtype=FT).transpose(0, 1, 3, 2)))


def _h4(vsm, t):
    return vsm.Architectures.to_host(t).transpose(0, 1, 3, 2)


def _setup(vsm, arch, FT, pol_name, S=14, l_trunc=7, seed=0):
    """Quadrature, a random-but-physical elemental layer and the RRS inputs on both sides."""
    rng = np.random.default_rng(seed)
    pol = O.polarization(pol_name)
    qp = O.rt_set_streams_gausslegquad(l_trunc, 35.0, [20.0, 50.0], pol, FT)
    Hm = vsm.host_model
    hpol = Hm.polarization_type(pol_name)
    hqp = Hm.rt_set_streams(l_trunc, 35.0, [20.0, 50.0], hpol, FT)
    N = qp.Nquad * pol.n
    greek = O.get_greek_rayleigh(0.03)
    graman = O.get_greek_rayleigh(0.1)
    tau = (0.02 + 0.2 * rng.random(S)) * (1 + 5 * (rng.random(S) < 0.3))
    varpi = 0.3 + 0.6 * rng.random(S)
    fscatt = 0.5 + 0.5 * rng.random(S)
    tau_sum = 0.1 * rng.random(S)
    F0 = np.zeros((pol.n, S))
    F0[0] = 0.5 + rng.random(S)
    return dict(rng=rng, pol=pol, qp=qp, hpol=hpol, hqp=hqp, N=N, S=S, greek=greek, graman=graman, tau=tau.astype(FT),
                varpi=varpi.astype(FT), fscatt=fscatt.astype(FT), tau_sum=tau_sum.astype(FT), F0=F0.astype(FT))


def _oracle_added(c, FT, m, ndoubl_mode):
    """Oracle elemental (elastic + inelastic) for the setup `c`; returns everything needed downstream."""
    pol, qp, S, N = c["pol"], c["qp"], c["S"], c["N"]
    dtau, nd = O.get_dtau_ndoubl(c["tau"], c["varpi"], qp, FT)
    if ndoubl_mode == 0:
        dtau, nd = c["tau"].astype(FT), 0
    elif ndoubl_mode > 0:
        nd = ndoubl_mode
        dtau = (c["tau"] / FT(2 ** nd)).astype(FT)
    Zpp, Zmp = O.compute_Z_moments(pol, qp.qp_mu, c["greek"], m)
    rs = OR.RRS(i_shift=SHIFTS, varpi_ie=W_IE, greek_raman=c["graman"], fscatt_rayl=c["fscatt"])
    rs.Zpp_ie, rs.Zmp_ie = O.compute_Z_moments(pol, qp.qp_mu, c["graman"], m)
    add = O.make_added_layer(FT, N, S)
    ars = OR.make_added_layer_rs(FT, len(SHIFTS), N, S)
    OR.elemental_inelastic(rs, pol, c["tau_sum"], dtau, c["F0"], m, nd, qp, ars, FT)
    O.elemental(pol, c["tau_sum"], dtau, c["F0"], c["varpi"], Zpp, Zmp, m, nd, qp, add, FT)
    return rs, add, ars, dtau, nd, (Zpp, Zmp)


def _device_side(vsm, arch, c, FT, rs, m):
    CR, RR = vsm.CoreRT, vsm.CoreRTRaman
    dq = CR.device_quad(c["hqp"], c["hpol"], arch, FT)
    hrs = RR.RRS(SHIFTS, W_IE, None)
    drs = RR.device_rrs(hrs, arch, FT)
    conv = vsm.Architectures.array_type(arch)
    drs.fscatt = conv(c["fscatt"])
    drs.Zpp, drs.Zmp = CR.to_device_matrix(rs.Zpp_ie, arch, FT), CR.to_device_matrix(rs.Zmp_ie, arch, FT)
    return dq, drs, conv


def _upload_added(vsm, arch, add, ars, FT):
    CR, RR = vsm.CoreRT, vsm.CoreRTRaman
    S, N = add.r_mp.shape[0], add.r_mp.shape[1]
    K = ars.ier_mp.shape[0]
    conv = vsm.Architectures.array_type(arch)
    pa = CR.make_added_layer(FT, arch, (N, N), S)
    for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
        getattr(pa, k).copy_(CR.to_device_matrix(getattr(add, k), arch, FT))
    pa.j0_p.copy_(conv(add.j0_p))
    pa.j0_m.copy_(conv(add.j0_m))
    pr = RR.AddedLayerRS(FT, arch, K, N, S)
    for k in ("ier_mp", "iet_pp", "ier_pm", "iet_mm"):
        getattr(pr, k).copy_(_m4(vsm, arch, getattr(ars, k), FT))
    pr.ieJ0_p.copy_(conv(ars.ieJ0_p))
    pr.ieJ0_m.copy_(conv(ars.ieJ0_m))
    return pa, pr


def _check_rs(vsm, dev, ora, tol, names_m, names_v):
    for k in names_m:
        got, ref = _h4(vsm, getattr(dev, k)), getattr(ora, k)
        assert _rel(got, ref) <= tol, (k, _rel(got, ref))
    for k in names_v:
        got, ref = vsm.Architectures.to_host(getattr(dev, k)), getattr(ora, k)
        assert _rel(got, ref) <= tol, (k, _rel(got, ref))


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("pol", ["I", "IQU", "IQUV"])
@pytest.mark.parametrize("m,ndmode", [(0, 0), (0, -1), (1, -1), (2, 3)])
def test_elemental_inelastic(vsm, arch, FT, pol, m, ndmode):
    c = _setup(vsm, arch, FT, pol)
    rs, add, ars, dtau, nd, _ = _oracle_added(c, FT, m, ndmode)
    dq, drs, conv = _device_side(vsm, arch, c, FT, rs, m)
    pr = vsm.CoreRTRaman.AddedLayerRS(FT, arch, len(SHIFTS), c["N"], c["S"])
    vsm.CoreRTRaman.elemental_inelastic_(drs, conv(c["tau_sum"]), conv(dtau), conv(np.ascontiguousarray(c["F0"].T)), m, nd, dq, pr)
    mats = ("ier_mp", "iet_pp") + (("ier_pm", "iet_mm") if (nd < 1 or c["pol"].n == 1) else ())
    _check_rs(vsm, pr, ars, 1e-12 if FT == np.float64 else 2e-5, mats, ("ieJ0_p", "ieJ0_m"))
    # out-of-band couplings are exactly zero
    assert float(pr.ier_mp[5].abs().max()) == 0.0 and float(pr.ieJ0_p[5].abs().max()) == 0.0


@pytest.mark.parametrize("FT", [np.float64, np.float32])
# FP64 N = 4, 7, 11, 16, 19, 21, 24, 27, 30: one wave per line (vsm_raman_wave.hip, k-step counts 1..8; pipelined body up to 24);
# FP32 N <= 30: one workgroup per point; 63: operator chain
@pytest.mark.parametrize("pol,l_trunc", [("I", 1), ("I", 7), ("I", 15), ("I", 25), ("I", 31), ("IQU", 7), ("IQU", 9),
                                         ("IQU", 11), ("IQU", 13), ("IQU", 35), ("IQU", 55)])   # 55: N = 93 (FP32: NP = 96 elastic kernels)
def test_doubling_inelastic(vsm, arch, FT, pol, l_trunc):
    S = 14 if l_trunc < 20 else 9
    c = _setup(vsm, arch, FT, pol, S=S, l_trunc=l_trunc, seed=3)
    nd = 5
    rs, add, ars, dtau, nd, _ = _oracle_added(c, FT, 0, nd)
    pa, pr = _upload_added(vsm, arch, add, ars, FT)
    dq, drs, conv = _device_side(vsm, arch, c, FT, rs, 0)
    expk = np.exp(-dtau / FT(c["qp"].mu0)).astype(FT)
    OR.doubling_inelastic(rs, c["pol"], expk, nd, add, ars, FT)
    vsm.CoreRTRaman.doubling_inelastic_(drs, c["hpol"], conv(expk), nd, pa, pr)
    tol = TOL[FT]
    _check_rs(vsm, pr, ars, tol, ("ier_mp", "iet_pp", "ier_pm", "iet_mm"), ("ieJ0_p", "ieJ0_m"))
    f = vsm.CoreRT.from_device_matrix
    for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
        assert _rel(f(getattr(pa, k)), getattr(add, k)) <= tol, k
    for k in ("j0_p", "j0_m"):
        assert _rel(vsm.Architectures.to_host(getattr(pa, k)), getattr(add, k)) <= tol, k


def _random_composite(c, FT, rng):
    S, N, K = c["S"], c["N"], len(SHIFTS)
    comp = O.make_composite_layer(FT, N, S)
    crs = OR.make_composite_layer_rs(FT, K, N, S)
    for k in ("R_mp", "R_pm"):
        getattr(comp, k)[...] = 0.3 * rng.random((S, N, N)) / N
    for k in ("T_pp", "T_mm"):
        getattr(comp, k)[...] = 0.3 * rng.random((S, N, N)) / N + 0.6 * np.eye(N)[None]
    comp.J0_p[...] = rng.random((S, N))
    comp.J0_m[...] = rng.random((S, N))
    for dn, sh in enumerate(SHIFTS):
        n0, n1 = OR.get_n0_n1(S, int(sh))
        L = n1.stop - n1.start
        if L <= 0:
            continue
        for k in ("ieR_mp", "ieR_pm", "ieT_pp", "ieT_mm"):
            getattr(crs, k)[dn, n1] = 0.05 * rng.standard_normal((L, N, N)) / N
        crs.ieJ0_p[dn, n1] = 0.05 * rng.standard_normal((L, N))
        crs.ieJ0_m[dn, n1] = 0.05 * rng.standard_normal((L, N))
    return comp, crs


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("pol,l_trunc,surface", [("I", 1, False), ("I", 7, False), ("I", 15, False), ("I", 25, False),
                                                 ("I", 31, True), ("IQU", 7, False), ("IQU", 7, True), ("IQU", 9, False),
                                                 ("IQU", 11, False), ("IQU", 13, True), ("IQU", 35, False)])
@pytest.mark.parametrize("iface", ["11", "00", "01", "10"])
def test_interaction_inelastic(vsm, arch, FT, pol, l_trunc, surface, iface):
    """interaction!(RS_type::RRS, scattering_interface, ...) for the four interface tags vs the oracle (random composite)."""
    if iface != "11" and (l_trunc in (1, 15, 25, 11) or FT == np.float32 and l_trunc > 9):
        pytest.skip("the plain interfaces are operator chains: a subset of the shapes covers them")
    CR, RR = vsm.CoreRT, vsm.CoreRTRaman
    S = 14 if l_trunc < 20 else 9
    c = _setup(vsm, arch, FT, pol, S=S, l_trunc=l_trunc, seed=5)
    rs, add, ars, dtau, nd, _ = _oracle_added(c, FT, 0, 4)
    expk = np.exp(-dtau / FT(c["qp"].mu0)).astype(FT)
    OR.doubling_inelastic(rs, c["pol"], expk, nd, add, ars, FT)
    if surface:
        O.create_surface_layer_lambertian(0.3, add, 0, c["pol"], c["qp"], c["tau_sum"], FT)
        for k in ("ier_mp", "iet_pp", "ier_pm", "iet_mm", "ieJ0_p", "ieJ0_m"):
            getattr(ars, k)[...] = 0
    comp, crs = _random_composite(c, FT, c["rng"])
    conv = vsm.Architectures.array_type(arch)
    # device copies
    pa, pr = _upload_added(vsm, arch, add, ars, FT)
    if surface:
        ps = CR.make_added_layer(FT, arch, (c["N"], c["N"]), S, shared=True)
        for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
            getattr(ps, k).copy_(CR.to_device_matrix(getattr(add, k)[:1], arch, FT))
        ps.j0_p.copy_(conv(add.j0_p))
        ps.j0_m.copy_(conv(add.j0_m))
        pa = ps
    pc = CR.make_composite_layer(FT, arch, (c["N"], c["N"]), S)
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        getattr(pc, k).copy_(CR.to_device_matrix(getattr(comp, k), arch, FT))
    pc.J0_p.copy_(conv(comp.J0_p))
    pc.J0_m.copy_(conv(comp.J0_m))
    pcr = RR.CompositeLayerRS(FT, arch, len(SHIFTS), c["N"], S)
    for k in ("ieR_mp", "ieR_pm", "ieT_pp", "ieT_mm"):
        getattr(pcr, k).copy_(_m4(vsm, arch, getattr(crs, k), FT))
    pcr.ieJ0_p.copy_(conv(crs.ieJ0_p))
    pcr.ieJ0_m.copy_(conv(crs.ieJ0_m))
    dq, drs, _ = _device_side(vsm, arch, c, FT, rs, 0)
    OR.interaction_inelastic(iface, rs, comp, crs, add, ars, FT)
    RR.interaction_inelastic_(drs, iface, pc, pcr, pa, pr)
    tol = TOL[FT]
    _check_rs(vsm, pcr, crs, tol, ("ieR_mp", "ieR_pm", "ieT_pp", "ieT_mm"), ("ieJ0_p", "ieJ0_m"))
    f = CR.from_device_matrix
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        assert _rel(f(getattr(pc, k)), getattr(comp, k)) <= tol, k
    for k in ("J0_p", "J0_m"):
        assert _rel(vsm.Architectures.to_host(getattr(pc, k)), getattr(comp, k)) <= tol, k


@pytest.mark.parametrize("l_trunc,N", [(3, 5), (5, 6), (9, 8), (11, 9), (13, 10), (19, 13), (21, 14), (27, 17), (33, 20), (37, 22)])
def test_raman_quad_kernels_every_size(vsm, arch, l_trunc, N):
    """vsm_raman_quad.hip (four Raman lines per wave on the 4 x 4 x 4 MFMA) owns FP64 3 <= N <= 22 -- the doubling of N = 20, 21, 22
    runs all its steps in vsm_raman_chain.hip (two lines per wave, the state on chip); the lists above hit N = 4, 7, 11, 12,
    15, 16, 18, 19, 21.  The remaining sizes -- every N mod 4, the rider columns inside / outside the last block of real columns, odd and
    even N^2 (the late slot of the LDS-DMA images) -- through five doubling steps incl. the last one (apply_D on the way out) and
    the _11 interaction, partial last quads (K = 7 lines) included."""
    c = _setup(vsm, arch, np.float64, "I", S=9, l_trunc=l_trunc, seed=3)
    assert c["N"] == N
    test_doubling_inelastic(vsm, arch, np.float64, "I", l_trunc)
    test_interaction_inelastic(vsm, arch, np.float64, "I", l_trunc, False, "11")


@pytest.mark.parametrize("K", [9, 70])
def test_raman_quad_kernels_line_lists(vsm, arch, monkeypatch, K):
    """The line lists of the quad / chain kernels (ranks of the in-band lines of a recipient, four / two per wave) on irregular offsets: K = 9 leaves
    a last quad with one line, K = 70 crosses the 64-line ballot; offsets of both signs, one that is never in band, a zero offset;
    recipients near the band edges get partial lists.  Doubling steps and the _11 interaction against the oracle (N = 21)."""
    rng = np.random.default_rng(K)
    if K == 9:
        shifts = np.array([-9, -5, -2, 0, 2, 3, 4, 6, 40])
    else:   # 55 offsets that are never in band (S = 14), then 15 in-band ones at line indices 55 .. 69
        shifts = np.concatenate([np.arange(-100, -45), np.arange(-6, 9)])
    assert len(shifts) == K and len(np.unique(shifts)) == K
    monkeypatch.setattr(sys.modules[__name__], "SHIFTS", shifts)
    monkeypatch.setattr(sys.modules[__name__], "W_IE", 0.3 * rng.random(K) / K)
    test_doubling_inelastic(vsm, arch, np.float64, "IQU", 13)
    test_interaction_inelastic(vsm, arch, np.float64, "IQU", 13, False, "11")


def _raman_models(vsm, arch, pol, l_trunc, S, L, FT, uniform=False, seed=11, m_max=2):
    rng = np.random.default_rng(seed)
    tau_rayl = np.tile(np.linspace(0.02, 0.05, L), (S, 1))
    if uniform:
        tau_abs = np.tile(np.linspace(0.03, 0.01, L), (S, 1))
    else:
        tau_abs = (10.0 ** rng.uniform(-3, 0.3, (S, 1))) * np.linspace(0.5, 1.5, L)[None, :] / L
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.12, m_max=m_max)
    om = O.build_model(pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 60.0], FT=FT, **kw)
    om.varpi_cabannes = 0.96
    pm = vsm.host_model.model_from_arrays(arch, pol, l_trunc, 35.0, [20.0, 50.0], [0.0, 60.0], float_type=FT, **kw)
    pm.varpi_Cabannes = 0.96
    F0 = np.zeros((om.pol.n, S))
    F0[0] = 0.8 + 0.4 * rng.random(S)
    om.F0, pm.F0 = F0, F0
    return om, pm


@pytest.mark.parametrize("FT,pol,l_trunc,S", [(np.float64, "I", 7, 24), (np.float64, "IQU", 7, 24), (np.float64, "IQU", 35, 8),
                                              (np.float32, "IQU", 7, 24)])
def test_rt_run_rrs_end_to_end(vsm, arch, FT, pol, l_trunc, S):
    """rt_run(RS_type::RRS, model, iBand) vs the oracle: spectrally varying absorption and source, 3 layers, Lambertian."""
    om, pm = _raman_models(vsm, arch, pol, l_trunc, S, 3, FT)
    graman = O.get_greek_rayleigh(6.0 / 7.0 * 0.5)
    ors = OR.RRS(i_shift=SHIFTS, varpi_ie=W_IE, greek_raman=graman)
    Hm = vsm.host_model
    prs = vsm.CoreRTRaman.RRS(SHIFTS, W_IE, Hm.GreekCoefs(**vars(graman)))
    tro, trg = [], []
    ref = OR.rt_run_rrs(om, ors, trace=tro)
    got = vsm.CoreRTRaman.rt_run(prs, pm, 1, trace=trg)
    assert [t["ndoubl"] for t in tro] == [t["ndoubl"] for t in trg]
    tol = 1e-8 if FT == np.float64 else 1e-2
    for name, g, r in zip(("R", "T", "ieR", "ieT"), got, ref):
        assert _rel(g, r) <= tol, (name, _rel(g, r))
    assert np.max(np.abs(ref[2])) > 1e-5


@pytest.mark.parametrize("pattern", ["01_10_11", "00_01_11"])
def test_rt_run_rrs_nonscattering_layers(vsm, arch, pattern):
    """rt_run(RS_type::RRS) on columns whose layers leave the _11 interface: a (numerically) non-scattering layer at the top, in
    the middle, or two of them at the top make interaction!(::RRS) dispatch on _01 / _10 / _00 (interaction_inelastic.jl:74-262;
    rt_kernel!(::RRS) itself always runs elemental / doubling, rt_kernel.jl:365) -- vs the oracle, with the tags checked."""
    FT, S, L = np.float64, 16, 4
    rng = np.random.default_rng(21)
    tiny = 1e-22
    tau_rayl = np.tile({"01_10_11": [tiny, 0.04, tiny, 0.03], "00_01_11": [tiny, tiny, 0.04, 0.03]}[pattern], (S, 1))
    tau_abs = (10.0 ** rng.uniform(-3, 0.0, (S, 1))) * np.linspace(0.5, 1.5, L)[None, :] / L
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.12, m_max=2)
    om = O.build_model("IQU", 7, 35.0, [20.0, 50.0], [0.0, 60.0], FT=FT, **kw)
    om.varpi_cabannes = 0.96
    pm = vsm.host_model.model_from_arrays(arch, "IQU", 7, 35.0, [20.0, 50.0], [0.0, 60.0], float_type=FT, **kw)
    pm.varpi_Cabannes = 0.96
    tags, _ = O.extract_effective_props(O.construct_core_optical_properties(om, 0), FT)
    assert list(tags[1:L]) == pattern.split("_")
    graman = O.get_greek_rayleigh(6.0 / 7.0 * 0.5)
    ors = OR.RRS(i_shift=SHIFTS, varpi_ie=W_IE, greek_raman=graman)
    prs = vsm.CoreRTRaman.RRS(SHIFTS, W_IE, vsm.host_model.GreekCoefs(**vars(graman)))
    ref = OR.rt_run_rrs(om, ors)
    got = vsm.CoreRTRaman.rt_run(prs, pm, 1)
    for name, g, r in zip(("R", "T", "ieR", "ieT"), got, ref):
        assert _rel(g, r) <= 1e-8, (name, _rel(g, r))
    assert np.max(np.abs(ref[2])) > 1e-6


@pytest.mark.parametrize("K", [100, 130])
def test_rt_run_rrs_many_lines(vsm, arch, K):
    """Line lists longer than one 64-bit ballot: K = 100 (both halves of the 128-bit line mask of the wave-per-line
    kernels, most lines out of band at the edges) and K = 130 (past the mask: the workgroup-per-point kernels)."""
    FT, S = np.float64, 40
    om, pm = _raman_models(vsm, arch, "I", 7, S, 2, FT, m_max=1)
    shifts = np.array([d for d in range(-(K // 2), K - K // 2 + 1) if d != 0])[:K]
    rng = np.random.default_rng(4)
    w_ie = (0.04 / K) * (0.5 + rng.random(K))
    graman = O.get_greek_rayleigh(6.0 / 7.0 * 0.5)
    ors = OR.RRS(i_shift=shifts, varpi_ie=w_ie, greek_raman=graman)
    prs = vsm.CoreRTRaman.RRS(shifts, w_ie, vsm.host_model.GreekCoefs(**vars(graman)))
    ref = OR.rt_run_rrs(om, ors)
    got = vsm.CoreRTRaman.rt_run(prs, pm, 1)
    for name, g, r in zip(("R", "T", "ieR", "ieT"), got, ref):
        assert _rel(g, r) <= 1e-8, (name, _rel(g, r))
    assert np.max(np.abs(ref[2])) > 1e-6


def test_rt_run_rrs_perturbation_property_large(vsm, arch):
    """Size-independent property at S = 96, K = 7, N = 24: on a spectrally uniform atmosphere with the Raman phase
    matrix equal to the elastic one, ieR(n1) = dR/d(varpi_Cabannes) * sum of the in-band varpi_ie (first-order
    perturbation identity); the derivative comes from central differences of the DEVICE elastic rt_run."""
    FT, S, L = np.float64, 96, 3
    om, pm = _raman_models(vsm, arch, "IQU", 7, S, L, FT, uniform=True)
    pm.F0 = None
    Hm = vsm.host_model
    fsc = pm.tau_rayl / (pm.tau_rayl + pm.tau_abs)
    prs = vsm.CoreRTRaman.RRS(SHIFTS, W_IE, pm.greek_rayleigh, fscattRayl=fsc)
    R, T, ieR, ieT = vsm.CoreRTRaman.rt_run(prs, pm, 1)
    h = 1e-5
    pm.varpi_Cabannes = 0.96 + h
    Rp, Tp = vsm.CoreRT.rt_run(pm)
    pm.varpi_Cabannes = 0.96 - h
    Rm, Tm = vsm.CoreRT.rt_run(pm)
    dR, dT = (Rp - Rm) / (2 * h), (Tp - Tm) / (2 * h)
    wsum = np.array([sum(w for s, w in zip(SHIFTS, W_IE) if 0 <= n1 + s < S) for n1 in range(S)])
    assert _rel(ieR, dR * wsum[None, None, :]) <= 5e-7
    assert _rel(ieT, dT * wsum[None, None, :]) <= 5e-7


def test_rt_run_rrs_halo_shard_equals_full(vsm, arch):
    """Multi-GPU decomposition of the Raman pass: a rank that owns a block of recipient points and
    computes on the halo-extended slice reproduces the full run on its block (ndoubl from the full axis)."""
    FT, S = np.float64, 40
    om, pm = _raman_models(vsm, arch, "IQU", 7, S, 3, FT)
    pm.tau_rayl[S - 1] *= 30.0     # one point dominates max(tau*varpi): ndoubl must come from the full axis
    graman = vsm.host_model.get_greek_rayleigh(0.2)
    prs = vsm.CoreRTRaman.RRS(SHIFTS, W_IE, graman)
    full = vsm.CoreRTRaman.rt_run(prs, pm, 1)
    for world in (2, 3):
        for rank in range(world):
            sl = vsm.parallel.shard_slice(S, rank, world)
            part = vsm.CoreRTRaman.rt_run(prs, pm, 1, spec_slice=sl)
            for g, f in zip(part, full):
                assert g.shape[2] == sl.stop - sl.start
                assert _rel(g, f[:, :, sl]) <= 1e-13


def test_rt_run_rrs_blocked_equals_one_pass(vsm, arch):
    """rt_run(RRS, ..., max_points = n): the recipient axis walked in halo-extended blocks (the bounded-footprint mode that replaces
    the reference's host paging of the N x N x nSpec x nRaman arrays) == the one-pass run, bit for bit, also when the last block
    is ragged and the blocks are shorter than the halo."""
    om, pm = _raman_models(vsm, arch, "IQU", 7, 37, 3, np.float64)
    graman = O.get_greek_rayleigh(6.0 / 7.0 * 0.5)
    prs = vsm.CoreRTRaman.RRS(SHIFTS, W_IE, vsm.host_model.GreekCoefs(**vars(graman)))
    full = vsm.CoreRTRaman.rt_run(prs, pm, 1)
    for n in (10, 3):
        blk = vsm.CoreRTRaman.rt_run(prs, pm, 1, max_points=n)
        for a, b in zip(full, blk):
            assert a.shape == b.shape and np.array_equal(a, b)
    dev = vsm.CoreRTRaman.rt_run(prs, pm, 1, max_points=16, device_out=True)
    assert dev[2].shape[0] == 37 and np.array_equal(vsm.Architectures.to_host(dev[2]).transpose(2, 1, 0), full[2])
    assert vsm.CoreRTRaman.raman_bytes_per_point(21, 40, 8) * 20000 > 50e9     # the C5 shape: 60 GB in one pass (measured)


@pytest.mark.parametrize("FT", [np.float64, np.float32])
def test_rt_run_rrs_reference_regression_phase1b(vsm, arch, golden_dir, FT):
    """The reference's own RRS regression (test/test_forward_raman_phase1b.jl; the stored run is Float32): device
    rt_run(RS_type::RRS, model, 1) on the scene of Phase1b_RRS_761-764nm.yaml, inputs from vsmartmom.jl_amd/raman_inputs.py,
    against the stored R, T, ieR, ieT at the reference's gate (atol 1e-6, rtol 0.02) -- and against the oracle."""
    from test_oracle_raman import phase1b_scene, check_phase1b
    fx, bs = phase1b_scene(golden_dir)
    Hm = vsm.host_model
    pm = Hm.model_from_arrays(arch, "IQU", 2 * fx["nstreams"] - 1, fx["sza"], fx["vza"], fx["vaz"], bs.tau_rayl,
                              depol=bs.depol_cabannes, albedo=fx["albedo"], m_max=2, float_type=FT)
    pm.varpi_Cabannes = bs.varpi_cabannes_rs
    prs = vsm.CoreRTRaman.RRS(bs.i_shift, bs.varpi_ie, Hm.GreekCoefs.from_dict(bs.greek_raman))
    got = [np.asarray(g) for g in vsm.CoreRTRaman.rt_run(prs, pm, 1)]
    check_phase1b(fx, got, tight=dict(R_rrs=1e-3, ieR=3e-3))
    om = O.build_model("IQU", 2 * fx["nstreams"] - 1, fx["sza"], fx["vza"], fx["vaz"], bs.tau_rayl,
                       depol=bs.depol_cabannes, albedo=fx["albedo"], m_max=2, FT=FT)
    om.varpi_cabannes = bs.varpi_cabannes_rs
    ref = OR.rt_run_rrs(om, OR.RRS(i_shift=bs.i_shift, varpi_ie=bs.varpi_ie, greek_raman=O.greek_from_dict(bs.greek_raman)))
    tol = 1e-8 if FT == np.float64 else 1e-2
    for name, g, r in zip(("R", "T", "ieR", "ieT"), got, ref):
        assert _rel(g, r) <= tol, (name, _rel(g, r))


def test_c5_full_size_raman_properties(vsm, arch):
    """BASELINE.json configs[4] at its FULL size (nStokes = 3, 20 000 spectral points, K = 40 Raman lines, N = 21, 12 layers,
    m = 0..2; the shape of tools/raman_timing.py), through size-independent properties: (1) on a spectrally uniform column
    with the Raman phase matrix equal to the elastic one, ieR(n1) = dR/d(varpi_Cabannes) * sum of the in-band line weights
    (first-order perturbation identity against central differences of the device's elastic rt_run); (2) recipients whose
    donors all fall outside the band get exactly zero inelastic signal from those lines (edge profile of the weight sum);
    (3) the halo-extended block of rank 1 of 2 (the 2-GPU decomposition of C5) equals the same rows of the full run."""
    FT, S, L, K = np.float64, 20000, 12, 40
    Hm = vsm.host_model
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S, 1))
    tau_abs = np.tile(0.05 * dp, (S, 1))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.05, m_max=2)
    pm = Hm.model_from_arrays(arch, "IQU", 9, 40.0, [30.0], [0.0], **kw)
    pm.varpi_Cabannes = 0.96
    assert pm.quad_points.Nquad * 3 == 21
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 7)
    assert len(shifts) == K
    w_ie = np.linspace(0.5, 1.5, K) * 0.04 / K
    fsc = pm.tau_rayl / (pm.tau_rayl + pm.tau_abs)
    rs = vsm.CoreRTRaman.RRS(shifts, w_ie, pm.greek_rayleigh, fscattRayl=fsc)
    R, T, ieR, ieT = vsm.CoreRTRaman.rt_run(rs, pm, 1)
    assert R.shape == (1, 3, S) and np.all(np.isfinite(ieR)) and np.all(np.isfinite(ieT))
    # the elastic leg of the Raman run equals the elastic rt_run
    h = 1e-5
    small = Hm.model_from_arrays(arch, "IQU", 9, 40.0, [30.0], [0.0], **{**kw, "tau_rayl": tau_rayl[:4], "tau_abs": tau_abs[:4]})
    small.varpi_Cabannes = 0.96
    R0, T0 = vsm.CoreRT.rt_run(small)
    assert _rel(R[:, :, :4], R0) < 1e-10 and _rel(T[:, :, :4], T0) < 1e-10
    small.varpi_Cabannes = 0.96 + h
    Rp, Tp = vsm.CoreRT.rt_run(small)
    small.varpi_Cabannes = 0.96 - h
    Rm, Tm = vsm.CoreRT.rt_run(small)
    dR, dT = (Rp - Rm)[:, :, :1] / (2 * h), (Tp - Tm)[:, :, :1] / (2 * h)     # uniform column: the same at every point
    n1 = np.arange(S)
    wsum = np.zeros(S)
    for sft, w in zip(shifts, w_ie):
        wsum += w * ((n1 + sft >= 0) & (n1 + sft < S))
    assert _rel(ieR, dR * wsum[None, None, :]) <= 5e-7
    assert _rel(ieT, dT * wsum[None, None, :]) <= 5e-7
    assert wsum[0] < wsum[S // 2] and abs(ieR[0, 0, 0]) < abs(ieR[0, 0, S // 2])     # band edge: fewer donors
    # 2-GPU decomposition: rank 1 owns the second half, computes on its halo-extended slice, no exchange
    sl = vsm.parallel.shard_slice(S, 1, 2)
    part = vsm.CoreRTRaman.rt_run(rs, pm, 1, spec_slice=sl)
    for g, f in zip(part, (R, T, ieR, ieT)):
        assert g.shape[2] == S // 2 and _rel(g, f[:, :, sl]) <= 1e-13


def test_c5_k100_blocked_full_size(vsm, arch):
    """The C5 shape with K = 100 Raman lines (nStokes = 3, N = 21, 20 000 points, 12 layers): 150 GB of N x N x nSpec x nRaman
    arrays in one pass.  Walked in blocks of 4 000 recipient points (+ halo: 31 GB per block) it reproduces the perturbation
    identity of test_c5_full_size_raman_properties on the spectrally uniform column, and a window of it equals the direct
    halo-extended run of that window bit for bit."""
    FT, S, L, K = np.float64, 20000, 12, 100
    Hm = vsm.host_model
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S, 1))
    tau_abs = np.tile(0.05 * dp, (S, 1))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.05, m_max=2)
    pm = Hm.model_from_arrays(arch, "IQU", 9, 40.0, [30.0], [0.0], **kw)
    pm.varpi_Cabannes = 0.96
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 3)
    assert len(shifts) == K
    w_ie = np.linspace(0.5, 1.5, K) * 0.04 / K
    rs = vsm.CoreRTRaman.RRS(shifts, w_ie, pm.greek_rayleigh, fscattRayl=pm.tau_rayl / (pm.tau_rayl + pm.tau_abs))
    assert vsm.CoreRTRaman.raman_bytes_per_point(21, K, 8) * S > 140e9
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()       # (whatever earlier tests of the session still hold, e.g. cached work buffers)
    R, T, ieR, ieT = vsm.CoreRTRaman.rt_run(rs, pm, 1, max_points=4000)
    assert torch.cuda.max_memory_allocated() - base < 45e9
    assert R.shape == (1, 3, S) and np.all(np.isfinite(ieR)) and np.all(np.isfinite(ieT))
    h = 1e-5
    small = Hm.model_from_arrays(arch, "IQU", 9, 40.0, [30.0], [0.0], **{**kw, "tau_rayl": tau_rayl[:4], "tau_abs": tau_abs[:4]})
    small.varpi_Cabannes = 0.96 + h
    Rp, _ = vsm.CoreRT.rt_run(small)
    small.varpi_Cabannes = 0.96 - h
    Rm, _ = vsm.CoreRT.rt_run(small)
    dR = (Rp - Rm)[:, :, :1] / (2 * h)
    n1 = np.arange(S)
    wsum = np.zeros(S)
    for sft, w in zip(shifts, w_ie):
        wsum += w * ((n1 + sft >= 0) & (n1 + sft < S))
    assert _rel(ieR, dR * wsum[None, None, :]) <= 5e-7
    win = vsm.CoreRTRaman.rt_run(rs, pm, 1, spec_slice=slice(7900, 8100))     # straddles the block boundary at 8000
    for a, b in zip((R, T, ieR, ieT), win):
        assert np.array_equal(a[:, :, 7900:8100], b)

