"""GPU parity tests of the native-layout run (vsm_run_*, csrc/vsm_native.hip): the layer loop of rt_run with the CompositeLayer in
the layer kernels' strip layout and Stokes blocks that do not couple as independent sub-problems -- against the oracle
(oracle/vsm_oracle.py, the restatement of rt_kernel.jl:175-250 / rt_helpers.jl:102-166 / interaction.jl:207-266), against the
reference-layout layer loop (vsm_layer_forward_multi) and through the C ABI directly."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import vsm_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsm():
    import vsmartmom_jl_amd as v
    v._lib.lib()
    return v


@pytest.fixture(scope="module")
def arch(vsm):
    return vsm.Architectures.GPU(0)


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def _groups(ns, coupling):
    import bench
    return bench.stokes_groups(ns, coupling)


@pytest.mark.parametrize("pol,ns", [("I", 1), ("IQ", 2), ("IQU", 3), ("IQUV", 4)])
def test_stokes_coupling_of_the_fourier_moments(vsm, arch, pol, ns):
    """vsm_stokes_coupling on the Z(m) the device builds (vsm_compute_Z_moments): for m = 0 no phase matrix couples (I,Q) with
    (U,V) (compute_Z_matrices.jl:26-110; the oracle's Z has exact zeros there), for m >= 1 Rayleigh couples I, Q, U."""
    H = vsm.host_model
    S, L = 3, 2
    model = H.model_from_arrays(arch, pol, 11, 40.0, [30.0], [0.0], tau_rayl=np.full((S, L), 0.1), tau_abs=np.full((S, L), 0.01),
                                depol=0.03, albedo=0.1, m_max=2)
    sc = vsm.CoreRT.prepare_scene(model)
    assert sc.coupling is not None and len(sc.coupling) == 3
    qp = O.rt_set_streams_gausslegquad(11, 40.0, [30.0], O.polarization(pol))
    for m in range(3):
        Zpp, Zmp = O.compute_Z_moments(O.polarization(pol), qp.qp_mu, O.get_greek_rayleigh(0.03), m)
        want = 0
        for a in range(ns):
            for b in range(ns):
                if np.any(Zpp[a::ns, b::ns] != 0) or np.any(Zmp[a::ns, b::ns] != 0):
                    want |= 1 << (4 * a + b)
        assert sc.coupling[m] == want, (m, hex(sc.coupling[m]), hex(want))
        for g in _groups(ns, sc.coupling[m]):                       # no block mixes (I,Q) with (U,V) at m = 0
            assert m > 0 or not (g & 0b0011 and g & 0b1100)
    if ns >= 3:
        assert len(_groups(ns, sc.coupling[0])) >= 2 and len(_groups(ns, sc.coupling[1])) < ns


@pytest.mark.parametrize("ns,nq,coupling", [(3, 20, -1), (3, 20, 0x33 | 0x400), (4, 9, 0x33 | 0xCC00), (1, 7, -1), (3, 4, 0x33),
                                             (4, 15, 0x8033), (2, 30, -1)])
def test_native_layout_import_export_roundtrip(vsm, arch, ns, nq, coupling):
    """vsm_run_import / vsm_run_export: reference layout -> native strips -> reference layout is the identity on composites whose
    uncoupled blocks are zero (what a run under that coupling produces); the blocks between the groups come back as zeros."""
    L = vsm._lib.lib()
    N, S = ns * nq, 5
    rng = np.random.default_rng(3)
    dev = "cuda:0"
    grp = _groups(ns, coupling)
    same = np.zeros((ns, ns), dtype=bool)
    for g in grp:
        for a in range(ns):
            for b in range(ns):
                same[a, b] |= bool(g >> a & 1) and bool(g >> b & 1)
    mask = np.tile(same, (nq, nq))                                    # [i, j]: rows / columns interleave the Stokes components
    mats = [rng.standard_normal((S, N, N)) * mask.T[None] for _ in range(4)]   # device tensors hold [s, j, i]
    vecs = [rng.standard_normal((S, N)) for _ in range(2)]
    comp_in = vsm.CoreRT.make_composite_layer(np.float64, arch, (N, N), S)
    comp_out = vsm.CoreRT.make_composite_layer(np.float64, arch, (N, N), S)
    for t, a in zip((comp_in.R_mp, comp_in.R_pm, comp_in.T_pp, comp_in.T_mm, comp_in.J0_p, comp_in.J0_m), mats + vecs):
        t.copy_(torch.as_tensor(a, device=dev))
    for t in (comp_out.R_mp, comp_out.R_pm, comp_out.T_pp, comp_out.T_mm, comp_out.J0_p, comp_out.J0_m):
        t.fill_(float("nan"))
    assert L.vsm_run_supported(N, ns, coupling) == 1
    carr, marr = (C.c_int * 1)(coupling), (C.c_int * 1)(0)
    nbytes = int(L.vsm_run_workspace_bytes(N, ns, S, 1, carr))
    assert nbytes > 0
    ws = torch.full((nbytes // 8,), float("nan"), dtype=torch.float64, device=dev)
    mu = torch.ones(N, dtype=torch.float64, device=dev)
    q = vsm._lib.vsm_quad_f64(mu.data_ptr(), mu.data_ptr(), N, ns, 0, 1.0)
    run = C.c_void_p()
    vsm._lib.check(L.vsm_run_create_f64(C.byref(q), S, 1, marr, carr, C.c_void_p(ws.data_ptr()), nbytes, C.byref(run)))
    try:
        ci, co = (type(comp_in.cstruct()) * 1)(comp_in.cstruct()), (type(comp_out.cstruct()) * 1)(comp_out.cstruct())
        vsm._lib.check(L.vsm_run_import_f64(run, ci, None))
        vsm._lib.check(L.vsm_run_export_f64(run, co, None))
        torch.cuda.synchronize()
    finally:
        L.vsm_run_destroy(run)
    for name in ("R_mp", "R_pm", "T_pp", "T_mm", "J0_p", "J0_m"):
        assert torch.equal(getattr(comp_in, name), getattr(comp_out, name)), name
    assert bool(torch.isfinite(ws).all())           # every word of the native records was written (padding = zeros)


def test_run_create_rejects_what_the_native_kernels_do_not_take(vsm, arch):
    L = vsm._lib.lib()
    assert L.vsm_run_supported(120, 4, -1) == 0 and L.vsm_run_supported(120, 4, 0x8033) == 1     # 60 + 30 + 30
    assert L.vsm_run_supported(99, 3, -1) == 0 and L.vsm_run_supported(99, 3, 0x33) == 1          # 66 + 33
    assert L.vsm_run_supported(96, 1, -1) == 1 and L.vsm_run_supported(97, 1, -1) == 0
    assert L.vsm_run_supported_f32(128, 1, -1) == 1 and L.vsm_run_supported_f32(129, 1, -1) == 0
    assert L.vsm_run_workspace_bytes(120, 4, 10, 1, (C.c_int * 1)(-1)) == 0
    assert L.vsm_run_workspace_bytes_f32(96, 3, 10, 1, (C.c_int * 1)(-1)) * 2 == L.vsm_run_workspace_bytes(96, 3, 10, 1, (C.c_int * 1)(-1))
    mu = torch.ones(120, dtype=torch.float64, device="cuda:0")
    q = vsm._lib.vsm_quad_f64(mu.data_ptr(), mu.data_ptr(), 120, 4, 0, 1.0)
    run = C.c_void_p()
    ws = torch.zeros(16, dtype=torch.float64, device="cuda:0")
    rc = L.vsm_run_create_f64(C.byref(q), 4, 1, (C.c_int * 1)(1), (C.c_int * 1)(-1), C.c_void_p(ws.data_ptr()), 128, C.byref(run))
    assert rc == 2 and b"beyond the native kernels" in L.vsm_last_error()                                # VSM_ERR_UNSUPPORTED
    q = vsm._lib.vsm_quad_f64(mu.data_ptr(), mu.data_ptr(), 60, 3, 0, 1.0)
    rc = L.vsm_run_create_f64(C.byref(q), 4, 1, (C.c_int * 1)(1), (C.c_int * 1)(-1), C.c_void_p(ws.data_ptr()), 128, C.byref(run))
    assert rc == 1 and b"workspace" in L.vsm_last_error()                                                # too small a workspace


SHAPES = [("I", 3, 2), ("I", 9, 3), ("I", 21, 3), ("I", 33, 2), ("I", 55, 2), ("I", 85, 2), ("I", 115, 2), ("I", 117, 2), ("I", 120, 2),
          ("IQ", 21, 3), ("IQ", 53, 2), ("IQ", 57, 2),
          ("IQU", 5, 4), ("IQU", 11, 3), ("IQU", 15, 3), ("IQU", 21, 3), ("IQU", 27, 2), ("IQU", 33, 3), ("IQU", 35, 3),
          ("IQUV", 5, 2), ("IQUV", 11, 4), ("IQUV", 21, 3), ("IQUV", 25, 2), ("IQUV", 35, 2),
          # five and six row tiles (65 .. 96 rows): N = 65, 78, 79 (no spare column), 87, 95 (no spare column), 66 / 93 (IQU), 96 (IQUV: 72 + 24)
          ("I", 123, 2), ("I", 149, 2), ("I", 151, 2), ("I", 167, 2), ("I", 183, 2), ("IQU", 37, 2), ("IQU", 55, 2), ("IQUV", 41, 2)]


@pytest.mark.parametrize("pol,l_trunc,L", SHAPES)
def test_native_run_vs_oracle_and_reference_layout_run(vsm, arch, monkeypatch, pol, l_trunc, L):
    """rt_run through the native-layout run against the oracle (1e-8, the FP64 gate of every rt_run test here) and against the
    reference-layout layer loop (the same operations in another summation order: 1e-10), for sub-problem sizes that land on every
    row-tile count RT = 1..6 (incl. n = 61..64, 77..80, 93..96: no spare columns, the source vectors by mat-vecs over the A-forms), dense and split
    moments, two viewing angles that add zero-weight streams."""
    H = vsm.host_model
    rng = np.random.default_rng(17)
    S = 11
    tau_rayl = np.tile(np.linspace(0.02, 0.3, L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0.5, (S, L))
    F0 = np.zeros((O.polarization(pol).n, S))
    F0[0] = 1.0 + 0.1 * rng.random(S)
    if F0.shape[0] >= 3:
        F0[1], F0[2] = 0.05 * rng.standard_normal(S), 0.04 * rng.standard_normal(S)      # a polarized beam drives the U block
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=2)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], **kw)
    model.F0 = F0
    ns = model.polarization_type.n
    N = model.quad_points.Nquad * ns
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    sc = vsm.CoreRT.prepare_scene(model)
    nat = sc._native_moments()
    want = {i for i in range(3) if max(bin(g).count("1") for g in _groups(ns, sc.coupling[i])) * (N // ns) <= 96}
    assert nat == want, (nat, want)
    sc.run()
    torch.cuda.synchronize()
    vsm._lib.check_device_status("native run")
    Rn, Tn = sc.results_host()
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", False)
    Rl, Tl = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    om = O.build_model(pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], **kw)
    om.F0 = F0
    Ro, To = O.rt_run(om)
    assert np.all(np.isfinite(Rn)) and np.all(np.isfinite(Tn))
    assert _rel(Rn, Ro) < 1e-8 and _rel(Tn, To) < 1e-8, (_rel(Rn, Ro), _rel(Tn, To))
    assert _rel(Rn, Rl) < 1e-10 and _rel(Tn, Tl) < 1e-10, (_rel(Rn, Rl), _rel(Tn, Tl))


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 35), ("IQU", 21), ("I", 25)])
def test_native_run_with_aerosols_mixed_per_point(vsm, arch, monkeypatch, pol, l_trunc):
    """Two aerosols + Rayleigh: Z is mixed per spectral point inside the elemental pre-pass (types.jl:1262-1292), more Fourier
    moments than one run takes per group (MOMENT_BATCH), layers with and without aerosol."""
    H = vsm.host_model
    rng = np.random.default_rng(23)
    S, L = 6, 4
    kw = dict(tau_rayl=np.tile(np.array([0.03, 0.05, 0.1, 0.2]), (S, 1)), tau_abs=10.0 ** rng.uniform(-3, 0, (S, L)), depol=0.03,
              albedo=0.1, m_max=6, tau_aer=np.array([[0.0, 0.0, 0.2, 0.1], [0.03, 0.0, 0.1, 0.3]]))
    aer = lambda M: [M.AerosolOptics(M.GreekCoefs(**vars(O.hg_greek(0.7, 12))), 0.95, 0.1),
                     M.AerosolOptics(M.GreekCoefs(**vars(O.hg_greek(0.5, 8))), 0.9, 0.0)]
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0], [0.0], aerosol_optics=aer(H), **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    sc = vsm.CoreRT.prepare_scene(model)
    assert len(sc._native_moments()) == 7
    sc.run()
    torch.cuda.synchronize()
    Rn, Tn = sc.results_host()
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", False)
    Rl, Tl = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    om = O.build_model(pol, l_trunc, 40.0, [30.0], [0.0], aerosols=[O.AerosolOptics(O.hg_greek(0.7, 12), 0.95, 0.1),
                                                                   O.AerosolOptics(O.hg_greek(0.5, 8), 0.9, 0.0)], **kw)
    Ro, To = O.rt_run(om)
    assert _rel(Rn, Ro) < 1e-8 and _rel(Tn, To) < 1e-8, (_rel(Rn, Ro), _rel(Tn, To))
    assert _rel(Rn, Rl) < 1e-10 and _rel(Tn, Tl) < 1e-10


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 35), ("IQU", 15), ("I", 21), ("IQUV", 21)])
def test_native_run_thick_layers_series_orders_and_pivoted_inverse(vsm, arch, monkeypatch, pol, l_trunc):
    """Thick near-conservative layers over a bright surface: the last doublings and the interactions leave the Horner orders
    (1..8) for the long series (15, 16, 31) and the pivoted Gauss-Jordan inverse (the contract of the reference's LU,
    cpu_batched.jl:32-47); vsm_device_status counts the pivoted inverses and raises no flag."""
    H = vsm.host_model
    rng = np.random.default_rng(29)
    S = 5
    tau_rayl = np.tile(np.array([0.5, 4.0, 30.0]), (S, 1))
    tau_abs = np.tile(np.array([1e-4, 1e-5, 1e-6]), (S, 1)) * (1 + rng.random((S, 1)))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.6, m_max=2)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0], [0.0], **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    Rn, Tn = vsm.CoreRT.rt_run(model)
    st = list(vsm._lib.last_device_status)
    assert st[0] == 0 and st[1] > 0, st
    Ro, To = O.rt_run(O.build_model(pol, l_trunc, 40.0, [30.0], [0.0], **kw))
    assert _rel(Rn, Ro) < 1e-8 and _rel(Tn, To) < 1e-7, (_rel(Rn, Ro), _rel(Tn, To))    # (T: e^-34 of the beam; the legacy path: the same)


@pytest.mark.parametrize("S", [1, 2, 3])
def test_native_run_float32_two_points_per_workgroup_tiny_batches(vsm, arch, monkeypatch, S):
    """Five / six row tiles run two spectral points per workgroup: a batch of one point (its partner is a filler that walks the
    barriers and stores nothing), of two, of three -- each point's numbers are those it has in a larger batch."""
    H = vsm.host_model
    rng = np.random.default_rng(5)
    L, S_big = 3, 6
    tau_rayl = np.tile(np.linspace(0.03, 0.3, L), (S_big, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S_big, L))
    kw = dict(depol=0.03, albedo=0.2, m_max=2)
    mk = lambda n: H.model_from_arrays(arch, "IQU", 57, 40.0, [30.0], [0.0], tau_rayl=tau_rayl[:n], tau_abs=tau_abs[:n], float_type=np.float32, **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    Rb, Tb = vsm.CoreRT.rt_run(mk(S_big))
    # (ndoubl is batch-global, rt_kernel.jl:282-283: the sub-batches here share the big batch's maxima -- every column has the same
    #  tau_rayl and the largest tau varpi of a layer is its Rayleigh part's)
    Rs, Ts = vsm.CoreRT.rt_run(mk(S))
    assert np.array_equal(Rs, Rb[:, :, :S]) and np.array_equal(Ts, Tb[:, :, :S])


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 35), ("I", 21), ("IQU", 57), ("I", 150), ("I", 247)])   # N = 60, 14, 96 (two points per workgroup), 79, 127
def test_native_run_float32_thick_layers_and_mixed_orders(vsm, arch, monkeypatch, pol, l_trunc):
    """The FP32 native kernels beyond the Horner orders: thick near-conservative layers over a bright surface (long series, the
    pivoted Gauss-Jordan inverse -- vsm_device_status counts them and raises no flag), in a batch that ALTERNATES thin and thick
    columns, so that the two points of a workgroup (five / six row tiles) take different inverse paths in the same step: each keeps
    the result of its own order (the arithmetic of a point does not depend on its neighbour: the batch reversed gives the same
    numbers, point by point)."""
    H = vsm.host_model
    S = 7                                           # (odd: the last workgroup of two carries a filler point)
    thick = np.array([0.5, 4.0, 30.0])
    thin = np.array([0.02, 0.03, 0.05])
    tau_rayl = np.stack([thick if s % 2 else thin for s in range(S)])
    tau_abs = np.stack([np.array([1e-4, 1e-5, 1e-6]) * (1 + 0.1 * s) for s in range(S)])
    kw = dict(depol=0.03, albedo=0.6, m_max=2)
    mk = lambda tr, ta: H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0], [0.0], tau_rayl=tr, tau_abs=ta, float_type=np.float32, **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    sc = vsm.CoreRT.prepare_scene(mk(tau_rayl, tau_abs))
    assert len(sc._native_moments()) == 3
    Rn, Tn = vsm.CoreRT.rt_run(mk(tau_rayl, tau_abs))
    st = list(vsm._lib.last_device_status)
    assert st[0] == 0 and st[1] > 0, st
    Rr, Tr = vsm.CoreRT.rt_run(mk(tau_rayl[::-1].copy(), tau_abs[::-1].copy()))
    assert np.array_equal(Rn, Rr[:, :, ::-1]) and np.array_equal(Tn, Tr[:, :, ::-1])
    Ro, To = O.rt_run(O.build_model(pol, l_trunc, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, FT=np.float32, **kw))
    assert _rel(Rn, Ro) < 1e-2, _rel(Rn, Ro)
    # T of these columns is the end of ~ 30 single-precision doublings of a conservative layer (the thick columns' direct beam is
    # e^-34): two Float32 evaluations of it differ by more than the reference's gate for ordinary atmospheres -- both are compared
    # with the FP64 oracle, and the kernels' error may not exceed twice the Float32 oracle's own
    O64 = O.rt_run(O.build_model(pol, l_trunc, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, **kw))[1]
    big = np.abs(O64) > 1e-3 * np.abs(O64).max()
    err = lambda T_: float(np.max(np.abs(T_[big] - O64[big]) / np.abs(O64[big])))
    assert err(Tn) < max(1e-2, 2 * err(To)), (err(Tn), err(To))
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", False)
    Rl, Tl = vsm.CoreRT.rt_run(mk(tau_rayl, tau_abs))
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    assert _rel(Rn, Rl) < 5e-3, _rel(Rn, Rl)


def test_native_run_moment_by_moment_equals_grouped(vsm, arch, monkeypatch):
    """The moments of a run are independent sub-problems: walking them one by one gives the bits of the grouped walk."""
    H = vsm.host_model
    rng = np.random.default_rng(31)
    S, L = 8, 3
    kw = dict(tau_rayl=np.full((S, L), 0.08), tau_abs=10.0 ** rng.uniform(-3, 0, (S, L)), depol=0.03, albedo=0.3, m_max=2)
    model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], **kw)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", True)
    a = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", False)
    b = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_native_run_layer_by_layer_through_the_c_abi(vsm, arch):
    """vsm_run_create / vsm_run_layer / vsm_run_export called directly, exporting after EVERY layer step: the composite equals the
    oracle's CompositeLayer layer by layer (rt_kernel.jl:175-250: TOA copy, then interaction!(_11)), matrices and source vectors,
    with NaN-poisoned workspace."""
    H, L = vsm.host_model, vsm._lib.lib()
    rng = np.random.default_rng(37)
    S, Nz = 4, 3
    kw = dict(tau_rayl=np.tile(np.array([0.05, 0.1, 0.3]), (S, 1)), tau_abs=10.0 ** rng.uniform(-3, 0, (S, Nz)), depol=0.03,
              albedo=0.0, m_max=1)
    model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], **kw)
    sc = vsm.CoreRT.prepare_scene(model)
    N, ns = sc.N, 3
    om = O.build_model("IQU", 35, 40.0, [30.0], [0.0], **kw)
    pol, qp = om.pol, om.quad_points
    FT = np.float64
    F0 = np.zeros((ns, S))
    F0[0] = 1.0
    for im in (0, 1):
        mom = sc.moments[im]
        carr, marr = (C.c_int * 1)(int(sc.coupling[im])), (C.c_int * 1)(im)
        nbytes = int(L.vsm_run_workspace_bytes(N, ns, S, 1, carr))
        ws = torch.full((nbytes // 8,), float("nan"), dtype=torch.float64, device="cuda:0")
        q = sc.dq.cstruct()
        run = C.c_void_p()
        vsm._lib.check(L.vsm_run_create_f64(C.byref(q), S, 1, marr, carr, C.c_void_p(ws.data_ptr()), nbytes, C.byref(run)))
        lods = O.construct_core_optical_properties(om, im)
        ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
        added, comp_o = O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
        comp = vsm.CoreRT.make_composite_layer(FT, arch, (N, N), S)
        try:
            for iz in range(Nz):
                ly = mom["layers"][iz]
                p = ly["props"]
                zpp, zmp = (C.c_void_p * 1)(p.Zpp.data_ptr()), (C.c_void_p * 1)(p.Zmp.data_ptr())
                vsm.CoreRT.run_layer_native_(run, 1, int(ly["nd"]), ly["dtau"], p.varpi, ly["tau_sum"], sc.F0, 0, zpp, zmp,
                                             p.z_stride, None, iz == 0)
                cc = (type(comp.cstruct()) * 1)(comp.cstruct())
                vsm._lib.check(L.vsm_run_export_f64(run, cc, None))
                torch.cuda.synchronize()
                props = O.expand_optical_properties(lods[iz], FT)
                O.rt_kernel(pol, added, comp_o, props, ifaces[iz], tau_sum_all[:, iz].astype(FT), im, qp, iz + 1, F0, FT)
                for name in ("R_mp", "R_pm", "T_pp", "T_mm"):
                    got = vsm.CoreRT.from_device_matrix(getattr(comp, name))
                    assert _rel(got, getattr(comp_o, name)) < 1e-10, (im, iz, name, _rel(got, getattr(comp_o, name)))
                for name in ("J0_p", "J0_m"):
                    got = vsm.Architectures.to_host(getattr(comp, name))
                    ref = np.asarray(getattr(comp_o, name)).reshape(S, N)
                    assert _rel(got, ref) < 1e-10, (im, iz, name)
        finally:
            L.vsm_run_destroy(run)


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 11), ("IQU", 35), ("IQUV", 21)])
def test_linearized_run_with_m0_as_stokes_iq_scene(vsm, arch, monkeypatch, pol, l_trunc):
    """rt_run(model, lin_model, 0, NGas, 1) on a batch large enough for the m = 0 reduction (CoreRTLin.REDUCE_M0: the moment m = 0
    as a Stokes_IQ scene -- no phase matrix couples (I,Q) with (U,V) at m = 0, compute_Z_matrices.jl:26-110): R, T, Rdot, Tdot equal
    the full-Stokes walk to rounding and the oracle's linearized run (doubling_lin.jl:216-339, interaction_lin.jl:217-331)."""
    from oracle import vsm_oracle_lin as OL
    H = vsm.host_model
    rng = np.random.default_rng(41)
    S, L = 70, 3
    tau_rayl = np.tile(0.04 * np.ones(L), (S, 1))
    ga, gb = 10.0 ** rng.uniform(-2.5, -0.3, (S, L)), 10.0 ** rng.uniform(-2.5, -0.5, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=ga + gb, depol=0.0279, m_max=2)
    pm = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0, 5.0], [0.0, 60.0], albedo=0.2, **kw)
    monkeypatch.setattr(vsm.CoreRTLin, "REDUCE_M0", True)
    sc = vsm.CoreRTLin.SceneLin(pm, H.LinModel([ga, gb]), 0, 2, 1)
    assert sc.sub0 is not None and sc.sub0.N == 2 * pm.quad_points.Nquad
    sc.run()
    torch.cuda.synchronize()
    red = sc.results_host()
    monkeypatch.setattr(vsm.CoreRTLin, "REDUCE_M0", False)
    full = vsm.CoreRTLin.rt_run_lin(pm, H.LinModel([ga, gb]), 0, 2, 1)
    monkeypatch.setattr(vsm.CoreRTLin, "REDUCE_M0", True)
    for a, b in zip(red, full):
        assert _rel(a, b) < 1e-11
    sub = slice(0, S, 23)                                   # the oracle on a sample of the points
    om = O.build_model(pol, l_trunc, 40.0, [30.0, 5.0], [0.0, 60.0], albedo=0.2,
                       **dict(kw, tau_rayl=tau_rayl[sub], tau_abs=(ga + gb)[sub]))
    Ro, To, Rdo, Tdo = OL.rt_run_lin(om, OL.LinModel([ga[sub], gb[sub]]))
    assert _rel(red[0][:, :, sub], Ro) < 1e-9 and _rel(red[1][:, :, sub], To) < 1e-9
    assert _rel(red[2][:, :, sub], Rdo) < 1e-8 and _rel(red[3][:, :, sub], Tdo) < 1e-8


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 9), ("IQUV", 7)])
def test_raman_run_with_m0_as_stokes_iq_scene(vsm, arch, monkeypatch, pol, l_trunc):
    """rt_run(RS_type::RRS, model) with the moment m = 0 as a Stokes_IQ scene (CoreRTRaman.REDUCE_M0) equals the full-Stokes walk
    to rounding, elastic and inelastic outputs (doubling_inelastic.jl:13-164, interaction_inelastic.jl:319-521; the oracle
    comparisons of tests/test_gpu_raman.py run with the reduction on: it is the default)."""
    H, R = vsm.host_model, vsm.CoreRTRaman
    rng = np.random.default_rng(43)
    S, L, K = 40, 3, 6
    tau_rayl = np.tile(np.array([0.05, 0.1, 0.15]), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, -0.5, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.1, m_max=2)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0], [0.0], **kw)
    model.varpi_Cabannes = 0.96
    shifts = np.array([-9, -4, -1, 2, 5, 11])
    rs = R.RRS(shifts, np.full(K, 0.04 / K), H.get_greek_rayleigh(0.75))
    monkeypatch.setattr(R, "REDUCE_M0", True)
    sc = R.SceneRRS(rs, model)
    assert sc.sub0 is not None and sc.sub0.N == 2 * model.quad_points.Nquad
    red = R.rt_run(rs, model, 1)
    monkeypatch.setattr(R, "REDUCE_M0", False)
    full = R.rt_run(rs, model, 1)
    monkeypatch.setattr(R, "REDUCE_M0", True)
    for a, b in zip(red, full):
        assert _rel(a, b) < 1e-11
        assert a.shape[1] < 3 or np.all(a[:, 2:, :] == b[:, 2:, :]) or _rel(a[:, 2:, :], b[:, 2:, :]) < 1e-11


@pytest.mark.parametrize("pol,l_trunc", [("I", 9), ("IQU", 21), ("IQU", 35), ("IQUV", 25), ("I", 120), ("IQU", 57), ("I", 130), ("I", 150),
                                         ("I", 170), ("IQU", 45), ("IQUV", 39), ("IQU", 61), ("I", 235), ("I", 247)])
def test_native_run_float32_models(vsm, arch, monkeypatch, pol, l_trunc):
    """A Float32 model through the native-layout run (vsm_run_*_f32: FP32 records and arithmetic, blocks of up to 128 rows -- every
    row-tile count 1..8, rider columns and the mat-vec source path, two points per workgroup at five and six row tiles with an odd
    batch): within the reference's own FP32 gate of the oracle's Float32 run (test/test_float32.jl:58-64: 1e-2 end to end)."""
    H = vsm.host_model
    rng = np.random.default_rng(47)
    S, L = 9, 3
    tau_rayl = np.tile(np.linspace(0.03, 0.3, L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=2)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], float_type=np.float32, **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    sc = vsm.CoreRT.prepare_scene(model)
    ns = model.polarization_type.n
    N = model.quad_points.Nquad * ns
    nat = sc._native_moments()
    assert nat == {i for i in range(3) if max(bin(g).count("1") for g in _groups(ns, sc.coupling[i])) * (N // ns) <= 128} and nat
    sc.run()
    torch.cuda.synchronize()
    vsm._lib.check_device_status("native run (f32)")
    Rn, Tn = sc.results_host()
    assert Rn.dtype == np.float32
    Ro, To = O.rt_run(O.build_model(pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], FT=np.float32, **kw))
    assert _rel(Rn, Ro) < 1e-2 and _rel(Tn, To) < 1e-2
    # and against the Float32 kernels of the reference-layout layer loop (the same Float32 algorithm -- its ndoubl is floor-limited,
    # so neither is close to an FP64 run --, rounded in single at every operation there)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", False)
    Rl, Tl = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    assert _rel(Rn, Rl) < 2e-3 and _rel(Tn, Tl) < 2e-3, (_rel(Rn, Rl), _rel(Tn, Tl))


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 35), ("I", 15)])
def test_native_run_diagonal_steps_of_rayleigh_layers_at_high_moments(vsm, arch, monkeypatch, pol, l_trunc):
    """The Rayleigh phase matrix vanishes for m >= 3: Rayleigh-only layers above an aerosol layer are diagonal steps of every Stokes
    block there (only the two diagonals of T change while the composite is still diagonal; a scaling of the composite below the first
    aerosol layer), and U at m = 0 is a diagonal step in every layer -- rt_run equals the oracle (which forms every product, like
    the reference) and the reference-layout layer loop."""
    H = vsm.host_model
    rng = np.random.default_rng(53)
    S, L = 6, 6
    tau_aer = np.array([[0.0, 0.0, 0.0, 0.15, 0.0, 0.25]])       # Rayleigh | Rayleigh | Rayleigh | mixed | Rayleigh | mixed
    kw = dict(tau_rayl=np.tile(np.linspace(0.02, 0.12, L), (S, 1)), tau_abs=10.0 ** rng.uniform(-3, -0.5, (S, L)), depol=0.03,
              albedo=0.15, m_max=7, tau_aer=tau_aer)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0], [0.0],
                                aerosol_optics=[H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.7, 10))), 0.95, 0.0)], **kw)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    sc = vsm.CoreRT.prepare_scene(model)
    assert len(sc._native_moments()) == 8
    ns = model.polarization_type.n
    diag = lambda m, iz: all(not any(sc._layer_coupling(m, iz) >> (4 * a + b) & 1 for a in range(ns) for b in range(ns) if g >> a & 1 and g >> b & 1)
                             for g in _groups(ns, sc.coupling[m]))
    assert diag(5, 0) and diag(5, 4) and not diag(5, 3) and not diag(1, 0)     # Rayleigh layers at m = 5; an aerosol layer; m = 1
    sc.run()
    torch.cuda.synchronize()
    Rn, Tn = sc.results_host()
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", False)
    Rl, Tl = vsm.CoreRT.rt_run(model)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", True)
    Ro, To = O.rt_run(O.build_model(pol, l_trunc, 40.0, [30.0], [0.0], aerosols=[O.AerosolOptics(O.hg_greek(0.7, 10), 0.95, 0.0)], **kw))
    assert _rel(Rn, Ro) < 1e-8 and _rel(Tn, To) < 1e-8, (_rel(Rn, Ro), _rel(Tn, To))
    assert _rel(Rn, Rl) < 1e-10 and _rel(Tn, Tl) < 1e-10


@pytest.mark.parametrize("N,ns", [(1, 1), (3, 3), (9, 3), (13, 1), (16, 4), (21, 3), (29, 1), (32, 4), (33, 3), (45, 3), (47, 1), (48, 4), (57, 3),
                                  (60, 3), (61, 1), (63, 3), (64, 4)])
@pytest.mark.parametrize("dsym", [False, True])
@pytest.mark.parametrize("scale", [1.0, 4.0, 12.0])
def test_standalone_interaction_native(vsm, arch, N, ns, dsym, scale):
    """interaction!(::ScatteringInterface_11) (interaction.jl:207-266) through vsm_interaction_f64 for N <= 64: k_ia_native, the
    native kernel body on the reference's [N,N,S] arrays -- every row-tile / k-step family (rider columns and the mat-vec source
    path of N = 13..16, 29..32, 45..48, 61..64), odd N (4-byte DMA pieces, 8-byte accesses) and even N (16-byte pieces, permlane pairs), an
    added layer with all four matrices in memory and a D-symmetric one (r+- = D r-+ D, t-- = D t++ D derived in the kernel, as
    doubling! leaves it), at reflectances that take the order-7 inverse in four products, the long orders (out of line) and the
    pivoted Gauss-Jordan -- against the oracle."""
    FT = np.float64
    rng = np.random.default_rng(100 * N + int(scale))
    S = 5
    CR = vsm.CoreRT

    def refl(sc):
        return (sc * rng.random((S, N, N)) / N).astype(FT)

    def trans():
        return (np.eye(N)[None] * rng.uniform(0.3, 0.95, (S, N, 1)) + 0.05 * rng.random((S, N, N)) / N).astype(FT)
    comp = O.CompositeLayer(refl(0.1 * scale), refl(0.1 * scale), trans(), trans(), rng.random((S, N)), rng.random((S, N)))
    r, t = refl(0.075 * scale), trans()
    if dsym:
        D = np.where(np.arange(N) % ns >= 2, -1.0, 1.0)
        add = O.AddedLayer(r, t, D[None, :, None] * r * D[None, None, :], D[None, :, None] * t * D[None, None, :],
                           rng.random((S, N)), rng.random((S, N)))
    else:
        add = O.AddedLayer(r, t, refl(0.075 * scale), trans(), rng.random((S, N)), rng.random((S, N)))
    conv_m = lambda x: CR.to_device_matrix(x, arch, FT)
    conv_v = vsm.Architectures.array_type(arch)
    pc = CR.make_composite_layer(FT, arch, (N, N), S)
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        getattr(pc, k).copy_(conv_m(getattr(comp, k)))
    pc.J0_p.copy_(conv_v(comp.J0_p))
    pc.J0_m.copy_(conv_v(comp.J0_m))
    pa = CR.make_added_layer(FT, arch, (N, N), S, d_symmetric=ns if dsym else 0)
    for k in ("r_mp", "t_pp") if dsym else ("r_mp", "t_pp", "r_pm", "t_mm"):
        getattr(pa, k).copy_(conv_m(getattr(add, k)))
    pa.j0_p.copy_(conv_v(add.j0_p))
    pa.j0_m.copy_(conv_v(add.j0_m))
    O.interaction("11", comp, add, FT)
    CR.interaction_("11", pc, pa)
    torch.cuda.synchronize()
    f, h = CR.from_device_matrix, vsm.Architectures.to_host
    got = dict(R_mp=f(pc.R_mp), R_pm=f(pc.R_pm), T_pp=f(pc.T_pp), T_mm=f(pc.T_mm), J0_p=h(pc.J0_p), J0_m=h(pc.J0_m))
    for k, v in got.items():
        want = getattr(comp, k)
        assert np.all(np.isfinite(v)), k
        assert np.max(np.abs(v - want)) / np.max(np.abs(want)) < 1e-10, (k, N, dsym, scale)


@pytest.mark.parametrize("N,ns,dsym", [(60, 3, True), (60, 3, False), (20, 4, False), (33, 3, True)])
def test_standalone_interaction_arrays_on_8_byte_boundaries(vsm, arch, N, ns, dsym):
    """vsm_interaction_f64 makes no alignment promise beyond the element type: composite and added-layer arrays that start on an
    8-byte boundary only (views one element into a larger allocation) must take the 4-byte DMA pieces and the 8-byte-aligned pair
    accesses of k_ia_native and give the oracle's result."""
    FT = np.float64
    rng = np.random.default_rng(7 * N + dsym)
    S = 4
    CR = vsm.CoreRT
    dev = torch.device("cuda:0")

    def refl(sc):
        return (sc * rng.random((S, N, N)) / N).astype(FT)

    def trans():
        return (np.eye(N)[None] * rng.uniform(0.3, 0.95, (S, N, 1)) + 0.05 * rng.random((S, N, N)) / N).astype(FT)
    comp = O.CompositeLayer(refl(0.1), refl(0.1), trans(), trans(), rng.random((S, N)), rng.random((S, N)))
    r, t = refl(0.075), trans()
    if dsym:
        D = np.where(np.arange(N) % ns >= 2, -1.0, 1.0)
        add = O.AddedLayer(r, t, D[None, :, None] * r * D[None, None, :], D[None, :, None] * t * D[None, None, :],
                           rng.random((S, N)), rng.random((S, N)))
    else:
        add = O.AddedLayer(r, t, refl(0.075), trans(), rng.random((S, N)), rng.random((S, N)))
    pc = CR.make_composite_layer(FT, arch, (N, N), S)
    pa = CR.make_added_layer(FT, arch, (N, N), S, d_symmetric=ns if dsym else 0)

    def odd_view(shape):   # a tensor of this shape whose first element sits 8 bytes into a 256-byte aligned allocation
        n = int(np.prod(shape))
        buf = torch.zeros(n + 1, dtype=torch.float64, device=dev)
        v = buf[1:].view(*shape)
        assert v.data_ptr() % 16 == 8
        return v
    conv_m = lambda x: CR.to_device_matrix(x, arch, FT)
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        v = odd_view((S, N, N))
        v.copy_(conv_m(getattr(comp, k)))
        setattr(pc, k, v)
    for k in ("r_mp", "t_pp") if dsym else ("r_mp", "t_pp", "r_pm", "t_mm"):
        v = odd_view((S, N, N))
        v.copy_(conv_m(getattr(add, k)))
        setattr(pa, k, v)
    conv_v = vsm.Architectures.array_type(arch)
    pc.J0_p.copy_(conv_v(comp.J0_p))
    pc.J0_m.copy_(conv_v(comp.J0_m))
    pa.j0_p.copy_(conv_v(add.j0_p))
    pa.j0_m.copy_(conv_v(add.j0_m))
    O.interaction("11", comp, add, FT)
    CR.interaction_("11", pc, pa)
    torch.cuda.synchronize()
    f, h = CR.from_device_matrix, vsm.Architectures.to_host
    got = dict(R_mp=f(pc.R_mp), R_pm=f(pc.R_pm), T_pp=f(pc.T_pp), T_mm=f(pc.T_mm), J0_p=h(pc.J0_p), J0_m=h(pc.J0_m))
    for k, v in got.items():
        want = getattr(comp, k)
        assert np.max(np.abs(v - want)) / np.max(np.abs(want)) < 1e-10, (k, N, dsym)


# ---- the drop-in form: rt_kernel_ called in the reference driver's order on ONE CompositeLayer (per-composite registry) ----------
def _spy_native(vsm, monkeypatch):
    calls = []
    orig = vsm.CoreRT._rt_kernel_native

    def spy(comp, props, tau_sum, m, dq, iz, F0, dtau, nd):
        ok = orig(comp, props, tau_sum, m, dq, iz, F0, dtau, nd)
        calls.append((int(m), int(iz), bool(ok)))
        return ok
    monkeypatch.setattr(vsm.CoreRT, "_rt_kernel_native", spy)
    return calls


@pytest.mark.parametrize("pol,l_trunc,L,FT", [("IQU", 35, 3, np.float64), ("IQU", 21, 4, np.float64), ("IQUV", 21, 3, np.float64),
                                               ("I", 55, 2, np.float64), ("IQU", 35, 3, np.float32), ("IQUV", 35, 2, np.float64)])
def test_reference_call_order_reaches_the_native_run_through_rt_kernel(vsm, arch, monkeypatch, pol, l_trunc, L, FT):
    """CoreRT.REFERENCE_ORDER: Scene.run issues exactly the call sequence of the unpatched driver (rt_run.jl:383-470: `for m` outside
    `for iz`, rt_kernel! on ONE CompositeLayer, then create_surface_layer! / interaction! / postprocessing_vza!).  rt_kernel_ keeps
    that composite in native layout from the TOA call on (a one-moment vsm_run per composite, NATIVE_DROPIN) and the first
    consumer of its arrays exports it.  Results: the oracle's (1e-8; FP32 at the reference's FP32 gate), the layers-outside native
    run's bits where the blocks are the same, the reference-layout walk's to rounding; the registry took every layer step of every
    moment whose blocks fit (IQUV, N = 80: m = 0 only -- the dense moments stay on vsm_layer_forward)."""
    H = vsm.host_model
    rng = np.random.default_rng(61)
    S = 9
    tau_rayl = np.tile(np.linspace(0.02, 0.3, L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0.5, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=2)
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], float_type=FT, **kw)
    ns = model.polarization_type.n
    N = model.quad_points.Nquad * ns
    Ra, Ta = vsm.CoreRT.rt_run(model)                                  # layers outside moments, vsm_run_* from Scene.run
    monkeypatch.setattr(vsm.CoreRT, "REFERENCE_ORDER", True)
    calls = _spy_native(vsm, monkeypatch)
    Rb, Tb = vsm.CoreRT.rt_run(model)
    sc = vsm.CoreRT.prepare_scene(model)
    fits = [max(bin(g).count("1") for g in _groups(ns, sc.coupling[m])) * (N // ns) <= (128 if FT == np.float32 else 96) for m in range(3)]
    assert [c for c in calls if c[2]] == [(m, iz, True) for m in range(3) if fits[m] for iz in range(1, L + 1)], calls
    assert any(fits)
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_DROPIN", False)
    Rc, Tc = vsm.CoreRT.rt_run(model)                                  # the same order on the reference-layout kernels
    om = O.build_model(pol, l_trunc, 40.0, [30.0, 10.0], [0.0, 75.0], **kw)
    Ro, To = O.rt_run(om)
    if FT == np.float64:
        assert np.array_equal(Ra, Rb) and np.array_equal(Ta, Tb)
        assert _rel(Rb, Rc) < 1e-10 and _rel(Tb, Tc) < 1e-10
        assert _rel(Rb, Ro) < 1e-8 and _rel(Tb, To) < 1e-8, (_rel(Rb, Ro), _rel(Tb, To))
    else:
        assert _rel(Rb, Ro) < 1e-2 and _rel(Tb, To) < 1e-2 and _rel(Rb, Ra) < 1e-4


def test_rt_kernel_registry_lazy_export_layer_by_layer(vsm, arch):
    """rt_kernel_ called directly like the reference's driver does, reading the composite after EVERY layer: the first access
    exports the native copy (CompositeLayer.materialize), the next rt_kernel_ call -- no longer at the TOA -- continues on the
    reference's arrays; both ways the composite equals the oracle's layer by layer (rt_kernel.jl:175-250)."""
    H, CR = vsm.host_model, vsm.CoreRT
    rng = np.random.default_rng(67)
    S, Nz = 4, 4
    kw = dict(tau_rayl=np.tile(np.array([0.05, 0.1, 0.2, 0.3]), (S, 1)), tau_abs=10.0 ** rng.uniform(-3, 0, (S, Nz)), depol=0.03,
              albedo=0.0, m_max=1)
    model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], **kw)
    sc = CR.prepare_scene(model)
    N, FT = sc.N, np.float64
    om = O.build_model("IQU", 35, 40.0, [30.0], [0.0], **kw)
    F0 = np.zeros((3, S))
    F0[0] = 1.0
    for im in (0, 1):
        for peek_every in (1, 3, 99):        # export after every layer / once in the middle / only at the end
            lods = O.construct_core_optical_properties(om, im)
            ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
            added_o, comp_o = O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
            comp = CR.make_composite_layer(FT, arch, (N, N), S)
            for t in comp._arr.values():
                t.fill_(float("nan"))
            exported = False
            for iz in range(Nz):
                ly = sc.moments[im]["layers"][iz]
                CR.rt_kernel_(sc.pol, sc.added, comp, ly["props"], ly["iface"], ly["tau_sum"], im, sc.dq, arch, iz + 1, sc.F0, FT,
                              model.numerics, dtau=ly["dtau"], ndoubl=ly["nd"])
                O.rt_kernel(om.pol, added_o, comp_o, O.expand_optical_properties(lods[iz], FT), ifaces[iz],
                            tau_sum_all[:, iz].astype(FT), im, om.quad_points, iz + 1, F0, FT)
                assert (comp._native is not None) == (not exported), (iz, peek_every)   # native until somebody read the arrays
                if (iz + 1) % peek_every == 0 or iz == Nz - 1:
                    exported = True
                    for name in ("R_mp", "R_pm", "T_pp", "T_mm"):
                        got = CR.from_device_matrix(getattr(comp, name))
                        assert _rel(got, getattr(comp_o, name)) < 1e-10, (im, iz, name)
                    assert comp._native is None
                    for name in ("J0_p", "J0_m"):
                        got = vsm.Architectures.to_host(getattr(comp, name))
                        assert _rel(got, np.asarray(getattr(comp_o, name)).reshape(S, N)) < 1e-10, (im, iz, name)
    torch.cuda.synchronize()
    vsm._lib.check_device_status("registry")


def test_rt_kernel_registry_reopens_the_run_when_a_layer_couples_more(vsm, arch, monkeypatch):
    """A column whose TOA layer holds only a Henyey-Greenstein aerosol (its phase matrix couples I with I alone: three one-component
    blocks) above Rayleigh layers (I-Q coupled): the registry re-opens the run under the wider mask at the first Rayleigh layer
    (export -> vsm_run_create -> vsm_run_import).  Same results as the layers-outside run (whose mask is the OR over the run's
    scatterers from the start) and as the oracle."""
    H = vsm.host_model
    rng = np.random.default_rng(71)
    S, L = 5, 3
    tau_rayl = np.tile(np.array([0.0, 0.1, 0.2]), (S, 1))
    kw = dict(tau_rayl=tau_rayl, tau_abs=10.0 ** rng.uniform(-3, -0.5, (S, L)), depol=0.03, albedo=0.1, m_max=3,
              tau_aer=np.array([[0.3, 0.0, 0.0]]))
    model = H.model_from_arrays(arch, "IQU", 21, 40.0, [30.0], [0.0],
                                aerosol_optics=[H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.6, 10))), 0.9, 0.0)], **kw)
    Ra, Ta = vsm.CoreRT.rt_run(model)
    opened = []
    orig = vsm.CoreRT._native_open
    monkeypatch.setattr(vsm.CoreRT, "_native_open", lambda comp, dq, m, mask, imp: (opened.append((m, mask, imp)), orig(comp, dq, m, mask, imp))[1])
    monkeypatch.setattr(vsm.CoreRT, "REFERENCE_ORDER", True)
    Rb, Tb = vsm.CoreRT.rt_run(model)
    assert (0, 0x1, False) in opened and any(m == 0 and imp and mask & 0x12 for m, mask, imp in opened), opened
    om = O.build_model("IQU", 21, 40.0, [30.0], [0.0], aerosols=[O.AerosolOptics(O.hg_greek(0.6, 10), 0.9, 0.0)], **kw)
    Ro, To = O.rt_run(om)
    assert _rel(Rb, Ra) < 1e-11 and _rel(Tb, Ta) < 1e-11, (_rel(Rb, Ra), _rel(Tb, Ta))
    assert _rel(Rb, Ro) < 1e-8 and _rel(Tb, To) < 1e-8, (_rel(Rb, Ro), _rel(Tb, To))


def test_run_layer_flags_a_phase_matrix_outside_the_declared_coupling(vsm, arch):
    """VSM_DEVSTAT_MASK: vsm_run_layer checks the layer's phase matrices against the masks on the device.  A run created for the
    blocks (I,Q) | U that is handed a Z with a non-zero (Q,U) element, and a layer whose `layer_coupling_h` calls the U block zero
    while its Z has a U-U element, both raise the flag (vsm_device_status; check_device_status throws); the clean call does not."""
    H, L, CR = vsm.host_model, vsm._lib.lib(), vsm.CoreRT
    S = 3
    kw = dict(tau_rayl=np.full((S, 2), 0.1), tau_abs=np.full((S, 2), 0.01), depol=0.03, albedo=0.1, m_max=1)
    model = H.model_from_arrays(arch, "IQU", 11, 40.0, [30.0], [0.0], **kw)
    sc = CR.prepare_scene(model)
    N, ns = sc.N, 3
    ly = sc.moments[0]["layers"][0]
    p = ly["props"]
    mask0 = int(sc.coupling[0])
    assert mask0 == 0x33                    # Rayleigh at m = 0: I-Q coupled, U alone and -- its block being zero -- a diagonal step

    def one_call(Zpp, Zmp, run_mask, layer_mask):
        carr, marr = (C.c_int * 1)(run_mask), (C.c_int * 1)(0)
        nbytes = int(L.vsm_run_workspace_bytes(N, ns, S, 1, carr))
        ws = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda:0")
        q = sc.dq.cstruct()
        run = C.c_void_p()
        vsm._lib.check(L.vsm_run_create_f64(C.byref(q), S, 1, marr, carr, C.c_void_p(ws.data_ptr()), nbytes, C.byref(run)))
        try:
            zpp, zmp = (C.c_void_p * 1)(Zpp.data_ptr()), (C.c_void_p * 1)(Zmp.data_ptr())
            CR.run_layer_native_(run, 1, int(ly["nd"]), ly["dtau"], p.varpi, ly["tau_sum"], sc.F0, 0, zpp, zmp, 0, None, True,
                                 (C.c_int * 1)(layer_mask))
            torch.cuda.synchronize()
        finally:
            L.vsm_run_destroy(run)
        flags = (C.c_int * 4)()
        vsm._lib.check(L.vsm_device_status(flags, 1, None))
        return flags[0]

    assert one_call(p.Zpp, p.Zmp, mask0, mask0) == 0
    Zbad = p.Zpp.clone()
    Zbad[0, 2, 1] = 1e-3                     # element (i = 1 (Q), j = 2 (U)): couples the two blocks of the run
    assert one_call(Zbad, p.Zmp, mask0, mask0) & 4
    Zuu = p.Zmp.clone()
    Zuu[0, 5, 2] = 1e-3                      # (i = 2 (U), j = 5 (U)): inside the U block, which layer_mask calls zero
    assert one_call(p.Zpp, Zuu, mask0 | 0x400, mask0) & 4
    assert one_call(p.Zpp, Zuu, mask0 | 0x400, mask0 | 0x400) == 0
    # the Python host raises on the flag at the end of a run
    zpp, zmp = (C.c_void_p * 1)(Zbad.data_ptr()), (C.c_void_p * 1)(p.Zmp.data_ptr())
    carr = (C.c_int * 1)(mask0)
    nbytes = int(L.vsm_run_workspace_bytes(N, ns, S, 1, carr))
    ws = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda:0")
    q = sc.dq.cstruct()
    run = C.c_void_p()
    vsm._lib.check(L.vsm_run_create_f64(C.byref(q), S, 1, (C.c_int * 1)(0), carr, C.c_void_p(ws.data_ptr()), nbytes, C.byref(run)))
    CR.run_layer_native_(run, 1, int(ly["nd"]), ly["dtau"], p.varpi, ly["tau_sum"], sc.F0, 0, zpp, zmp, 0, None, True, carr)
    torch.cuda.synchronize()
    L.vsm_run_destroy(run)
    with pytest.raises(vsm.VSMError, match="VSM_DEVSTAT_MASK"):
        vsm._lib.check_device_status("mask test")
