"""ctypes front of oracle/vsm_oracle_c.c (C + OpenMP restatement of the elastic forward path) -- test infrastructure and
the CPU baseline of bench.py; NOT the product path.  The host-side inputs (streams, Z moments, layer optics, ndoubl) come
from the pinned numpy oracle (vsm_oracle.py); the C code runs the per-spectral-point loop (elemental -> doubling ->
interaction -> Lambertian surface -> post-processing) with one LU per point and OpenMP threads over the spectral axis,
the structure of the reference's CPU path (src/CoreRT/tools/cpu_batched.jl:25-82)."""
import ctypes as C
import math
import os

import numpy as np

from . import vsm_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libvsm_oracle_c.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("oracle/_build/libvsm_oracle_c.so not built: run `make -C oracle`")
        _lib = C.CDLL(LIB_PATH)
        D, I = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _lib.vsm_oracle_c_rt_run.restype = C.c_int
        _lib.vsm_oracle_c_rt_run.argtypes = [C.c_int, C.c_int, D, D, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, D, D, D, I, D, D,
                                            D, C.c_double, C.c_int, I, D, C.c_int, D, D]
    return _lib


def rt_run(model: O.RTModel, nthreads: int = 0, ndoubl_float_type=np.float64):
    """rt_run(model) -> (R_SFI, T_SFI) [nVZA, nStokes, S] for FP64 scenes with ONE scatterer (Rayleigh) + absorption over a
    scalar Lambertian surface whose layers all scatter -- the scenes of BASELINE configs C1 / C2 / C4."""
    if model.aerosol_optics or np.ndim(model.albedo) != 0:
        raise ValueError("the C restatement covers Rayleigh + absorption over a scalar Lambertian surface")
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N, n = qp.Nquad * pol.n, pol.n
    M = model.m_max + 1
    lods = O.construct_core_optical_properties(model, 0)
    ifaces, tau_sum = O.extract_effective_props(lods, np.float64)
    if any(t != "11" for t in ifaces):
        raise ValueError("every layer must scatter (ScatteringInterface_11)")
    tau = np.ascontiguousarray(np.stack([np.atleast_1d(lo.tau) for lo in lods], axis=1), dtype=np.float64)
    varpi = np.ascontiguousarray(np.stack([np.broadcast_to(lo.varpi, (S,)) for lo in lods], axis=1), dtype=np.float64)
    # (ndoubl_float_type: the float type whose doubling_number rule applies -- rt_kernel.jl:266-287; Float32 models double less)
    nd = np.array([O.get_dtau_ndoubl(tau[:, l], varpi[:, l], qp, ndoubl_float_type, model.numerics)[1] for l in range(L)], dtype=np.int32)
    Zpp = np.zeros((M, N, N))
    Zmp = np.zeros((M, N, N))
    for m in range(M):
        Zpp[m], Zmp[m] = O.compute_Z_moments(pol, qp.qp_mu.astype(np.float64), model.greek_rayleigh, m)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((n, S))
        F0[0] = 1.0
    F0 = np.ascontiguousarray(np.asarray(F0, dtype=np.float64).T)
    nV = len(model.vza)
    row0 = np.array([n * int(np.argmin(np.abs(qp.qp_mu - O.cosd(v)))) for v in model.vza], dtype=np.int32)
    wgt = np.zeros((M, nV, n))
    for m in range(M):
        weight = 0.5 / math.pi if m == 0 else 1.0 / math.pi
        for v in range(nV):
            c, s = O.cosd(m * model.vaz[v]), O.sind(m * model.vaz[v])
            wgt[m, v] = weight * np.array([c, c, s, s][:n])
    R = np.zeros((S, n, nV))
    T = np.zeros((S, n, nV))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    mu = np.ascontiguousarray(qp.qp_muN, dtype=np.float64)
    wt = np.ascontiguousarray(qp.wt_muN, dtype=np.float64)
    ts = np.ascontiguousarray(tau_sum, dtype=np.float64)
    rc = lib().vsm_oracle_c_rt_run(N, n, dp(mu), dp(wt), int(qp.imu0), float(qp.mu0), S, L, M, dp(tau), dp(varpi), dp(ts), ip(nd),
                                   dp(Zpp), dp(Zmp), dp(F0), float(model.albedo), nV, ip(row0), dp(wgt), int(nthreads), dp(R), dp(T))
    if rc != 0:
        raise RuntimeError("vsm_oracle_c_rt_run failed (rc %d)" % rc)
    return R.transpose(2, 1, 0).copy(), T.transpose(2, 1, 0).copy()
