"""CPU ORACLE (test infrastructure -- NOT the product path).

A numpy restatement of vSmartMOM.jl's `rt_run` CoreRT hot path: quadrature,
Z-moments, layer-optics mixing, elemental -> doubling -> interaction, the
Lambertian surface and the VZA post-processing.  Every function cites the
reference file:line it follows (paths relative to the reference repo).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module.  The shipped package (vsmartmom.jl_amd) never does.

PARITY PIN: the reference is Julia and cannot run here, so this oracle is
pinned by the reference's committed known-answer tables (tests/golden/*.json,
extracted by tests/golden/make_fixtures.py): Siewert-2000 IIA / VLIDORT 2.8.3,
Natraj 2009, 6SV1, VLIDORT solar_tester scalar + vector.  See
tests/test_oracle_golden.py.  Linearized (Jacobian) outputs have no committed
reference numbers => "parity unpinned" for those; they are checked by finite
differences of this oracle only (same as the reference's own tests do).

Array conventions here (numpy, batch-first):  matrices are A[s, i, j]
(spectral point s, row i = outgoing stream*Stokes, column j = incoming),
vectors are v[s, i].  The reference's Julia layout is A[i, j, s] column-major;
`to_ref_layout` / `from_ref_layout` convert at the C-ABI boundary.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------


def eps(FT):
    return np.finfo(FT).eps


def _exact_trig_deg(x, fn):
    """Julia cosd/sind: exact at multiples of 30/45/90 degrees."""
    x = float(x)
    r = math.fmod(x, 360.0)
    if r < 0:
        r += 360.0
    table_cos = {0.0: 1.0, 60.0: 0.5, 90.0: 0.0, 120.0: -0.5, 180.0: -1.0, 240.0: -0.5, 270.0: 0.0, 300.0: 0.5}
    table_sin = {0.0: 0.0, 30.0: 0.5, 90.0: 1.0, 150.0: 0.5, 180.0: 0.0, 210.0: -0.5, 270.0: -1.0, 330.0: -0.5}
    if fn == "cos":
        return table_cos.get(r, math.cos(math.radians(r)))
    return table_sin.get(r, math.sin(math.radians(r)))


def cosd(x):
    return _exact_trig_deg(x, "cos")


def sind(x):
    return _exact_trig_deg(x, "sin")


def to_ref_layout(A):
    """[S,N,M] math-order -> flat buffer in the reference's [N,M,S] column-major order."""
    A = np.asarray(A)
    if A.ndim == 2:  # vector [S,N] -> [N,1,S]
        return np.ascontiguousarray(A).reshape(-1)
    return np.ascontiguousarray(A.transpose(0, 2, 1)).reshape(-1)


def from_ref_layout(buf, S, N, M=None):
    buf = np.asarray(buf)
    if M is None:
        return buf.reshape(S, N).copy()
    return buf.reshape(S, M, N).transpose(0, 2, 1).copy()


# ----------------------------------------------------------------------------
# polarization types  (src/Scattering/types.jl:139-197)
# ----------------------------------------------------------------------------


@dataclass
class Polarization:
    name: str
    n: int
    D: np.ndarray
    I0: np.ndarray


def polarization(name: str) -> Polarization:
    name = name.replace("Stokes_", "").replace("()", "")
    table = {
        "I": (1, [1.0], [1.0]),
        "IQ": (2, [1.0, 1.0], [1.0, 0.0]),
        "IQU": (3, [1.0, 1.0, -1.0], [1.0, 0.0, 0.0]),
        "IQUV": (4, [1.0, 1.0, -1.0, -1.0], [1.0, 0.0, 0.0, 0.0]),
    }
    n, D, I0 = table[name]
    return Polarization(name, n, np.array(D), np.array(I0))


# ----------------------------------------------------------------------------
# quadrature  (src/Scattering/mie_helper_functions.jl:330-335,
#              src/CoreRT/tools/rt_set_streams.jl:25-47)
# ----------------------------------------------------------------------------


def gauleg(n, xmin, xmax):
    xi, w = np.polynomial.legendre.leggauss(n)
    xi = (xmax - xmin) / 2 * xi + (xmin + xmax) / 2
    w = w * (xmax - xmin) / 2
    return xi, w


@dataclass
class QuadPoints:
    mu0: float
    imu0: int          # 0-based index of the SZA node
    qp_mu: np.ndarray  # [Nquad]
    wt_mu: np.ndarray
    qp_muN: np.ndarray  # [N] each node repeated nStokes times
    wt_muN: np.ndarray
    Nquad: int
    Nstreams: int


def rt_set_streams_gausslegquad(Ltrunc: int, sza_deg: float, vza_deg: Sequence[float], pol: Polarization,
                                FT=np.float64) -> QuadPoints:
    """rt_set_streams.jl:25-47 (GaussLegQuad)."""
    Nquad = (Ltrunc + 2) // 2
    qp, wt = gauleg(Nquad, 0.0, 1.0)
    mu0 = cosd(sza_deg)
    cand = np.concatenate([qp, [cosd(v) for v in vza_deg], [mu0]]).astype(FT)
    seen, uniq = set(), []
    for v in cand.tolist():  # Julia `unique`: first occurrence kept, order preserved
        if v not in seen:
            seen.add(v)
            uniq.append(v)
    qp_mu = np.array(uniq, dtype=FT)
    wt_mu = np.concatenate([wt.astype(FT), np.zeros(len(qp_mu) - len(wt), dtype=FT)])
    imu0 = int(np.argmin(np.abs(qp_mu - FT(mu0))))
    return QuadPoints(mu0=float(FT(mu0)), imu0=imu0, qp_mu=qp_mu, wt_mu=wt_mu,
                      qp_muN=np.repeat(qp_mu, pol.n), wt_muN=np.repeat(wt_mu, pol.n),
                      Nquad=len(qp_mu), Nstreams=int(np.count_nonzero(wt_mu)))


# ----------------------------------------------------------------------------
# Greek coefficients and Z moments
# ----------------------------------------------------------------------------


@dataclass
class GreekCoefs:
    alpha: np.ndarray
    beta: np.ndarray
    gamma: np.ndarray
    delta: np.ndarray
    epsilon: np.ndarray
    zeta: np.ndarray


def greek_from_dict(d) -> GreekCoefs:
    return GreekCoefs(*(np.asarray(d[k], dtype=np.float64) for k in
                        ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")))


def get_greek_rayleigh(depol: float) -> GreekCoefs:
    """src/Scattering/mie_helper_functions.jl:454-468."""
    dpl_p = (1 - depol) / (1 + depol / 2)
    dpl_r = (1 - 2 * depol) / (1 - depol)
    return GreekCoefs(
        alpha=np.array([0.0, 0.0, 3 * dpl_p]),
        beta=np.array([1.0, 0.0, 0.5 * dpl_p]),
        gamma=np.array([0.0, 0.0, dpl_p * math.sqrt(1.5)]),
        delta=np.array([0.0, dpl_p * dpl_r * 1.5, 0.0]),
        epsilon=np.zeros(3), zeta=np.zeros(3))


def hg_greek(g: float, nmoments: int) -> GreekCoefs:
    """Scalar Henyey-Greenstein moments beta_L=(2L+1) g^L (case_B_solar_tester.jl:30-45)."""
    L = np.arange(nmoments + 1)
    beta = (2 * L + 1) * g ** L.astype(float)
    z = np.zeros(nmoments + 1)
    return GreekCoefs(z.copy(), beta, z.copy(), z.copy(), z.copy(), z.copy())


def compute_associated_legendre_PRT(mu, Lmax):
    """src/Scattering/legendre_functions.jl:24-183.  Returns (P, R, -T_internal), each [len(mu), Lmax, Lmax]."""
    mu = np.asarray(mu, dtype=np.float64)
    nmu = len(mu)
    P = np.zeros((nmu, Lmax, Lmax))
    R = np.zeros((nmu, Lmax, Lmax))
    T = np.zeros((nmu, Lmax, Lmax))
    smu = np.sqrt(1.0 - mu ** 2)
    cmu = mu
    for m in range(Lmax):
        for l in range(m, Lmax):
            im, il = m, l
            if m == 0:
                if l == 0:
                    P[:, il, im] = 1
                elif l == 1:
                    P[:, il, im] = cmu
                elif l == 2:
                    P[:, il, im] = 0.5 * (3.0 * cmu * cmu - 1.0)
                    R[:, il, im] = 0.5 * math.sqrt(1.5) * smu * smu
                else:
                    P[:, il, im] = (P[:, il - 1, im] * (2 * l - 1) * cmu - P[:, il - 2, im] * (l - 1)) / l
                    Y = math.sqrt((l + 1) * (l - 3))
                    X = math.sqrt(l * l - 4)
                    R[:, il, im] = (R[:, il - 1, im] * (2 * l - 1) * cmu - R[:, il - 2, im] * Y) / X
            elif m == 1:
                if l == 1:
                    P[:, il, im] = math.sqrt(0.5) * smu
                elif l == 2:
                    m1 = math.sqrt(1 / 6)
                    cA = 3 * cmu * smu
                    cB = math.sqrt(1.5) * smu
                    P[:, il, im] = m1 * cA
                    R[:, il, im] = -m1 * cmu * cB
                    T[:, il, im] = m1 * cB
                else:
                    m1 = math.sqrt((l - 1) / (l + 1))
                    m2 = m1 * math.sqrt((l - 2) / l)
                    Y = l - 1 + m
                    X = l - m
                    P[:, il, im] = (m1 * P[:, il - 1, im] * (2 * l - 1) * cmu - m2 * P[:, il - 2, im] * Y) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    Y = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, il, im] = (m1 * R[:, il - 1, im] * (2 * l - 1) * cmu - m2 * R[:, il - 2, im] * Y
                                    + m1 * T[:, il - 1, im] * Z) / X
                    T[:, il, im] = (m1 * T[:, il - 1, im] * (2 * l - 1) * cmu - m2 * T[:, il - 2, im] * Y
                                    + m1 * R[:, il - 1, im] * Z) / X
            else:
                if l == m:
                    fact1 = np.ones(nmu)
                    fact2 = np.ones(nmu)
                    sfull = smu
                    shalf = sfull / 2
                    for i in range(1, m + 1):
                        fact1 = fact1 * ((2 * i - 1) * sfull) / math.sqrt(i * (i + m))
                        if i > 2:
                            fact2 = fact2 * shalf * math.sqrt((m + i) / (i - 2))
                        else:
                            fact2 = fact2 * shalf
                    big = smu > 1e-8
                    with np.errstate(divide="ignore", invalid="ignore"):
                        Aii = np.where(big, fact2 * (1.0 + cmu * cmu) / (smu * smu), 0.5 if m == 2 else 0.0)
                        Aij = np.where(big, fact2 * (2 * cmu) / (smu * smu), 0.5 if m == 2 else 0.0)
                    P[:, il, im] = fact1
                    R[:, il, im] = Aii
                    T[:, il, im] = -Aij
                elif l == m + 1:
                    m1 = math.sqrt(1 / (l + m))
                    X = l - m
                    P[:, il, im] = (m1 * P[:, il - 1, im] * (2 * l - 1) * cmu) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, il, im] = (m1 * R[:, il - 1, im] * (2 * l - 1) * cmu + m1 * T[:, il - 1, im] * Z) / X
                    T[:, il, im] = (m1 * T[:, il - 1, im] * (2 * l - 1) * cmu + m1 * R[:, il - 1, im] * Z) / X
                else:
                    m1 = math.sqrt((l - m) / (l + m))
                    m2 = m1 * math.sqrt((l - m - 1) / (l + m - 1))
                    Y = l - 1 + m
                    X = l - m
                    P[:, il, im] = (m1 * P[:, il - 1, im] * (2 * l - 1) * cmu - m2 * P[:, il - 2, im] * Y) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    Y = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, il, im] = (m1 * R[:, il - 1, im] * (2 * l - 1) * cmu - m2 * R[:, il - 2, im] * Y
                                    + m1 * T[:, il - 1, im] * Z) / X
                    T[:, il, im] = (m1 * T[:, il - 1, im] * (2 * l - 1) * cmu - m2 * T[:, il - 2, im] * Y
                                    + m1 * R[:, il - 1, im] * Z) / X
    return P, R, -T


def _construct_Pi(pol: Polarization, P, R, T, l, m):
    """src/Scattering/mie_helper_functions.jl:532-582 (sign_change=false); l, m 0-based here.
    Returns [nmu, B, B]."""
    nmu = P.shape[0]
    n = pol.n
    Pi = np.zeros((nmu, n, n))
    p, r, t = P[:, l, m], R[:, l, m], T[:, l, m]
    Pi[:, 0, 0] = p
    if n >= 2:
        Pi[:, 1, 1] = r
    if n >= 3:
        Pi[:, 1, 2] = -t
        Pi[:, 2, 1] = -t
        Pi[:, 2, 2] = r
    if n == 4:
        Pi[:, 3, 3] = p
    return Pi


def _construct_B(pol: Polarization, g: GreekCoefs, l):
    """src/Scattering/mie_helper_functions.jl:593-615; l 0-based."""
    n = pol.n
    B = np.zeros((n, n))
    B[0, 0] = g.beta[l]
    if n >= 2:
        B[0, 1] = B[1, 0] = g.gamma[l]
        B[1, 1] = g.alpha[l]
    if n >= 3:
        B[2, 2] = g.zeta[l]
    if n == 4:
        B[2, 3] = g.epsilon[l]
        B[3, 2] = -g.epsilon[l]
        B[3, 3] = g.delta[l]
    return B


def compute_Z_moments(pol: Polarization, mu, greek: GreekCoefs, m: int):
    """src/Scattering/compute_Z_matrices.jl:26-110.  Returns (Zpp, Zmp), each [N, N], N = len(mu)*pol.n."""
    mu = np.asarray(mu, dtype=np.float64)
    assert np.all((0 < mu) & (mu <= 1)), "all mu within compute_Z_moments have to be in ]0,1]"
    n = len(mu)
    fact = 0.5 if m == 0 else 1.0
    l_max = len(greek.beta)
    P, R, T = compute_associated_legendre_PRT(mu, l_max)
    Pm, Rm, Tm = compute_associated_legendre_PRT(-mu, l_max)
    Bd = pol.n
    App = np.zeros((n, Bd, n, Bd))
    Amp = np.zeros((n, Bd, n, Bd))
    for l in range(m, l_max):
        B = _construct_B(pol, greek, l)
        Pi = _construct_Pi(pol, P, R, T, l, m)
        Pim = _construct_Pi(pol, Pm, Rm, Tm, l, m)
        left = np.einsum("iab,bc->iac", Pi, B)
        App += np.einsum("iac,jcd->iajd", left, Pi)
        Amp += np.einsum("iac,jcd->iajd", left, Pim)
    Zpp = 2 * fact * App
    Zmp = 2 * fact * Amp
    for a in range(Bd):
        for d in range(Bd):
            if (a <= 1 and d >= 2) or (a >= 2 and d <= 1):
                Zmp[:, a, :, d] = -Zmp[:, a, :, d]
    N = n * Bd
    return Zpp.reshape(N, N), Zmp.reshape(N, N)


# ----------------------------------------------------------------------------
# layer optics  (src/CoreRT/types.jl:1262-1308,
#                src/CoreRT/LayerOpticalProperties/compEffectiveLayerProperties.jl:11-117)
# ----------------------------------------------------------------------------


@dataclass
class AerosolOptics:
    greek: GreekCoefs
    ssa: float           # omega-tilde
    f_trunc: float = 0.0  # f^t


@dataclass
class CoreScatteringOpticalProperties:
    tau: np.ndarray    # [S] or scalar
    varpi: np.ndarray  # [S] or scalar
    Zpp: np.ndarray    # [N,N] or [S,N,N]
    Zmp: np.ndarray


def _mix(x: CoreScatteringOpticalProperties, y: CoreScatteringOpticalProperties):
    """types.jl:1262-1292  (`+` of two scattering property sets)."""
    tau = x.tau + y.tau
    wx = x.tau * x.varpi
    wy = y.tau * y.varpi
    w = wx + wy
    varpi = w / np.where(tau > 0, tau, 1.0)
    if np.all(wx == 0.0):
        return CoreScatteringOpticalProperties(tau, varpi, y.Zpp, y.Zmp)
    if np.all(wy == 0.0):
        return CoreScatteringOpticalProperties(tau, varpi, x.Zpp, x.Zmp)
    w = np.atleast_1d(w)
    fx = (np.atleast_1d(wx) / w).reshape(-1, 1, 1)
    fy = (np.atleast_1d(wy) / w).reshape(-1, 1, 1)
    Zpp = fx * x.Zpp + fy * y.Zpp
    Zmp = fx * x.Zmp + fy * y.Zmp
    return CoreScatteringOpticalProperties(tau, varpi, Zpp, Zmp)


def _add_absorption(x: CoreScatteringOpticalProperties, tau_abs):
    """types.jl:1302-1308."""
    tau = x.tau + tau_abs
    wx = x.tau * x.varpi
    varpi = wx / np.where(tau > 0, tau, 1.0)
    return CoreScatteringOpticalProperties(tau, varpi, x.Zpp, x.Zmp)


def create_aero(tau_aer, ao: AerosolOptics, Zpp, Zmp):
    """compEffectiveLayerProperties.jl:67-72 (delta-M scaling)."""
    tau_mod = (1 - ao.f_trunc * ao.ssa) * tau_aer
    varpi_mod = (1 - ao.f_trunc) * ao.ssa / (1 - ao.f_trunc * ao.ssa)
    return CoreScatteringOpticalProperties(np.float64(tau_mod), np.float64(varpi_mod), Zpp, Zmp)


@dataclass
class Numerics:
    """src/CoreRT/types.jl:713-756."""
    dtau_max_threshold: Optional[float] = None  # default 0.001
    dtau_min_floor: Optional[float] = None      # default 1024*eps(FT)


@dataclass
class RTModel:
    """The subset of the reference's RTModel that the hot path consumes (single band)."""
    pol: Polarization
    quad_points: QuadPoints
    sza: float
    vza: np.ndarray
    vaz: np.ndarray
    tau_rayl: np.ndarray            # [S, L]
    tau_abs: np.ndarray             # [S, L]
    tau_aer: np.ndarray             # [nAer, L]
    aerosol_optics: List[AerosolOptics]
    greek_rayleigh: GreekCoefs
    albedo: float                   # LambertianSurfaceScalar
    m_max: int
    FT: type = np.float64
    varpi_cabannes: float = 1.0
    numerics: Numerics = field(default_factory=Numerics)
    F0: Optional[np.ndarray] = None  # [nStokes, S]; default SolarBeam e1 (solar_beam.jl:115-121)


def construct_core_optical_properties(model: RTModel, m: int) -> List[CoreScatteringOpticalProperties]:
    """compEffectiveLayerProperties.jl:11-65 (single band, noRS)."""
    mu = model.quad_points.qp_mu.astype(np.float64)
    L = model.tau_rayl.shape[1]
    RZpp, RZmp = compute_Z_moments(model.pol, mu, model.greek_rayleigh, m)
    combo = [CoreScatteringOpticalProperties(model.tau_rayl[:, i].astype(np.float64),
                                             np.float64(model.varpi_cabannes), RZpp, RZmp) for i in range(L)]
    for ia, ao in enumerate(model.aerosol_optics):
        AZpp, AZmp = compute_Z_moments(model.pol, mu, ao.greek, m)
        combo = [_mix(combo[i], create_aero(model.tau_aer[ia, i], ao, AZpp, AZmp)) for i in range(L)]
    return [_add_absorption(combo[i], model.tau_abs[:, i].astype(np.float64)) for i in range(L)]


def get_scattering_interface(prev: str, scatter: bool, iz: int) -> str:
    """src/CoreRT/tools/rt_helper_functions.jl:15-33.  iz is 1-based."""
    if iz == 1:
        return "11" if scatter else "00"
    if prev == "00":
        return "01" if scatter else "00"
    return "11" if scatter else "10"


def extract_effective_props(lods: List[CoreScatteringOpticalProperties], FT):
    """compEffectiveLayerProperties.jl:75-93."""
    S = len(np.atleast_1d(lods[0].tau))
    L = len(lods)
    iface = "00"
    ifaces = []
    tau_sum = np.zeros((S, L + 1), dtype=np.float64)
    for iz in range(L):
        scatter = np.max(lods[iz].tau * lods[iz].varpi) > 2 * eps(FT)
        iface = get_scattering_interface(iface, bool(scatter), iz + 1)
        ifaces.append(iface)
        tau_sum[:, iz + 1] = tau_sum[:, iz] + lods[iz].tau
    return ifaces, tau_sum


def expand_optical_properties(p: CoreScatteringOpticalProperties, FT) -> CoreScatteringOpticalProperties:
    """compEffectiveLayerProperties.jl:106-117."""
    tau = np.atleast_1d(p.tau).astype(FT)
    S = len(tau)
    varpi = np.broadcast_to(np.asarray(p.varpi, dtype=FT), (S,)).copy()
    Zpp, Zmp = np.asarray(p.Zpp), np.asarray(p.Zmp)
    if Zpp.ndim == 2:
        Zpp = np.broadcast_to(Zpp, (S,) + Zpp.shape)
        Zmp = np.broadcast_to(Zmp, (S,) + Zmp.shape)
    return CoreScatteringOpticalProperties(tau, varpi, Zpp.astype(FT), Zmp.astype(FT))


# ----------------------------------------------------------------------------
# hot path: doubling number, elemental, doubling, interaction
# ----------------------------------------------------------------------------


def doubling_number(dtau_max, tau_end, FT):
    """src/CoreRT/tools/rt_helper_functions.jl:49-69."""
    dtau_max = FT(dtau_max)
    tau_end = FT(tau_end)
    if tau_end <= dtau_max:
        return tau_end, 0
    q1 = np.log10(FT(2))
    q2 = np.log10(dtau_max)
    q3 = np.log10(tau_end)
    tlimit = FT((q3 - q2) / q1)
    nlimit = int(math.floor(tlimit))
    diff = FT(tlimit - FT(nlimit))
    if diff < eps(FT):
        return dtau_max, nlimit
    ndoubl = nlimit + 1
    x = q3 - q1 * FT(ndoubl)
    return FT(10) ** x, ndoubl


def get_dtau_ndoubl(tau, varpi, qp: QuadPoints, FT, numerics: Numerics = Numerics()):
    """src/CoreRT/CoreKernel/rt_kernel.jl:266-287."""
    threshold = FT(0.001 if numerics.dtau_max_threshold is None else numerics.dtau_max_threshold)
    floor_val = FT(1024 * eps(FT) if numerics.dtau_min_floor is None else numerics.dtau_min_floor)
    real = qp.qp_mu[qp.wt_mu > eps(FT)]
    mu_min = FT(np.min(qp.qp_mu) if len(real) == 0 else np.min(real))
    tw = FT(np.max(tau.astype(FT) * varpi.astype(FT)))
    dtau_max = max(floor_val, min(tw, FT(threshold * mu_min)))
    _, ndoubl = doubling_number(dtau_max, tw, FT)
    dtau = (tau / FT(2 ** ndoubl)).astype(FT)
    return dtau, ndoubl


def expdiff_neg(a, b):
    """src/CoreRT/CoreKernel/rt_helpers.jl:32-40, vectorized."""
    a, b = np.broadcast_arrays(a, b)
    lo = np.exp(-a) * (-np.expm1(-(b - a)))
    hi = -np.exp(-b) * (-np.expm1(-(a - b)))
    out = np.where(a < b, lo, hi)
    return np.where(a == b, np.zeros_like(out), out)


def _dsign(pol: Polarization, N):
    """D_i = +1 if mod1(i,n) <= 2 else -1."""
    s = np.arange(N) % pol.n
    return np.where(s <= 1, 1.0, -1.0)


@dataclass
class AddedLayer:
    """src/CoreRT/types.jl:155-230 (fields used by the elastic path)."""
    r_mp: np.ndarray  # r^{-+}  [S,N,N]
    t_pp: np.ndarray  # t^{++}
    r_pm: np.ndarray  # r^{+-}
    t_mm: np.ndarray  # t^{--}
    j0_p: np.ndarray  # j0^{+}  [S,N]
    j0_m: np.ndarray  # j0^{-}


@dataclass
class CompositeLayer:
    R_mp: np.ndarray
    R_pm: np.ndarray
    T_pp: np.ndarray
    T_mm: np.ndarray
    J0_p: np.ndarray
    J0_m: np.ndarray


def make_added_layer(FT, N, S) -> AddedLayer:
    z = lambda: np.zeros((S, N, N), dtype=FT)
    v = lambda: np.zeros((S, N), dtype=FT)
    return AddedLayer(z(), z(), z(), z(), v(), v())


def make_composite_layer(FT, N, S) -> CompositeLayer:
    z = lambda: np.zeros((S, N, N), dtype=FT)
    v = lambda: np.zeros((S, N), dtype=FT)
    return CompositeLayer(z(), z(), z(), z(), v(), v())


def elemental(pol: Polarization, tau_sum, dtau, F0, varpi, Zpp, Zmp, m: int, ndoubl: int,
              qp: QuadPoints, added: AddedLayer, FT):
    """src/CoreRT/CoreKernel/elemental.jl:174-230 with kernels :289-334 (get_elem_rt!),
    :348-392 (get_elem_rt_SFI!), :403-422 (apply_D_elemental!)."""
    mu = qp.qp_muN.astype(FT)
    N = len(mu)
    S = len(dtau)
    wct02 = FT(0.5) if m == 0 else FT(0.25)
    wct = (qp.wt_muN.astype(FT) / FT(2)) if m == 0 else (qp.wt_muN.astype(FT) / FT(4))
    mi = mu[None, :, None]
    mj = mu[None, None, :]
    wj = wct[None, None, :]
    d = dtau.astype(FT)[:, None, None]
    w = varpi.astype(FT)[:, None, None]
    Zpp = np.broadcast_to(np.asarray(Zpp, dtype=FT), (S, N, N))
    Zmp = np.broadcast_to(np.asarray(Zmp, dtype=FT), (S, N, N))
    one = FT(1)
    active = (wct > eps(FT))[None, None, :]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        r = w * Zmp * (mj / (mi + mj)) * wj * (-np.expm1(-d * ((one / mi) + (one / mj))))
        t_off = w * Zpp * (mj / (mi - mj)) * wj * expdiff_neg(d / mi, d / mj)
        eye = np.eye(N, dtype=bool)[None]
        t_diag = np.exp(-d / mi) * (one + w * Zpp * (d / mi) * wj)          # i == j (w_j == w_i there)
        t_same = np.exp(-d / mj) * (w * Zpp * (d / mi) * wj)                 # mu_i == mu_j, i != j
        same_mu = (mi == mj)
        t = np.where(same_mu, np.where(eye, t_diag, t_same), t_off)
        t_inact = np.where(eye, np.exp(-d / mi) * np.ones((1, 1, N), dtype=FT), FT(0))
    r = np.where(active, r, FT(0)).astype(FT)
    t = np.where(active, t, t_inact).astype(FT)

    # --- SFI source (get_elem_rt_SFI!) ---
    n = pol.n
    i_start = n * qp.imu0
    i_end = n * (qp.imu0 + 1)  # exclusive
    F0 = np.asarray(F0, dtype=FT)  # [n, S]
    ZF_p = np.einsum("sik,ks->si", Zpp[:, :, i_start:i_end], F0)
    ZF_m = np.einsum("sik,ks->si", Zmp[:, :, i_start:i_end], F0)
    mu_s = mu[i_start]
    mi1 = mu[None, :]
    d1 = dtau.astype(FT)[:, None]
    w1 = varpi.astype(FT)[:, None]
    in_sun = np.zeros(N, dtype=bool)
    in_sun[i_start:i_end] = True
    with np.errstate(divide="ignore", invalid="ignore"):
        jp_sun = wct02 * w1 * ZF_p * (d1 / mi1) * np.exp(-d1 / mi1)
        jp_off = wct02 * w1 * ZF_p * (mu_s / (mi1 - mu_s)) * expdiff_neg(d1 / mi1, d1 / mu_s)
        jp = np.where(in_sun[None, :], jp_sun, jp_off)
        jm = wct02 * w1 * ZF_m * (mu_s / (mi1 + mu_s)) * (-np.expm1(-d1 * ((one / mi1) + (one / mu_s))))
    att = np.exp(-np.asarray(tau_sum, dtype=FT) / mu_s)[:, None]
    jp = (jp * att).astype(FT)
    jm = (jm * att).astype(FT)
    D = _dsign(pol, N).astype(FT)
    if ndoubl >= 1:
        jm = jm * D[None, :]
    added.j0_p[...] = jp
    added.j0_m[...] = jm

    # --- apply_D_elemental! ---
    if ndoubl < 1:
        par = (D[:, None] * D[None, :])[None]
        added.r_mp[...] = r
        added.t_pp[...] = t
        added.r_pm[...] = par * r
        added.t_mm[...] = par * t
    else:
        added.r_mp[...] = r * D[None, :, None]
        added.t_pp[...] = t


def batch_inv(A):
    """cpu_batched.jl:32-47 (A \\ I per spectral slice; LAPACK getrf/getrs = partial pivoting)."""
    return np.linalg.inv(A)


def doubling(pol: Polarization, expk, ndoubl: int, added: AddedLayer, FT):
    """src/CoreRT/CoreKernel/doubling.jl:38-99 + rt_helpers.jl:102-166; apply_D doubling.jl:178-252."""
    if ndoubl == 0:
        return
    r, t = added.r_mp, added.t_pp
    jp, jm = added.j0_p, added.j0_m
    N = r.shape[1]
    I = np.eye(N, dtype=FT)[None]
    expk = expk.astype(FT).copy()
    for _ in range(ndoubl):
        G = batch_inv(I - r @ r).astype(FT)
        tt = t @ G
        j1p = jp * expk[:, None]
        j1m = jm * expk[:, None]
        jm_new = jm + np.einsum("sij,sj->si", tt, j1m + np.einsum("sij,sj->si", r, jp))
        jp_new = j1p + np.einsum("sij,sj->si", tt, jp + np.einsum("sij,sj->si", r, j1m))
        jm, jp = jm_new.astype(FT), jp_new.astype(FT)
        r_new = r + tt @ r @ t
        t = (tt @ t).astype(FT)
        r = r_new.astype(FT)
        expk = expk ** 2
    D = _dsign(pol, N).astype(FT)
    if pol.n == 1:
        added.r_mp[...] = r
        added.t_pp[...] = t
        added.r_pm[...] = r
        added.t_mm[...] = t
        added.j0_p[...] = jp
        added.j0_m[...] = jm
        return
    r = r * D[None, :, None]
    par = (D[:, None] * D[None, :])[None]
    added.r_mp[...] = r
    added.t_pp[...] = t
    added.r_pm[...] = par * r
    added.t_mm[...] = par * t
    added.j0_p[...] = jp
    added.j0_m[...] = jm * D[None, :]


def _mv(A, v):
    return np.einsum("sij,sj->si", A, v)


def interaction(iface: str, comp: CompositeLayer, add: AddedLayer, FT):
    """src/CoreRT/CoreKernel/interaction.jl:52-266 (statement order preserved)."""
    r_mp, r_pm, t_pp, t_mm, j0_p, j0_m = add.r_mp, add.r_pm, add.t_pp, add.t_mm, add.j0_p, add.j0_m
    N = r_mp.shape[1]
    I = np.eye(N, dtype=FT)[None]
    c = comp
    if iface == "00":
        c.J0_p[...] = j0_p + _mv(t_pp, c.J0_p)
        c.J0_m[...] = c.J0_m + _mv(c.T_mm, j0_m)
        c.T_mm[...] = t_mm @ c.T_mm
        c.T_pp[...] = t_pp @ c.T_pp
    elif iface == "01":
        c.J0_m[...] = c.J0_m + _mv(c.T_mm, _mv(r_mp, c.J0_p) + j0_m)
        c.J0_p[...] = j0_p + _mv(t_pp, c.J0_p)
        c.R_mp[...] = c.T_mm @ r_mp @ c.T_pp
        c.R_pm[...] = r_pm
        c.T_pp[...] = t_pp @ c.T_pp
        c.T_mm[...] = c.T_mm @ t_mm
    elif iface == "10":
        c.J0_p[...] = j0_p + _mv(t_pp, c.J0_p + _mv(c.R_pm, j0_m))
        c.J0_m[...] = c.J0_m + _mv(c.T_mm, j0_m)
        c.T_pp[...] = t_pp @ c.T_pp
        c.T_mm[...] = c.T_mm @ t_mm
        c.R_pm[...] = t_pp @ c.R_pm @ t_mm
    elif iface == "11":
        G1 = batch_inv(I - r_mp @ c.R_pm).astype(FT)
        T01_inv = c.T_mm @ G1
        c.J0_m[...] = c.J0_m + _mv(T01_inv, _mv(r_mp, c.J0_p) + j0_m)
        c.R_mp[...] = c.R_mp + T01_inv @ r_mp @ c.T_pp
        c.T_mm[...] = T01_inv @ t_mm
        G2 = batch_inv(I - c.R_pm @ r_mp).astype(FT)
        T21_inv = t_pp @ G2
        c.J0_p[...] = j0_p + _mv(T21_inv, c.J0_p + _mv(c.R_pm, j0_m))
        c.T_pp[...] = T21_inv @ c.T_pp
        c.R_pm[...] = r_pm + T21_inv @ c.R_pm @ t_mm
    else:
        raise ValueError(iface)


def zero_added_noscat(added: AddedLayer, tau, qp: QuadPoints, FT):
    """rt_helpers.jl:174-180 + rt_kernel.jl:36-45.  NOTE: j0^+ is deliberately NOT zeroed (reference quirk)."""
    added.r_mp[...] = 0
    added.r_pm[...] = 0
    added.j0_m[...] = 0
    N = added.t_pp.shape[1]
    e = np.exp(-tau.astype(FT)[:, None] / qp.qp_muN.astype(FT)[None, :])
    added.t_pp[...] = 0
    added.t_mm[...] = 0
    idx = np.arange(N)
    added.t_pp[:, idx, idx] = e
    added.t_mm[:, idx, idx] = e


def copy_added_to_composite(comp: CompositeLayer, add: AddedLayer):
    """rt_helpers.jl:188-200."""
    comp.T_pp[...] = add.t_pp
    comp.T_mm[...] = add.t_mm
    comp.R_mp[...] = add.r_mp
    comp.R_pm[...] = add.r_pm
    comp.J0_p[...] = add.j0_p
    comp.J0_m[...] = add.j0_m


def rt_kernel(pol, added, comp, props: CoreScatteringOpticalProperties, iface, tau_sum, m, qp, iz, F0, FT,
              numerics: Numerics = Numerics(), trace=None):
    """src/CoreRT/CoreKernel/rt_kernel.jl:175-250 (noRS).  iz is 1-based."""
    tau, varpi = props.tau, props.varpi
    scatter = np.max(tau * varpi) > 2 * eps(FT)
    ndoubl = 0
    if scatter:
        dtau, ndoubl = get_dtau_ndoubl(tau, varpi, qp, FT, numerics)
        expk = np.exp(-dtau / FT(qp.mu0)).astype(FT)
        elemental(pol, tau_sum, dtau, F0, varpi, props.Zpp, props.Zmp, m, ndoubl, qp, added, FT)
        doubling(pol, expk, ndoubl, added, FT)
    else:
        zero_added_noscat(added, tau, qp, FT)
    if trace is not None:
        trace.append(dict(iz=iz, m=m, scatter=bool(scatter), ndoubl=ndoubl, iface=iface))
    if iz == 1:
        copy_added_to_composite(comp, added)
    else:
        interaction(iface, comp, added, FT)


# ----------------------------------------------------------------------------
# surface + postprocessing + driver
# ----------------------------------------------------------------------------


def create_surface_layer_lambertian(albedo, added: AddedLayer, m, pol, qp: QuadPoints, tau_sum, FT):
    """src/CoreRT/Surfaces/lambertian_surface.jl:41-95 (LambertianSurfaceScalar, forward)."""
    N = added.r_mp.shape[1]
    S = added.r_mp.shape[0]
    n = pol.n
    Nquad = N // n
    I = np.eye(N, dtype=FT)
    if m == 0:
        rho = FT(2) * FT(albedo)
        R_surf = np.zeros((N, N), dtype=FT)
        R_surf[0::n, 0::n] = rho  # diag(rho,0,..) tiled over all stream pairs
        I0N = np.zeros(N, dtype=FT)
        i_start = n * qp.imu0
        I0N[i_start:i_start + n] = pol.I0
        att = np.exp(-np.asarray(tau_sum, dtype=FT) / FT(qp.mu0))
        added.j0_p[...] = I0N[None, :] * att[:, None]
        added.j0_m[...] = (FT(qp.mu0) * (R_surf @ I0N))[None, :] * att[:, None]
        R_surf = R_surf * (qp.qp_muN.astype(FT) * qp.wt_muN.astype(FT))[None, :]
        added.r_mp[...] = R_surf[None]
        added.r_pm[...] = 0
        added.t_pp[...] = I[None]
        added.t_mm[...] = I[None]
    else:
        added.r_mp[...] = 0  # (reference zeroes r^{-+} twice and never r^{+-}; r^{+-} is 0 from m=0)
        added.t_pp[...] = I[None]
        added.t_mm[...] = I[None]
        added.j0_p[...] = 0
        added.j0_m[...] = 0


def legendre_albedo(legendre_coeff, nSpec, FT=np.float64):
    """lambertian_surface.jl:113-117: albedo = P(x) coeff on x = range(-1, 1, length = nSpec) with P from
    Scattering.compute_legendre_poly (legendre_functions.jl:223-252, the P0 column)."""
    c = np.asarray(legendre_coeff, dtype=FT)
    assert len(c) > 1
    x = np.linspace(FT(-1), FT(1), nSpec).astype(FT)
    P = np.zeros((nSpec, len(c)), dtype=FT)
    P[:, 0] = 1
    P[:, 1] = x
    for n in range(2, len(c)):
        l = n - 1
        P[:, n] = ((2 * l + 1) * x * P[:, n - 1] - l * P[:, n - 2]) / (l + 1)
    return P @ c


def create_surface_layer_lambertian_spectral(albedo, added: AddedLayer, m, pol, qp: QuadPoints, tau_sum, FT):
    """src/CoreRT/Surfaces/lambertian_surface.jl:97-213 (LambertianSurfaceLegendre / LambertianSurfaceSpline): `albedo` [S].
    Reference quirks kept: j0+ = 0; for m > 0 also t++ = t-- = 0."""
    S, N = added.j0_p.shape
    n = pol.n
    if m == 0:
        rho = FT(2) * np.asarray(albedo, dtype=FT)
        R_surf = np.zeros((N, N), dtype=FT)
        R_surf[0::n, 0::n] = 1
        I0N = np.zeros(N, dtype=FT)
        I0N[n * qp.imu0:n * qp.imu0 + n] = pol.I0
        att = np.exp(-np.asarray(tau_sum, dtype=FT) / FT(qp.mu0))
        added.j0_p[...] = 0
        added.j0_m[...] = (FT(qp.mu0) * (R_surf @ I0N))[None, :] * (rho * att)[:, None]
        Rw = R_surf * (qp.qp_muN.astype(FT) * qp.wt_muN.astype(FT))[None, :]
        added.r_mp[...] = rho[:, None, None] * Rw[None]
        added.r_pm[...] = 0
        added.t_pp[...] = np.eye(N, dtype=FT)[None]
        added.t_mm[...] = np.eye(N, dtype=FT)[None]
    else:
        for arr in (added.r_mp, added.t_pp, added.t_mm, added.j0_p, added.j0_m):
            arr[...] = 0


def postprocessing_vza(pol, comp: CompositeLayer, vza, vaz, qp: QuadPoints, m, weight, R_SFI, T_SFI):
    """src/CoreRT/tools/postprocessing_vza.jl:23-94 (noRS, SFI branch).  R_SFI/T_SFI: [nVZA, nStokes, S]."""
    n = pol.n
    for i in range(len(vza)):
        imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(cosd(vza[i])))))
        istart = imu * n
        c, s = cosd(m * vaz[i]), sind(m * vaz[i])
        w = weight * np.array([c, c, s, s][:n])
        R_SFI[i] += w[:, None] * comp.J0_m[:, istart:istart + n].T
        T_SFI[i] += w[:, None] * comp.J0_p[:, istart:istart + n].T


def interaction_hdrf(comp: CompositeLayer, added_surf: AddedLayer, m, pol, qp: QuadPoints, bhr_uw, bhr_dw):
    """src/CoreRT/CoreKernel/interaction_hdrf.jl:4-42.  Returns hdr_J0- [S, N]; for m == 0 fills bhr_uw / bhr_dw [n, S]."""
    N = comp.J0_p.shape[1]
    n = pol.n
    hdr_J = np.einsum("sij,sj->si", added_surf.r_mp, comp.J0_p) + added_surf.j0_m
    if m == 0:
        wmu = qp.wt_muN * qp.qp_muN
        i0 = n * qp.imu0
        for i in range(n):
            j = slice(i, N, n)
            bhr_uw[i] = np.sum(hdr_J[:, j] * wmu[j][None, :], axis=1)
            bhr_dw[i] = np.sum(comp.J0_p[:, j] * wmu[j][None, :], axis=1) + added_surf.j0_p[:, i0] * qp.qp_muN[i0]
    return hdr_J


def rt_run(model: RTModel, trace=None, per_m=None, hdrf=None):
    """src/CoreRT/rt_run.jl:238-539 (noRS, SFI=true, Lambertian surface).  Returns (R_SFI, T_SFI); with `hdrf` = a dict it is
    filled with hdr [nVZA, nStokes, S], bhr_uw, bhr_dw [nStokes, S] (the reference returns bhr_uw[1,:], bhr_dw[1,:])."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    R_SFI = np.zeros((nV, pol.n, S), dtype=FT)
    T_SFI = np.zeros((nV, pol.n, S), dtype=FT)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S), dtype=FT)
        F0[0, :] = 1
    added = make_added_layer(FT, N, S)
    added_surf = make_added_layer(FT, N, S)
    comp = make_composite_layer(FT, N, S)
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = construct_core_optical_properties(model, m)
        ifaces, tau_sum_all = extract_effective_props(lods, FT)
        for iz in range(L):
            lo = expand_optical_properties(lods[iz], FT)
            rt_kernel(pol, added, comp, lo, ifaces[iz], tau_sum_all[:, iz].astype(FT), m, qp, iz + 1, F0, FT,
                      model.numerics, trace)
        if np.ndim(model.albedo) == 1:     # spectrally varying Lambertian albedo (Legendre / spline surfaces)
            create_surface_layer_lambertian_spectral(model.albedo, added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        else:
            create_surface_layer_lambertian(model.albedo, added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        interaction(ifaces[-1], comp, added_surf, FT)
        if per_m is not None:
            per_m.append(dict(m=m, J0_m=comp.J0_m.copy(), J0_p=comp.J0_p.copy(), weight=weight))
        if hdrf is not None:
            if m == 0:
                hdrf.update(hdr=np.zeros((nV, pol.n, S), dtype=FT), bhr_uw=np.zeros((pol.n, S), dtype=FT),
                            bhr_dw=np.zeros((pol.n, S), dtype=FT))
            hj = interaction_hdrf(comp, added_surf, m, pol, qp, hdrf["bhr_uw"], hdrf["bhr_dw"])
            hc = CompositeLayer(None, None, None, None, np.zeros_like(hj), hj)        # postprocessing_vza_hdrf!: the J0- leg only
            postprocessing_vza(pol, hc, model.vza, model.vaz, qp, m, weight, hdrf["hdr"], np.zeros_like(hdrf["hdr"]))
        postprocessing_vza(pol, comp, model.vza, model.vaz, qp, m, weight, R_SFI, T_SFI)
    return R_SFI, T_SFI


# ----------------------------------------------------------------------------
# thermal emission: the per-source slot `:thermal` of the source-term framework
# ----------------------------------------------------------------------------


def planck_spectrum_wn(T, nu):
    """src/SolarModel/SolarModel.jl:26-35: Planck radiance in mW / (m^2 sr cm^-1) on a wavenumber grid [cm^-1]."""
    nu = np.asarray(nu, dtype=np.float64)
    return 1.1910427e-5 * nu ** 3 / (np.exp(1.4387752 * nu / T) - 1.0)


def thermal_source(pol: Polarization, qp: QuadPoints, dtau, varpi, B, FT):
    """contribute!(::PreparedThermalEmission, ...) (src/CoreRT/Sources/thermal_emission.jl:241-301): the elemental layer's
    thermal source, identical up and down, on the Stokes-I rows:  2 pi (1 - varpi) B (1 - exp(-dtau / mu_i)).  [S, N]"""
    mu = qp.qp_mu.astype(FT)
    n = pol.n
    S = len(dtau)
    j = np.zeros((S, len(mu) * n), dtype=FT)
    coeff = FT(2 * math.pi) * (FT(1) - np.asarray(varpi, dtype=FT)) * np.asarray(B, dtype=FT)
    for i, mi in enumerate(mu):
        if mi > eps(FT):
            j[:, i * n] = coeff * (-np.expm1(-np.asarray(dtau, dtype=FT) / mi))
    return j


def rt_run_thermal(model: RTModel, B_layer, reset_slot_in_nonscattering_layers: bool = False):
    """The `:thermal` per-source slot of rt_run (rt_kernel.jl:204-232 slot reset + contribute!, doubling.jl:62-81 per-source
    source update with the slot's own expk = 1, interaction.jl:52-75 ... per-source J0 recurrences, postprocessing_vza.jl:68-82):
    the same linear source recurrences as the solar slot, driven by the thermal source.  Returns the slot's contribution
    (R_th, T_th) -- rt_run adds it to R_SFI / T_SFI.  B_layer: [L, S] Planck radiance per layer.  The slot recurrences are
    evaluated on a copy of the layer (its r, t evolve exactly like the solar pass's).

    State the reference carries, restated as written (rt_kernel.jl:204-232): the slot's j0+- live in the AddedLayer, which is
    allocated ONCE per run (rt_run.jl:326-335).  The scatter branch resets them (:217-221), lets contribute! fill them (m = 0 only:
    thermal emission is isotropic) and doubles them; the non-scattering branch (:229-232, zero_added_noscat!) does not touch
    them, so a non-scattering layer interacts with the slot of the LAST SCATTERING LAYER BEFORE IT -- and when the column
    begins with non-scattering layers, moment m = 1 starts on the doubled thermal slot that the last scattering layer of moment
    m = 0 left behind (every later moment starts on zeros: the scattering layers of m >= 1 reset the slot and contribute nothing).
    The Fourier loop below walks all moments the way the reference does and skips those in which the slot provably stays zero.
    `reset_slot_in_nonscattering_layers=True` is the corrected variant (the slot of a non-scattering layer is zero: such a layer
    emits nothing here, like the reference's, and carries nothing over) -- an explicit option, not the default."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    R_th = np.zeros((nV, pol.n, S), dtype=FT)
    T_th = np.zeros((nV, pol.n, S), dtype=FT)
    B_layer = np.asarray(B_layer, dtype=FT)
    F0 = np.zeros((pol.n, S), dtype=FT)
    added, added_surf = make_added_layer(FT, N, S), make_added_layer(FT, N, S)
    slot_p, slot_m = np.zeros((S, N), dtype=FT), np.zeros((S, N), dtype=FT)     # the AddedLayer's slot: persists over (m, layer)
    for m in range(model.m_max + 1):
        if m > 0 and not (np.any(slot_p != 0) or np.any(slot_m != 0)):
            continue            # nothing is ever contributed at m > 0: with a zero slot on entry the whole moment stays zero
        comp = make_composite_layer(FT, N, S)
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = construct_core_optical_properties(model, m)
        ifaces, tau_sum_all = extract_effective_props(lods, FT)
        for iz in range(L):
            lo = expand_optical_properties(lods[iz], FT)
            tau, varpi = lo.tau, lo.varpi
            if np.max(tau * varpi) > 2 * eps(FT):
                dtau, nd = get_dtau_ndoubl(tau, varpi, qp, FT, model.numerics)
                elemental(pol, tau_sum_all[:, iz].astype(FT), dtau, F0, varpi, lo.Zpp, lo.Zmp, m, nd, qp, added, FT)
                j = (thermal_source(pol, qp, dtau, varpi, B_layer[iz], FT) if (m == 0 and iz < B_layer.shape[0])
                     else np.zeros((S, N), dtype=FT))                            # reset (:217-221) + contribute! (m = 0 only)
                added.j0_p[...] = j
                added.j0_m[...] = j
                doubling(pol, np.ones(S, dtype=FT), nd, added, FT)
                slot_p[...], slot_m[...] = added.j0_p, added.j0_m
            else:
                zero_added_noscat(added, tau, qp, FT)
                if reset_slot_in_nonscattering_layers:
                    slot_p[...] = 0
                    slot_m[...] = 0
                added.j0_p[...], added.j0_m[...] = slot_p, slot_m                # the slot as the last scattering layer left it
            if iz == 0:
                copy_added_to_composite(comp, added)
            else:
                interaction(ifaces[iz], comp, added, FT)
        if np.ndim(model.albedo) == 1:
            create_surface_layer_lambertian_spectral(model.albedo, added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        else:
            create_surface_layer_lambertian(model.albedo, added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        added_surf.j0_p[...] = 0          # no solar beam in this slot (surface emission is a separate source type)
        added_surf.j0_m[...] = 0
        interaction(ifaces[-1], comp, added_surf, FT)
        postprocessing_vza(pol, comp, model.vza, model.vaz, qp, m, weight, R_th, T_th)
    return R_th, T_th


# ----------------------------------------------------------------------------
# model builders used by the golden tests and the synthetic benchmarks
# ----------------------------------------------------------------------------


def m_max_from_components(Nstreams: int, max_m_count: int, l_max: int, greeks: Sequence[GreekCoefs],
                          lambertian: bool = True) -> int:
    """src/CoreRT/tools/model_from_parameters.jl:109-141 + component_m_max.jl:72-131 (trait aggregator):
    m_max = min(max(2 [Rayleigh], len(beta)-1 [aerosols], 0 [Lambertian, SolarBeam]), user_l_cap)."""
    user_l_cap = min(max(2 * Nstreams - 1, 0), max(max_m_count - 1, 0), l_max)
    mm = 2
    for g in greeks:
        mm = max(mm, len(g.beta) - 1)
    return max(min(mm, user_l_cap), 0)


def build_model(pol_name, l_trunc, sza, vza, vaz, tau_rayl, tau_abs=None, tau_aer=None, aerosols=(),
                depol=0.0, albedo=0.0, m_max=None, FT=np.float64, numerics=None) -> RTModel:
    pol = polarization(pol_name)
    qp = rt_set_streams_gausslegquad(l_trunc, sza, vza, pol, FT)
    tau_rayl = np.atleast_2d(np.asarray(tau_rayl, dtype=np.float64))
    S, L = tau_rayl.shape
    tau_abs = np.zeros((S, L)) if tau_abs is None else np.atleast_2d(np.asarray(tau_abs, dtype=np.float64))
    tau_aer = np.zeros((len(aerosols), L)) if tau_aer is None else np.atleast_2d(np.asarray(tau_aer, dtype=np.float64))
    if m_max is None:
        m_max = 2
    return RTModel(pol=pol, quad_points=qp, sza=sza, vza=np.asarray(vza, dtype=float), vaz=np.asarray(vaz, dtype=float),
                   tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer, aerosol_optics=list(aerosols),
                   greek_rayleigh=get_greek_rayleigh(depol), albedo=albedo, m_max=m_max, FT=FT,
                   numerics=numerics or Numerics())
