"""Host-side producers of the hot path's inputs (numpy, FP64): streams, Greek
coefficients, Fourier moments of the phase matrix, layer-optics mixing.

These mirror the reference's host code that runs once per Fourier moment
*outside* the per-spectral-point loop (SURVEY.md 8f rank 1 -- "next" rows):
  src/CoreRT/tools/rt_set_streams.jl:25-47                 rt_set_streams (GaussLegQuad)
  src/Scattering/mie_helper_functions.jl:454-468            get_greek_rayleigh
  src/Scattering/legendre_functions.jl:24-183               generalized spherical functions
  src/Scattering/compute_Z_matrices.jl:26-110               compute_Z_moments
  src/CoreRT/types.jl:1262-1308, LayerOpticalProperties/compEffectiveLayerProperties.jl:11-117
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


# ---- polarization (src/Scattering/types.jl:139-197) ---------------------------
@dataclass(frozen=True)
class PolarizationType:
    name: str
    n: int
    D: tuple
    I0: tuple


def Stokes_I():
    return PolarizationType("Stokes_I", 1, (1.0,), (1.0,))


def Stokes_IQ():
    return PolarizationType("Stokes_IQ", 2, (1.0, 1.0), (1.0, 0.0))


def Stokes_IQU():
    return PolarizationType("Stokes_IQU", 3, (1.0, 1.0, -1.0), (1.0, 0.0, 0.0))


def Stokes_IQUV():
    return PolarizationType("Stokes_IQUV", 4, (1.0, 1.0, -1.0, -1.0), (1.0, 0.0, 0.0, 0.0))


def polarization_type(name: str) -> PolarizationType:
    key = name.replace("()", "").replace("Stokes_", "")
    return {"I": Stokes_I, "IQ": Stokes_IQ, "IQU": Stokes_IQU, "IQUV": Stokes_IQUV}[key]()


# ---- degrees trig with Julia's exact special angles -------------------------------
_COS_EXACT = {0: 1.0, 60: 0.5, 90: 0.0, 120: -0.5, 180: -1.0, 240: -0.5, 270: 0.0, 300: 0.5}
_SIN_EXACT = {0: 0.0, 30: 0.5, 90: 1.0, 150: 0.5, 180: 0.0, 210: -0.5, 270: -1.0, 330: -0.5}


def cosd(x: float) -> float:
    r = math.fmod(float(x), 360.0)
    r = r + 360.0 if r < 0 else r
    return _COS_EXACT[int(r)] if r == int(r) and int(r) in _COS_EXACT else math.cos(math.radians(r))


def sind(x: float) -> float:
    r = math.fmod(float(x), 360.0)
    r = r + 360.0 if r < 0 else r
    return _SIN_EXACT[int(r)] if r == int(r) and int(r) in _SIN_EXACT else math.sin(math.radians(r))


# ---- streams -------------------------------------------------------------------------
@dataclass
class QuadPoints:
    """src/CoreRT/types.jl QuadPoints (host copy; device copies live in CoreRT.DeviceQuad)."""
    mu0: float
    imu0: int            # 0-based index of the SZA node (reference iμ₀ - 1)
    qp_mu: np.ndarray
    wt_mu: np.ndarray
    qp_muN: np.ndarray
    wt_muN: np.ndarray
    Nquad: int
    Nstreams: int


def gauleg(n: int, xmin: float, xmax: float):
    xi, w = np.polynomial.legendre.leggauss(n)
    return (xmax - xmin) / 2 * xi + (xmin + xmax) / 2, w * (xmax - xmin) / 2


def rt_set_streams(l_trunc: int, sza: float, vza: Sequence[float], pol: PolarizationType, FT=np.float64) -> QuadPoints:
    nq = (l_trunc + 2) // 2
    x, w = gauleg(nq, 0.0, 1.0)
    mu0 = cosd(sza)
    nodes = [FT(v) for v in x] + [FT(cosd(v)) for v in vza] + [FT(mu0)]
    uniq = list(dict.fromkeys(float(v) for v in nodes))  # first occurrence kept, order preserved
    qp = np.array(uniq, dtype=FT)
    wt = np.zeros(len(qp), dtype=FT)
    wt[:nq] = w.astype(FT)
    imu0 = int(np.argmin(np.abs(qp - FT(mu0))))
    return QuadPoints(float(FT(mu0)), imu0, qp, wt, np.repeat(qp, pol.n), np.repeat(wt, pol.n), len(qp),
                      int(np.count_nonzero(wt)))


# ---- Greek coefficients -----------------------------------------------------------------
@dataclass
class GreekCoefs:
    alpha: np.ndarray
    beta: np.ndarray
    gamma: np.ndarray
    delta: np.ndarray
    epsilon: np.ndarray
    zeta: np.ndarray

    @staticmethod
    def from_dict(d):
        return GreekCoefs(*(np.asarray(d[k], dtype=np.float64) for k in
                            ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")))


def get_greek_rayleigh(depol: float) -> GreekCoefs:
    p = (1 - depol) / (1 + depol / 2)
    r = (1 - 2 * depol) / (1 - depol)
    z = np.zeros(3)
    return GreekCoefs(np.array([0, 0, 3 * p]), np.array([1, 0, 0.5 * p]), np.array([0, 0, p * math.sqrt(1.5)]),
                      np.array([0, 1.5 * p * r, 0]), z.copy(), z.copy())


def henyey_greenstein_greek(g: float, nmoments: int) -> GreekCoefs:
    L = np.arange(nmoments + 1)
    z = np.zeros(nmoments + 1)
    return GreekCoefs(z.copy(), (2 * L + 1) * g ** L.astype(float), z.copy(), z.copy(), z.copy(), z.copy())


@dataclass
class AerosolOptics:
    greek_coefs: GreekCoefs
    ssa: float          # ω̃
    f_trunc: float = 0.0  # fᵗ


# ---- generalized spherical functions, one Fourier order at a time ---------------------------
def _prt_for_m(x: np.ndarray, lmax: int, m: int):
    """P_l^m, R_l^m, T_l^m (reference sign convention: returns the published `-T` array) for
    l = 0..lmax-1 at fixed m.  Recurrences of legendre_functions.jl:24-183 restricted to one m."""
    n = len(x)
    P = np.zeros((lmax, n))
    R = np.zeros((lmax, n))
    T = np.zeros((lmax, n))
    s = np.sqrt(1.0 - x * x)
    c = x
    for l in range(m, lmax):
        if m == 0:
            if l == 0:
                P[l] = 1.0
            elif l == 1:
                P[l] = c
            elif l == 2:
                P[l] = 0.5 * (3 * c * c - 1)
                R[l] = 0.5 * math.sqrt(1.5) * s * s
            else:
                P[l] = (P[l - 1] * (2 * l - 1) * c - P[l - 2] * (l - 1)) / l
                R[l] = (R[l - 1] * (2 * l - 1) * c - R[l - 2] * math.sqrt((l + 1) * (l - 3))) / math.sqrt(l * l - 4)
            continue
        if l == m and m == 1:
            P[l] = math.sqrt(0.5) * s
            continue
        if m == 1 and l == 2:
            m1 = math.sqrt(1 / 6)
            P[l] = m1 * 3 * c * s
            R[l] = -m1 * c * math.sqrt(1.5) * s
            T[l] = m1 * math.sqrt(1.5) * s
            continue
        if l == m:  # m >= 2
            f1 = np.ones(n)
            f2 = np.ones(n)
            for i in range(1, m + 1):
                f1 = f1 * ((2 * i - 1) * s) / math.sqrt(i * (i + m))
                f2 = f2 * (s / 2) * (math.sqrt((m + i) / (i - 2)) if i > 2 else 1.0)
            ok = s > 1e-8
            lim = 0.5 if m == 2 else 0.0
            with np.errstate(divide="ignore", invalid="ignore"):
                P[l] = f1
                R[l] = np.where(ok, f2 * (1 + c * c) / (s * s), lim)
                T[l] = -np.where(ok, f2 * (2 * c) / (s * s), lim)
            continue
        zc = (2 * m * (2 * l - 1)) / (l * (l - 1))
        xr = ((l - m) / l) * math.sqrt(l * l - 4)
        if l == m + 1 and m >= 2:
            m1 = math.sqrt(1 / (l + m))
            P[l] = (m1 * P[l - 1] * (2 * l - 1) * c) / (l - m)
            R[l] = (m1 * R[l - 1] * (2 * l - 1) * c + m1 * T[l - 1] * zc) / xr
            T[l] = (m1 * T[l - 1] * (2 * l - 1) * c + m1 * R[l - 1] * zc) / xr
            continue
        if m == 1:
            m1 = math.sqrt((l - 1) / (l + 1))
            m2 = m1 * math.sqrt((l - 2) / l)
        else:
            m1 = math.sqrt((l - m) / (l + m))
            m2 = m1 * math.sqrt((l - m - 1) / (l + m - 1))
        yr = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
        P[l] = (m1 * P[l - 1] * (2 * l - 1) * c - m2 * P[l - 2] * (l - 1 + m)) / (l - m)
        R[l] = (m1 * R[l - 1] * (2 * l - 1) * c - m2 * R[l - 2] * yr + m1 * T[l - 1] * zc) / xr
        T[l] = (m1 * T[l - 1] * (2 * l - 1) * c - m2 * T[l - 2] * yr + m1 * R[l - 1] * zc) / xr
    return P, R, -T


def compute_Z_moments(pol: PolarizationType, mu: np.ndarray, greek: GreekCoefs, m: int):
    """Z⁺⁺(m), Z⁻⁺(m) on the stream cosines `mu` (compute_Z_matrices.jl:26-110). Returns [N,N] each."""
    mu = np.asarray(mu, dtype=np.float64)
    if not np.all((mu > 0) & (mu <= 1)):
        raise ValueError("all mu within compute_Z_moments have to be in ]0,1]")
    nq, n, lmax = len(mu), pol.n, len(greek.beta)
    fact = 0.5 if m == 0 else 1.0

    def pis(x):
        P, R, T = _prt_for_m(x, lmax, m)
        Pi = np.zeros((lmax, nq, n, n))
        Pi[:, :, 0, 0] = P
        if n >= 2:
            Pi[:, :, 1, 1] = R
        if n >= 3:
            Pi[:, :, 1, 2] = -T
            Pi[:, :, 2, 1] = -T
            Pi[:, :, 2, 2] = R
        if n == 4:
            Pi[:, :, 3, 3] = P
        return Pi

    B = np.zeros((lmax, n, n))
    B[:, 0, 0] = greek.beta
    if n >= 2:
        B[:, 0, 1] = B[:, 1, 0] = greek.gamma
        B[:, 1, 1] = greek.alpha
    if n >= 3:
        B[:, 2, 2] = greek.zeta
    if n == 4:
        B[:, 2, 3] = greek.epsilon
        B[:, 3, 2] = -greek.epsilon
        B[:, 3, 3] = greek.delta
    Pp, Pm = pis(mu)[m:], pis(-mu)[m:]
    left = np.einsum("liab,lbc->liac", Pp, B[m:])
    App = np.einsum("liac,ljcd->iajd", left, Pp)
    Amp = np.einsum("liac,ljcd->iajd", left, Pm)
    sgn = np.ones((n, n))
    sgn[:2, 2:] = -1
    sgn[2:, :2] = -1
    Zpp = (2 * fact * App).reshape(nq * n, nq * n)
    Zmp = (2 * fact * Amp * sgn[None, :, None, :]).reshape(nq * n, nq * n)
    return Zpp, Zmp


# ---- layer optics ---------------------------------------------------------------------------
@dataclass
class CoreScatteringOpticalProperties:
    """src/CoreRT/types.jl CoreScatteringOpticalProperties (host numpy version)."""
    tau: np.ndarray     # [S] or scalar
    varpi: np.ndarray   # [S] or scalar
    Zpp: np.ndarray     # [N,N] (shared by all S) or [S,N,N]
    Zmp: np.ndarray

    def __add__(self, other):
        if isinstance(other, CoreAbsorptionOpticalProperties):
            tau = self.tau + other.tau
            varpi = (self.tau * self.varpi) / np.where(tau > 0, tau, 1.0)
            return CoreScatteringOpticalProperties(tau, varpi, self.Zpp, self.Zmp)
        x, y = self, other
        tau = x.tau + y.tau
        wx, wy = x.tau * x.varpi, y.tau * y.varpi
        w = wx + wy
        varpi = w / np.where(tau > 0, tau, 1.0)
        if np.all(wx == 0.0):
            return CoreScatteringOpticalProperties(tau, varpi, y.Zpp, y.Zmp)
        if np.all(wy == 0.0):
            return CoreScatteringOpticalProperties(tau, varpi, x.Zpp, x.Zmp)
        w = np.atleast_1d(w)
        fx = (np.atleast_1d(wx) / w)[:, None, None]
        fy = (np.atleast_1d(wy) / w)[:, None, None]
        return CoreScatteringOpticalProperties(tau, varpi, fx * x.Zpp + fy * y.Zpp, fx * x.Zmp + fy * y.Zmp)


@dataclass
class CoreAbsorptionOpticalProperties:
    tau: np.ndarray


def createAero(tau_aer: float, ao: AerosolOptics, Zpp, Zmp) -> CoreScatteringOpticalProperties:
    f, w = ao.f_trunc, ao.ssa
    return CoreScatteringOpticalProperties(np.float64((1 - f * w) * tau_aer), np.float64((1 - f) * w / (1 - f * w)), Zpp, Zmp)


def get_scattering_interface(prev: str, scatter: bool, iz: int) -> str:
    """rt_helper_functions.jl:15-33 (iz 1-based)."""
    if iz == 1:
        return "11" if scatter else "00"
    if prev == "00":
        return "01" if scatter else "00"
    return "11" if scatter else "10"


# ---- surfaces (src/CoreRT/types.jl:470-536) -----------------------------------------------------------------
@dataclass
class LambertianSurfaceScalar:
    albedo: float


@dataclass
class LambertianSurfaceLegendre:
    """types.jl:540-543: albedo = sum_k legendre_coeff[k] P_k(x) on x = range(-1, 1, length = nSpec)."""
    legendre_coeff: Sequence[float]


@dataclass
class LambertianSurfaceSpline:
    """types.jl:546-549 with the interpolator already evaluated on the band grid: albedo[nSpec] = interpolator(wlGrid)."""
    albedo: Sequence[float]


def surface_albedo_spectrum(surface, nSpec: int, FT=np.float64) -> np.ndarray:
    """The per-point albedo of the spectrally varying Lambertian surfaces (lambertian_surface.jl:113-117, 178): Legendre
    basis by the three-term recurrence of compute_legendre_poly (legendre_functions.jl:223-252), arithmetic in FT."""
    if isinstance(surface, LambertianSurfaceSpline):
        a = np.asarray(surface.albedo, dtype=FT)
        if a.shape != (nSpec,):
            raise ValueError("LambertianSurfaceSpline: albedo must be given on the band grid (%d points)" % nSpec)
        return a
    c = np.asarray(surface.legendre_coeff, dtype=FT)
    if len(c) < 2:
        raise ValueError("LambertianSurfaceLegendre needs at least two coefficients (compute_legendre_poly asserts nmax > 1)")
    x = np.linspace(FT(-1), FT(1), nSpec).astype(FT)
    P = np.zeros((nSpec, len(c)), dtype=FT)
    P[:, 0] = 1
    P[:, 1] = x
    for n in range(2, len(c)):
        l = n - 1
        P[:, n] = ((2 * l + 1) * x * P[:, n - 1] - l * P[:, n - 2]) / (l + 1)
    return (P @ c).astype(FT)


@dataclass
class CoxMunkSurface:
    """src/CoreRT/types.jl:525-536: Cox-Munk (1954) ocean, isotropic slope variance 0.003 + 0.00512 U."""
    wind_speed: float
    n_water: Optional[complex] = None     # None -> Segelstein (1981) table at 550 nm (coxmunk_surface.jl:434-444)
    whitecap_albedo: float = 0.22
    include_whitecaps: bool = True
    shadowing: bool = True


# Segelstein (1981) liquid-water refractive index, the columns of src/CoreRT/Surfaces/water_refraction.jl:15-57 (data)
_WATER_NM = (
    200.0, 210.0, 220.0, 230.0, 240.0, 250.0, 260.0, 270.0, 280.0, 290.0,
    300.0, 310.0, 320.0, 330.0, 340.0, 350.0, 360.0, 370.0, 380.0, 390.0,
    400.0, 410.0, 420.0, 430.0, 440.0, 450.0, 460.0, 470.0, 480.0, 490.0,
    500.0, 510.0, 520.0, 530.0, 540.0, 550.0, 560.0, 570.0, 580.0, 590.0,
    600.0, 610.0, 620.0, 630.0, 640.0, 650.0, 660.0, 670.0, 680.0, 690.0,
    700.0, 720.0, 740.0, 760.0, 780.0, 800.0, 820.0, 840.0, 860.0, 880.0,
    900.0, 920.0, 940.0, 960.0, 980.0, 1000.0, 1050.0, 1100.0, 1150.0, 1200.0,
    1250.0, 1300.0, 1350.0, 1400.0, 1450.0, 1500.0, 1550.0, 1600.0, 1650.0, 1700.0,
    1750.0, 1800.0, 1850.0, 1900.0, 1950.0, 2000.0, 2100.0, 2200.0, 2300.0, 2400.0,
    2500.0, 2600.0,
)
_WATER_N = (
    1.396, 1.373, 1.362, 1.354, 1.349, 1.346, 1.343, 1.341, 1.339, 1.338,
    1.337, 1.336, 1.335, 1.335, 1.334, 1.334, 1.333, 1.333, 1.333, 1.332,
    1.332, 1.332, 1.331, 1.331, 1.331, 1.331, 1.330, 1.330, 1.330, 1.330,
    1.329, 1.329, 1.329, 1.329, 1.328, 1.328, 1.328, 1.328, 1.327, 1.327,
    1.327, 1.326, 1.326, 1.326, 1.325, 1.325, 1.325, 1.325, 1.324, 1.324,
    1.324, 1.323, 1.322, 1.322, 1.321, 1.320, 1.319, 1.319, 1.318, 1.317,
    1.316, 1.315, 1.314, 1.313, 1.312, 1.311, 1.308, 1.306, 1.303, 1.300,
    1.296, 1.293, 1.289, 1.285, 1.277, 1.268, 1.261, 1.255, 1.253, 1.255,
    1.260, 1.268, 1.279, 1.295, 1.306, 1.304, 1.279, 1.232, 1.188, 1.147,
    1.131, 1.129,
)
_WATER_K = (
    1.42e-07, 7.00e-08, 4.00e-08, 2.60e-08, 1.80e-08, 1.40e-08, 1.10e-08, 9.00e-09, 7.50e-09, 6.50e-09,
    6.00e-09, 4.60e-09, 3.50e-09, 2.70e-09, 2.20e-09, 1.80e-09, 1.60e-09, 1.40e-09, 1.30e-09, 1.30e-09,
    1.30e-09, 1.40e-09, 1.50e-09, 1.60e-09, 1.70e-09, 1.80e-09, 1.90e-09, 2.05e-09, 2.30e-09, 2.69e-09,
    3.21e-09, 3.81e-09, 4.36e-09, 4.78e-09, 5.14e-09, 5.69e-09, 6.49e-09, 7.63e-09, 9.22e-09, 1.09e-08,
    1.26e-08, 1.39e-08, 1.48e-08, 1.55e-08, 1.63e-08, 1.74e-08, 1.91e-08, 2.20e-08, 2.72e-08, 3.59e-08,
    4.78e-08, 7.50e-08, 1.10e-07, 1.43e-07, 1.65e-07, 1.72e-07, 1.63e-07, 1.46e-07, 1.32e-07, 1.28e-07,
    1.38e-07, 1.65e-07, 2.41e-07, 4.42e-07, 7.40e-07, 1.06e-06, 1.79e-06, 1.65e-06, 1.10e-06, 9.60e-07,
    1.32e-06, 2.26e-06, 4.58e-06, 1.07e-05, 2.94e-05, 5.88e-05, 7.15e-05, 6.71e-05, 5.68e-05, 4.65e-05,
    3.85e-05, 3.44e-05, 3.72e-05, 5.63e-05, 1.27e-04, 2.98e-04, 6.56e-04, 1.14e-03, 1.67e-03, 1.89e-03,
    1.67e-03, 1.19e-03,
)


@dataclass
class rpvSurfaceScalar:
    """rpvSurfaceScalar (types.jl:482-491): Rahman-Pinty-Verstraete BRDF, scalar (I -> I) only."""
    rho0: float
    rho_c: float
    k: float
    Theta: float


@dataclass
class RossLiSurfaceScalar:
    """RossLiSurfaceScalar (types.jl:505-512): f_vol K_RossThick + f_geo K_LiSparse + f_iso, scalar only."""
    fvol: float
    fgeo: float
    fiso: float


def brdf_reflectance(surface, n_stokes: int, mu: np.ndarray, m: int, nquad: int = 100) -> np.ndarray:
    """reflectance(brdf, pol_type, mu, m) (Surfaces/rpv_surface.jl:160-190) for the kernel-driven land BRDFs
    (rpv_surface.jl:99-150, rossli_surface.jl:12-98): the Fourier block [N, N] = ff / pi * int_0^pi rho(mu_i, mu_j, x) cos(m x) dx on
    the I -> I elements, ff = 1 (m = 0) or 2 -- a scene constant per moment like Z(m), evaluated on the host (N^2 x 100 values)
    and uploaded once; create_surface_layer! (vsm_brdf_surface) doubles the m = 0 block."""
    mu = np.asarray(mu, dtype=np.float64)
    x, w = np.polynomial.legendre.leggauss(nquad)
    phi = (0.5 * math.pi * (x + 1.0))[:, None, None]
    wphi = 0.5 * math.pi * w
    mi, mr = mu[None, :, None], mu[None, None, :]
    si, sr = np.sqrt(1.0 - mi ** 2), np.sqrt(1.0 - mr ** 2)
    ti, tr = si / mi, sr / mr                                       # tan(theta)
    if isinstance(surface, rpvSurfaceScalar):
        cosg = -mi * mr + si * sr * np.cos(phi)
        G = np.sqrt(ti ** 2 + tr ** 2 + 2.0 * ti * tr * np.cos(phi))
        th = -surface.Theta
        rho = (surface.rho0 * (mi * mr) ** (surface.k - 1.0) / (mi + mr) ** (1.0 - surface.k)
               * (1.0 - th ** 2) / (1.0 + th ** 2 + 2.0 * th * cosg) ** 1.5 * (1.0 + (1.0 - surface.rho_c) / (1.0 + G)))
    elif isinstance(surface, RossLiSurfaceScalar):
        cd, sd = np.cos(math.pi - phi), np.sin(math.pi - phi)
        cxi = np.clip(mi * mr + si * sr * cd, -1.0, 1.0)
        xi = np.arccos(cxi)
        K_vol = ((0.5 * math.pi - xi) * cxi + np.sin(xi)) / (mi + mr) - 0.25 * math.pi
        sec = 1.0 / mi + 1.0 / mr                                    # b/r = 1: primed angles == angles
        D2 = np.maximum(ti ** 2 + tr ** 2 - 2.0 * ti * tr * cd, 0.0)
        t = np.arccos(np.clip(2.0 * np.sqrt(D2 + (ti * tr * sd) ** 2) / sec, -1.0, 1.0))      # h/b = 2
        K_geo = (t - np.sin(t) * np.cos(t)) * sec / math.pi - sec + 0.5 * (1.0 + cxi) / (mi * mr)
        rho = surface.fiso + surface.fvol * K_vol + surface.fgeo * K_geo
    else:
        raise TypeError("brdf_reflectance: %r" % (surface,))
    block = np.tensordot(wphi * np.cos(m * phi[:, 0, 0]), rho, axes=(0, 0)) / math.pi
    N = len(mu) * n_stokes
    R = np.zeros((N, N))
    R[0::n_stokes, 0::n_stokes] = block
    return (1.0 if m == 0 else 2.0) * R


def water_refractive_index(lam_nm: float) -> complex:
    """water_refraction.jl:61-102: n linear, k log-linear in log(wavelength); clamped outside 200-2600 nm."""
    lg = [math.log(x) for x in _WATER_NM]
    x = math.log(float(lam_nm))
    if x <= lg[0]:
        return complex(_WATER_N[0], _WATER_K[0])
    if x >= lg[-1]:
        return complex(_WATER_N[-1], _WATER_K[-1])
    lo, hi = 1, len(lg)                     # the reference's 1-based bisection
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        if lg[mid - 1] <= x:
            lo = mid
        else:
            hi = mid
    lo, hi = lo - 1, hi - 1
    t = (x - lg[lo]) / (lg[hi] - lg[lo])
    lk0, lk1 = math.log(_WATER_K[lo]), math.log(_WATER_K[hi])
    return complex(_WATER_N[lo] + t * (_WATER_N[hi] - _WATER_N[lo]), math.exp(lk0 + t * (lk1 - lk0)))


def get_n_water(surf: CoxMunkSurface, lam_nm: float = 550.0) -> complex:
    """_get_n_water (coxmunk_surface.jl:434-444); every reference call site evaluates it at 550 nm."""
    return water_refractive_index(lam_nm) if surf.n_water is None else complex(surf.n_water)


NQUAD_PHI_BRDF = 100   # azimuth nodes of reflectance() (coxmunk_surface.jl:394, rpv_surface.jl:174: "hardcoded for now")


@dataclass
class RTNumericalParameters:
    """src/CoreRT/types.jl:713-756."""
    dtau_max_threshold: Optional[float] = None
    dtau_min_floor: Optional[float] = None


def planck_spectrum_wn(T: float, nu) -> np.ndarray:
    """src/SolarModel/SolarModel.jl:26-35: Planck radiance [mW / (m^2 sr cm^-1)] on a wavenumber grid [cm^-1]."""
    nu = np.asarray(nu, dtype=np.float64)
    return 1.1910427e-5 * nu ** 3 / (np.exp(1.4387752 * nu / T) - 1.0)


class SolarBeam:
    """The default source (Sources/solar_beam.jl): the SFI solar slot of the layer kernels."""


class ThermalEmission:
    """ThermalEmission (Sources/thermal_emission.jl): per-layer Planck volume source.  `ThermalEmission(B_layer=B)` with
    B [Nz, nSpec], or `ThermalEmission(T_layers, nu)` which evaluates planck_spectrum_wn per layer.

    `reset_slot_in_nonscattering_layers` (default False = the reference as written): rt_kernel! resets the `:thermal` slot of
    the AddedLayer only in its scatter branch (rt_kernel.jl:217-221); a non-scattering layer therefore interacts with the doubled
    slot the last scattering layer before it left in place, and a column that BEGINS with non-scattering layers carries the slot
    of moment m = 0 into moment m = 1.  True selects the corrected variant (a non-scattering layer emits nothing and carries
    nothing over)."""

    def __init__(self, T_layers=None, nu=None, B_layer=None, reset_slot_in_nonscattering_layers: bool = False):
        if B_layer is None and T_layers is not None:
            B_layer = np.stack([planck_spectrum_wn(float(T), nu) for T in T_layers])
        self.B_layer = None if B_layer is None else np.atleast_2d(np.asarray(B_layer, dtype=np.float64))
        self.reset_slot_in_nonscattering_layers = bool(reset_slot_in_nonscattering_layers)


@dataclass
class RTModel:
    """The fields of the reference's RTModel that rt_run consumes on this path (one band)."""
    architecture: object
    polarization_type: PolarizationType
    quad_points: QuadPoints
    sza: float
    vza: np.ndarray
    vaz: np.ndarray
    tau_rayl: np.ndarray          # τ_rayl[1]  [S, Nz]
    tau_abs: np.ndarray           # τ_abs[1]   [S, Nz]
    tau_aer: np.ndarray           # τ_aer[1]   [nAer, Nz]
    aerosol_optics: List[AerosolOptics]
    greek_rayleigh: GreekCoefs
    albedo: float                 # LambertianSurfaceScalar(albedo)
    m_max: int
    float_type: type = np.float64
    varpi_Cabannes: float = 1.0
    numerics: RTNumericalParameters = field(default_factory=RTNumericalParameters)
    F0: Optional[np.ndarray] = None   # [nStokes, S]; None = SolarBeam default e1
    surface: Optional[object] = None  # LambertianSurfaceScalar | CoxMunkSurface; None = LambertianSurfaceScalar(albedo)
    sources: Optional[Sequence] = None   # None = (SolarBeam(),); e.g. (ThermalEmission(...),) or (SolarBeam(), ThermalEmission(...))

    def __post_init__(self):
        if self.surface is None:
            self.surface = LambertianSurfaceScalar(float(self.albedo))
        elif isinstance(self.surface, LambertianSurfaceScalar):
            self.albedo = float(self.surface.albedo)


def model_from_arrays(architecture, polarization: str, l_trunc: int, sza: float, vza, vaz, tau_rayl, tau_abs=None,
                      tau_aer=None, aerosol_optics=(), depol=0.0, albedo=0.0, m_max=2, float_type=np.float64,
                      numerics=None, surface=None, sources=None) -> RTModel:
    """Build the RTModel subset directly from optical-depth arrays -- what the reference's tests
    do after model_from_parameters by overwriting model.τ_rayl/τ_abs/τ_aer
    (e.g. test/vlidort_baseline/cases/case_B_solar_tester.jl:62-74)."""
    pol = polarization_type(polarization)
    qp = rt_set_streams(l_trunc, sza, vza, pol, float_type)
    tau_rayl = np.atleast_2d(np.asarray(tau_rayl, dtype=np.float64))
    S, L = tau_rayl.shape
    tau_abs = np.zeros((S, L)) if tau_abs is None else np.atleast_2d(np.asarray(tau_abs, dtype=np.float64))
    tau_aer = (np.zeros((len(aerosol_optics), L)) if tau_aer is None
               else np.atleast_2d(np.asarray(tau_aer, dtype=np.float64)))
    return RTModel(architecture, pol, qp, float(sza), np.asarray(vza, float), np.asarray(vaz, float), tau_rayl,
                   tau_abs, tau_aer, list(aerosol_optics), get_greek_rayleigh(depol), float(albedo), int(m_max),
                   float_type, 1.0, numerics or RTNumericalParameters(), None, surface, sources)


def constructCoreOpticalProperties(model: RTModel, m: int) -> List[CoreScatteringOpticalProperties]:
    """compEffectiveLayerProperties.jl:11-65 for one band, noRS."""
    mu = model.quad_points.qp_mu.astype(np.float64)
    Nz = model.tau_rayl.shape[1]
    Rpp, Rmp = compute_Z_moments(model.polarization_type, mu, model.greek_rayleigh, m)
    combo = [CoreScatteringOpticalProperties(model.tau_rayl[:, i], np.float64(model.varpi_Cabannes), Rpp, Rmp)
             for i in range(Nz)]
    for ia, ao in enumerate(model.aerosol_optics):
        App, Amp = compute_Z_moments(model.polarization_type, mu, ao.greek_coefs, m)
        combo = [combo[i] + createAero(model.tau_aer[ia, i], ao, App, Amp) for i in range(Nz)]
    return [combo[i] + CoreAbsorptionOpticalProperties(model.tau_abs[:, i]) for i in range(Nz)]


@dataclass
class LayerOpticsComponents:
    """One layer's optics with Z carried as coefficients over the component phase matrices (not materialised)."""
    tau: np.ndarray       # [S]
    varpi: np.ndarray     # [S]
    coef: np.ndarray      # [C] (unit vector: ONE component's Z, shared by all points) or [S, C] (per-point mix)


def constructLayerOpticsComponents(model: RTModel, m: int):
    """compEffectiveLayerProperties.jl:11-65 with the SAME pairwise arithmetic as constructCoreOpticalProperties
    (types.jl:1262-1308), except that Z = fx Zx + fy Zy is tracked as coefficients over the components
    [Rayleigh, aerosol 1, ...] -- the device mixes where Z is consumed (vsm_layer_forward_mix).
    Returns (Zc_pp [C,N,N], Zc_mp [C,N,N], [LayerOpticsComponents per layer])."""
    mu = model.quad_points.qp_mu.astype(np.float64)
    S, Nz = model.tau_rayl.shape
    Zs = [compute_Z_moments(model.polarization_type, mu, model.greek_rayleigh, m)]
    for ao in model.aerosol_optics:
        Zs.append(compute_Z_moments(model.polarization_type, mu, ao.greek_coefs, m))
    C = len(Zs)
    layers = []
    for i in range(Nz):
        tau = model.tau_rayl[:, i].astype(np.float64)
        varpi = np.full(S, np.float64(model.varpi_Cabannes))
        coef = np.eye(C)[0]
        for ia, ao in enumerate(model.aerosol_optics):
            y = createAero(model.tau_aer[ia, i], ao, None, None)
            tau2 = tau + y.tau
            wx, wy = tau * varpi, y.tau * y.varpi
            w = wx + wy
            varpi2 = w / np.where(tau2 > 0, tau2, 1.0)
            e = np.eye(C)[ia + 1]
            if np.all(wx == 0.0):
                coef = e
            elif np.all(wy == 0.0):
                pass
            else:
                fx, fy = (np.atleast_1d(wx) / w)[:, None], (np.atleast_1d(wy) / w)[:, None]
                coef = fx * np.broadcast_to(coef, (S, C)) + fy * e[None, :]
            tau, varpi = tau2, varpi2
        tau3 = tau + model.tau_abs[:, i]
        varpi = (tau * varpi) / np.where(tau3 > 0, tau3, 1.0)
        layers.append(LayerOpticsComponents(tau3, varpi, coef))
    return np.stack([z[0] for z in Zs]), np.stack([z[1] for z in Zs]), layers


def extractEffectiveProps(lods: List[CoreScatteringOpticalProperties], FT):
    """compEffectiveLayerProperties.jl:75-93 -> (interface tags, τ_sum_all [S, Nz+1])."""
    S = len(np.atleast_1d(lods[0].tau))
    tau_sum = np.zeros((S, len(lods) + 1))
    tags, tag = [], "00"
    for iz, lo in enumerate(lods):
        scatter = bool(np.max(lo.tau * lo.varpi) > 2 * np.finfo(FT).eps)
        tag = get_scattering_interface(tag, scatter, iz + 1)
        tags.append(tag)
        tau_sum[:, iz + 1] = tau_sum[:, iz] + lo.tau
    return tags, tau_sum


def doubling_number(dtau_max, tau_end, FT):
    """rt_helper_functions.jl:49-69 (arithmetic in FT)."""
    dtau_max, tau_end = FT(dtau_max), FT(tau_end)
    if tau_end <= dtau_max:
        return tau_end, 0
    q1, q2, q3 = np.log10(FT(2)), np.log10(dtau_max), np.log10(tau_end)
    tlimit = FT((q3 - q2) / q1)
    nlimit = int(math.floor(tlimit))
    if FT(tlimit - FT(nlimit)) < np.finfo(FT).eps:
        return dtau_max, nlimit
    nd = nlimit + 1
    return FT(10) ** (q3 - q1 * FT(nd)), nd


def ndoubl_from_max(tw, qp: QuadPoints, FT, numerics: RTNumericalParameters) -> int:
    """The ndoubl rule of rt_kernel.jl:266-287 from tw = maximum(tau .* varpi) (already in FT)."""
    thr = FT(0.001 if numerics.dtau_max_threshold is None else numerics.dtau_max_threshold)
    floor_val = FT(1024 * np.finfo(FT).eps if numerics.dtau_min_floor is None else numerics.dtau_min_floor)
    real = qp.qp_mu[qp.wt_mu > np.finfo(FT).eps]
    mu_min = FT(np.min(real) if len(real) else np.min(qp.qp_mu))
    tw = FT(tw)
    dtau_max = max(floor_val, min(tw, FT(thr * mu_min)))
    _, nd = doubling_number(dtau_max, tw, FT)
    return nd


def get_dtau_ndoubl(tau: np.ndarray, varpi: np.ndarray, qp: QuadPoints, FT, numerics: RTNumericalParameters):
    """rt_kernel.jl:266-287.  Returns (dτ[S] in FT, ndoubl); ndoubl is batch-global by construction."""
    nd = ndoubl_from_max(FT(np.max(tau.astype(FT) * varpi.astype(FT))), qp, FT, numerics)
    return (tau / FT(2 ** nd)).astype(FT), nd


def layer_mix_modes(model: "RTModel"):
    """How `rayleigh + createAero(...) + ...` (compEffectiveLayerProperties.jl:43-54) resolves per (aerosol, layer): the
    mixing `+` (types.jl:1262-1292) branches on batch-global conditions -- 1: no spectral point scatters before this aerosol
    (all tau_x varpi_x == 0) -> the sum takes the aerosol's Z; 2: the aerosol does not scatter -> Z unchanged; 0: per-point
    mix.  Returns (modes int[nAer, Nz], per layer (mixed?, index of the single component whose Z the layer uses))."""
    nA, L = len(model.aerosol_optics), model.tau_rayl.shape[1]
    modes = np.zeros((nA, L), dtype=np.int32)
    rayl_zero = np.all(np.asarray(model.tau_rayl) * float(model.varpi_Cabannes) == 0.0, axis=0)
    zcomp = []
    for l in range(L):
        x_zero, mixed, k = bool(rayl_zero[l]), False, 0
        for ia, ao in enumerate(model.aerosol_optics):
            y = createAero(model.tau_aer[ia, l], ao, None, None)
            wy_zero = bool(y.tau * y.varpi == 0.0)
            if x_zero:
                modes[ia, l], k, mixed = 1, ia + 1, False
                x_zero = wy_zero
            elif wy_zero:
                modes[ia, l] = 2
            else:
                modes[ia, l], mixed = 0, True
        zcomp.append((mixed, k))
    return modes, zcomp


# ---- linearized inputs (src/CoreRT/parameter_layout.jl:20-66, types_lin.jl:141-150) -------------------------
@dataclass
class ParameterLayout:
    """Order of the Jacobian slots: [7 per aerosol] ++ gases ++ surface (parameter_layout.jl:28-56)."""
    n_aerosols: int = 0
    n_gases: int = 0
    n_surface: int = 0
    aerosol_params: int = 7

    @property
    def n_layer_params(self):
        return self.aerosol_params * self.n_aerosols + self.n_gases

    @property
    def n_total(self):
        return self.n_layer_params + self.n_surface

    def surface_index(self, i: int) -> int:  # 0-based slot of the i-th (0-based) surface parameter
        return self.n_layer_params + i


@dataclass
class CoreScatteringOpticalPropertiesLin:
    """d(tau, varpi, Z++, Z-+)/dx for one layer (host numpy)."""
    tau_dot: np.ndarray             # [S, p]
    varpi_dot: np.ndarray           # [S, p]
    Zpp_dot: Optional[np.ndarray]   # [p, N, N] / [p, S, N, N] / None
    Zmp_dot: Optional[np.ndarray]


@dataclass
class LinAerosolOptics:
    """lin_aerosol_optics[iaer] (compEffectiveLayerProperties_lin.jl:330-395): derivatives of the aerosol's Greek coefficients,
    single-scattering albedo and truncation factor with respect to the four Mie parameters (n_r, n_i, r_m, sigma_r).  Mie
    theory (their producer) is upstream of the hot path: the numbers are inputs here."""
    lin_greek_coefs: List[GreekCoefs]   # 4 entries
    ssa_dot: np.ndarray                 # [4]  (lin_aerosol_optics.ω̃̇)
    f_trunc_dot: np.ndarray             # [4]  (lin_aerosol_optics.ḟᵗ)


@dataclass
class LinModel:
    """The part of the reference's lin_model the hot path consumes: tau_abs_dot[g][S, Nz] = d tau_abs / d x_g (lin_model.τ̇_abs);
    for aerosol Jacobians tau_aer_dot[iaer][7, Nz] = d tau_aer / d (tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p) per layer
    (lin_model.τ̇_aer) and lin_aerosol_optics[iaer]."""
    tau_abs_dot: List[np.ndarray]
    tau_aer_dot: Optional[np.ndarray] = None
    lin_aerosol_optics: Optional[List[LinAerosolOptics]] = None

    @property
    def n_aer(self) -> int:
        return 0 if self.tau_aer_dot is None else len(self.tau_aer_dot)


def createAeroLin(tau_aer: float, ao: AerosolOptics, tau_aer_dot7: np.ndarray, lao: LinAerosolOptics):
    """The scalar part of createAero with derivatives (compEffectiveLayerProperties_lin.jl:330-395): delta-M scaled
    (tau, varpi) and d/d(7 sub-parameters); only the Mie slots 1..4 (0-based) move varpi."""
    f, w = ao.f_trunc, ao.ssa
    wd, fd = np.zeros(7), np.zeros(7)
    wd[1:5], fd[1:5] = lao.ssa_dot, lao.f_trunc_dot
    g = 1.0 - f * w
    td = g * np.asarray(tau_aer_dot7, dtype=np.float64) - (f * wd + w * fd) * tau_aer
    vd = (wd * (1.0 - f) - fd * (w * (1.0 - w))) / g ** 2
    return g * tau_aer, (1.0 - f) * w / g, td, vd


def constructCoreOpticalPropertiesLin(model: RTModel, lin_model: LinModel, lods: List[CoreScatteringOpticalProperties],
                                      m: int = 0) -> List[CoreScatteringOpticalPropertiesLin]:
    """d(tau, varpi, Z)/dx per layer (constructCoreOpticalProperties with lin_model, compEffectiveLayerProperties_lin.jl:43-197).

    The reference chains the quotient rule through its pairwise `+` (types_lin.jl:196-380); the layer's optics are
    tau = sum_c tau_c + tau_abs, W = sum_c w_c (w_c = tau_c varpi_c), varpi = W / tau, Z = sum_c w_c Z_c / W over the scatterers
    c = Rayleigh, aerosol 1, ..., so for a parameter p of aerosol a (w_dot = tau_dot_a varpi_a + tau_a varpi_dot_a):
        tau_dot = tau_dot_a,  varpi_dot = (w_dot - varpi tau_dot) / tau,  Z_dot = (w_dot (Z_a - Z) + w_a Zdot_a) / W
    and for a gas: tau_dot = tau_abs_dot, varpi_dot = -varpi tau_dot / tau, Z_dot = 0 -- evaluated here in that closed form."""
    out = []
    nA = lin_model.n_aer
    if nA == 0:
        for iz, lo in enumerate(lods):
            tau = np.atleast_1d(lo.tau)
            varpi = np.broadcast_to(np.asarray(lo.varpi), tau.shape)
            td = np.stack([g[:, iz] for g in lin_model.tau_abs_dot], axis=1)
            wd = -(varpi / np.where(tau > 0, tau, 1.0))[:, None] * td
            out.append(CoreScatteringOpticalPropertiesLin(td, wd, None, None))
        return out
    if nA != len(model.aerosol_optics):
        raise ValueError("lin_model carries %d aerosol blocks, the model %d aerosols" % (nA, len(model.aerosol_optics)))
    pol, mu = model.polarization_type, model.quad_points.qp_mu.astype(np.float64)
    S, Nz = model.tau_rayl.shape
    nG = len(lin_model.tau_abs_dot)
    pl = 7 * nA + nG
    Zc = [compute_Z_moments(pol, mu, model.greek_rayleigh, m)] + [compute_Z_moments(pol, mu, ao.greek_coefs, m)
                                                                  for ao in model.aerosol_optics]
    Zcp, Zcm = np.stack([z[0] for z in Zc]), np.stack([z[1] for z in Zc])          # [C, N, N]
    Zdp = [np.stack([compute_Z_moments(pol, mu, g, m)[0] for g in lao.lin_greek_coefs]) for lao in lin_model.lin_aerosol_optics]
    Zdm = [np.stack([compute_Z_moments(pol, mu, g, m)[1] for g in lao.lin_greek_coefs]) for lao in lin_model.lin_aerosol_optics]
    N = Zcp.shape[-1]
    for iz in range(Nz):
        tau_c = [model.tau_rayl[:, iz].astype(np.float64)]
        var_c = [np.full(S, np.float64(model.varpi_Cabannes))]
        aer = []
        for ia, ao in enumerate(model.aerosol_optics):
            tm, vm, td7, vd7 = createAeroLin(model.tau_aer[ia, iz], ao, lin_model.tau_aer_dot[ia][:, iz], lin_model.lin_aerosol_optics[ia])
            tau_c.append(np.full(S, tm))
            var_c.append(np.full(S, vm))
            aer.append((td7, vd7))
        w_c = np.stack([t * v for t, v in zip(tau_c, var_c)])                      # [C, S]
        W = w_c.sum(axis=0)
        tau = np.sum(tau_c, axis=0) + model.tau_abs[:, iz]
        varpi = W / tau
        fz = w_c / W                                                                # [C, S]
        Zp = np.einsum("cs,cij->sij", fz, Zcp)
        Zm = np.einsum("cs,cij->sij", fz, Zcm)
        tau_dot = np.zeros((S, pl))
        varpi_dot = np.zeros((S, pl))
        Zpp_dot = np.zeros((pl, S, N, N))
        Zmp_dot = np.zeros((pl, S, N, N))
        for ia in range(nA):
            td7, vd7 = aer[ia]
            ta, va = tau_c[1 + ia], var_c[1 + ia]
            for k in range(7):
                p = 7 * ia + k
                wdot = td7[k] * va + ta * vd7[k]
                tau_dot[:, p] = td7[k]
                varpi_dot[:, p] = (wdot - varpi * td7[k]) / tau
                Zpp_dot[p] = (wdot / W)[:, None, None] * (Zcp[1 + ia][None] - Zp)
                Zmp_dot[p] = (wdot / W)[:, None, None] * (Zcm[1 + ia][None] - Zm)
                if 1 <= k <= 4:
                    Zpp_dot[p] += fz[1 + ia][:, None, None] * Zdp[ia][k - 1][None]
                    Zmp_dot[p] += fz[1 + ia][:, None, None] * Zdm[ia][k - 1][None]
        for g in range(nG):
            p = 7 * nA + g
            tau_dot[:, p] = lin_model.tau_abs_dot[g][:, iz]
            varpi_dot[:, p] = -(varpi / tau) * tau_dot[:, p]
        out.append(CoreScatteringOpticalPropertiesLin(tau_dot, varpi_dot, Zpp_dot, Zmp_dot))
    return out
