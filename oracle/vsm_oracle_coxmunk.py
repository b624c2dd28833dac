"""CPU ORACLE, Cox-Munk ocean surface (forward + wind-speed Jacobian) -- test infrastructure, NOT the product path.

numpy restatement (vectorised over the azimuth quadrature and the stream pairs, same arithmetic per element) of
  src/CoreRT/Surfaces/fresnel.jl:25-124               fresnel_coefficients, fresnel_mueller, stokes_rotation_matrix
  src/CoreRT/Surfaces/water_refraction.jl:61-102      water_refractive_index (Segelstein 1981 table = DATA fixture)
  src/CoreRT/Surfaces/coxmunk_surface.jl:23-128       slope variance, PDF, whitecaps, Smith shadowing (+ d/dsigma^2)
  src/CoreRT/Surfaces/coxmunk_surface.jl:146-267      coxmunk_geometry (facet tilt, local incidence, rotation angles)
  src/CoreRT/Surfaces/coxmunk_surface.jl:277-370      coxmunk_brdf_mueller(_and_deriv)
  src/CoreRT/Surfaces/coxmunk_surface.jl:381-460      reflectance / reflectance_and_deriv (100-point GL over [0, pi])
  src/CoreRT/Surfaces/coxmunk_surface.jl:481-569      apply_ss_correction! (TMS), _fourier_coeff_element
  src/CoreRT/Surfaces/rpv_surface.jl:51-97            create_surface_layer!(::AbstractSurfaceType) -- the forward builder
  src/CoreRT/Surfaces/coxmunk_surface_lin.jl:27-102   create_surface_layer! (lin): wind-speed slot, t-- = 0 quirk

PARITY PIN: the reference commits no Cox-Munk numbers; its own tests (test/test_coxmunk.jl) are analytic
known answers and properties (Fresnel at normal / Brewster / grazing incidence, Mueller block structure, rotation
group laws, slope-PDF normalisation, reciprocity, energy bound, analytic-vs-finite-difference derivatives at
rtol 1e-3 / 1e-2).  tests/test_oracle_coxmunk.py runs every one of them against this file at the reference's own
tolerances.  The linearized outputs stay "parity unpinned" in the sense of oracle/vsm_oracle_lin.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
from scipy.special import erfc

from . import vsm_oracle as O
from . import vsm_oracle_lin as OL

_FIX = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                    "segelstein1981_water.json")


# ---- water refractive index (water_refraction.jl:61-102) -----------------------------------------
def water_refractive_index(lam_nm: float) -> complex:
    with open(_FIX) as f:
        tab = json.load(f)
    nm, nr, ki = (np.asarray(tab[k], dtype=np.float64) for k in ("wavelength_nm", "n_real", "k_imag"))
    lg, lk = np.log(nm), np.log(ki)
    x = math.log(float(lam_nm))
    if x <= lg[0]:
        return complex(nr[0], ki[0])
    if x >= lg[-1]:
        return complex(nr[-1], ki[-1])
    lo, hi = 0, len(nm) - 1
    while hi - lo > 1:
        mid = (lo + hi + 2) // 2 - 1          # (lo + hi) >> 1 in the reference's 1-based indices
        if lg[mid] <= x:
            lo = mid
        else:
            hi = mid
    t = (x - lg[lo]) / (lg[hi] - lg[lo])
    return complex(nr[lo] + t * (nr[hi] - nr[lo]), math.exp(lk[lo] + t * (lk[hi] - lk[lo])))


@dataclass
class CoxMunkSurface:
    """src/CoreRT/types.jl:525-536."""
    wind_speed: float
    n_water: Optional[complex] = None
    whitecap_albedo: float = 0.22
    include_whitecaps: bool = True
    shadowing: bool = True


def get_n_water(surf: CoxMunkSurface, lam_nm: float = 550.0) -> complex:
    """coxmunk_surface.jl:434-444 (_get_n_water; every call site uses the 550 nm default)."""
    return water_refractive_index(lam_nm) if surf.n_water is None else complex(surf.n_water)


# ---- Fresnel (fresnel.jl) -----------------------------------------------------------------------------
def fresnel_coefficients(n_rel: complex, cos_i):
    cos_i = np.asarray(cos_i)
    CT = np.complex64 if cos_i.dtype == np.float32 else np.complex128
    n_rel = CT(n_rel)
    sin2 = np.maximum(0, 1 - cos_i ** 2)
    cos_t = np.sqrt((1 - sin2 / n_rel ** 2).astype(CT))
    r_s = (cos_i - n_rel * cos_t) / (cos_i + n_rel * cos_t)
    r_p = (n_rel * cos_i - cos_t) / (n_rel * cos_i + cos_t)
    return r_s, r_p


def fresnel_mueller(r_s, r_p, n: int):
    """[..., n, n]; the reference fills SMatrix column-major (fresnel.jl:52-90)."""
    rs2, rp2 = np.abs(r_s) ** 2, np.abs(r_p) ** 2
    M = np.zeros(np.shape(rs2) + (n, n), dtype=rs2.dtype)
    M[..., 0, 0] = (rs2 + rp2) / 2
    if n == 1:
        return M
    rsp = r_s * np.conj(r_p)
    M[..., 1, 1] = (rs2 + rp2) / 2
    M[..., 0, 1] = M[..., 1, 0] = (rs2 - rp2) / 2
    if n >= 3:
        M[..., 2, 2] = rsp.real
    if n == 4:
        M[..., 3, 3] = rsp.real
        M[..., 2, 3] = rsp.imag
        M[..., 3, 2] = -rsp.imag
    return M


def stokes_rotation_matrix(phi, n: int):
    phi = np.asarray(phi)
    L = np.zeros(phi.shape + (n, n), dtype=phi.dtype)
    L[..., 0, 0] = 1
    if n == 1:
        return L
    c2, s2 = np.cos(2 * phi), np.sin(2 * phi)
    L[..., 1, 1] = c2
    if n >= 3:
        L[..., 2, 2] = c2
        L[..., 1, 2] = s2
        L[..., 2, 1] = -s2
    if n == 4:
        L[..., 3, 3] = 1
    return L


# ---- helpers (coxmunk_surface.jl:23-128) ----------------------------------------------------------------
def wind_to_sigma2(U):
    return 0.003 + 0.00512 * U


def cox_munk_pdf(zx, zy, s2):
    return np.exp(-(zx ** 2 + zy ** 2) / (2 * s2)) / (2 * np.pi * s2)


def whitecap_fraction(U):
    return 0.0 if U <= 0 else 2.95e-6 * U ** 3.52


def whitecap_fraction_deriv(U):
    return 0.0 if U <= 0 else 2.95e-6 * 3.52 * U ** 2.52


def _nu(mu, s2):
    cot = mu / np.sqrt(np.maximum(1e-30, 1 - mu ** 2))
    return cot / (np.sqrt(2.0) * np.sqrt(s2))


def smith_Lambda(mu, s2):
    mu = np.asarray(mu, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        nu = _nu(mu, s2)
        L = (np.exp(-nu ** 2) / (np.sqrt(2 * np.pi) * nu) - erfc(nu)) / 2
    return np.where(mu <= 0, 1e10, np.maximum(0.0, L))


def shadow_factor(mu_i, mu_r, s2):
    return 1.0 / (1.0 + smith_Lambda(mu_i, s2) + smith_Lambda(mu_r, s2))


def cox_munk_pdf_dsigma2(zx, zy, s2):
    Z2 = zx ** 2 + zy ** 2
    return cox_munk_pdf(zx, zy, s2) * (Z2 - 2 * s2) / (2 * s2 ** 2)


def smith_Lambda_dsigma2(mu, s2):
    mu = np.asarray(mu, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        nu = _nu(mu, s2)
        e = np.exp(-nu ** 2)
        raw = (e / (np.sqrt(2 * np.pi) * nu) - erfc(nu)) / 2
        dL = (e * (-2 * nu ** 2 - 1) / (np.sqrt(2 * np.pi) * nu ** 2) + 2 / np.sqrt(np.pi) * e) / 2
        out = dL * (-nu / (2 * s2))
    return np.where((mu <= 0) | (raw <= 0), 0.0, out)


def shadow_factor_dsigma2(mu_i, mu_r, s2):
    S = shadow_factor(mu_i, mu_r, s2)
    return -S ** 2 * (smith_Lambda_dsigma2(mu_i, s2) + smith_Lambda_dsigma2(mu_r, s2))


# ---- facet geometry (coxmunk_surface.jl:146-267) -------------------------------------------------------
def coxmunk_geometry(mu_i, mu_r, dphi):
    mu_i, mu_r, dphi = np.broadcast_arrays(*(np.asarray(x, dtype=np.float64) for x in (mu_i, mu_r, dphi)))
    si = np.sqrt(np.maximum(0.0, 1 - mu_i ** 2))
    sr = np.sqrt(np.maximum(0.0, 1 - mu_r ** 2))
    cd, sd = np.cos(dphi), np.sin(dphi)
    nx, ny, nz = -si + sr * cd, sr * sd, mu_i + mu_r
    norm = np.sqrt(nx ** 2 + ny ** 2 + nz ** 2)
    degenerate = norm < 1e-15
    nrm = np.where(degenerate, 1.0, norm)
    nx, ny, nz = nx / nrm, ny / nrm, nz / nrm
    cos_b = np.maximum(1e-10, nz)
    cos_loc = np.clip((mu_i + mu_r) / (2 * cos_b), 0.0, 1.0)
    zx, zy = -nx / cos_b, -ny / cos_b
    cosT = -mu_i * mu_r + si * sr * cd
    sinT = np.sqrt(np.maximum(0.0, 1 - cosT ** 2))
    sp = np.stack([-mu_i * sr * sd, mu_i * sr * cd - si * mu_r, -si * sr * sd])
    ip = np.stack([mu_i * ny, -mu_i * nx - si * nz, si * ny])
    rp = np.stack([(-sr * sd) * nz - mu_r * ny, mu_r * nx - (-sr * cd) * nz, (-sr * cd) * ny - (-sr * sd) * nx])
    ki = np.stack([si, np.zeros_like(si), -mu_i])
    kr = np.stack([-sr * cd, -sr * sd, mu_r])

    def angle(pl, k):
        dot = np.sum(sp * pl, axis=0)
        msp, mpl = np.sqrt(np.sum(sp ** 2, axis=0)), np.sqrt(np.sum(pl ** 2, axis=0))
        tiny = (msp < 1e-15) | (mpl < 1e-15)
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.clip(dot / (msp * mpl), -1.0, 1.0)
        cr = np.cross(sp, pl, axis=0)
        sgn = np.sum(k * cr, axis=0)
        a = np.where(sgn >= 0, np.arccos(c), -np.arccos(c))
        return np.where(tiny, 0.0, a)

    small = sinT < 1e-12
    a1 = np.where(small, 0.0, angle(ip, ki))
    a2 = np.where(small, 0.0, angle(rp, kr))
    one, zero = np.ones_like(cos_b), np.zeros_like(cos_b)
    pick = lambda v, d: np.where(degenerate, d, v)
    return dict(cos_beta=pick(cos_b, one), cos_theta_local=pick(cos_loc, one), zx=pick(zx, zero), zy=pick(zy, zero),
                alpha1=pick(a1, zero), alpha2=pick(a2, zero))


# ---- BRDF Mueller matrix (+ d/dU) (coxmunk_surface.jl:277-370) -----------------------------------------
def coxmunk_brdf_mueller_and_deriv(surf: CoxMunkSurface, n: int, mu_i, mu_r, dphi, n_water: Optional[complex] = None):
    """Returns (M, dM/dU) of shape broadcast(mu_i, mu_r, dphi) + (n, n).  `mu_i` is the FIRST positional argument of
    the reference function (reflectance() passes the ROW stream there)."""
    nw = get_n_water(surf) if n_water is None else complex(n_water)
    U = float(surf.wind_speed)
    s2 = wind_to_sigma2(U)
    mu_i, mu_r, dphi = np.broadcast_arrays(*(np.asarray(x, dtype=np.float64) for x in (mu_i, mu_r, dphi)))
    g = coxmunk_geometry(mu_i, mu_r, dphi)
    r_s, r_p = fresnel_coefficients(nw, g["cos_theta_local"])
    MF = fresnel_mueller(r_s, r_p, n)
    L1 = stokes_rotation_matrix(-g["alpha1"], n)
    L2 = stokes_rotation_matrix(g["alpha2"], n)
    Mfac = L2 @ MF @ L1
    gw = 1.0 / (4 * mu_i * mu_r * g["cos_beta"] ** 4)
    P = cox_munk_pdf(g["zx"], g["zy"], s2)
    dP = cox_munk_pdf_dsigma2(g["zx"], g["zy"], s2)
    if surf.shadowing:
        S = shadow_factor(mu_i, mu_r, s2)
        dS = shadow_factor_dsigma2(mu_i, mu_r, s2)
        pre, dpre = P * S * gw, (dP * S + P * dS) * gw
    else:
        pre, dpre = P * gw, dP * gw
    glint = pre[..., None, None] * Mfac
    dglint = (dpre * 0.00512)[..., None, None] * Mfac
    if not surf.include_whitecaps:
        return glint, dglint
    f, df = whitecap_fraction(U), whitecap_fraction_deriv(U)
    wc = np.zeros((n, n))
    wc[0, 0] = surf.whitecap_albedo / np.pi
    return (1 - f) * glint + f * wc, (1 - f) * dglint + df * (wc - glint)


def coxmunk_brdf_mueller(surf, n, mu_i, mu_r, dphi, n_water=None):
    return coxmunk_brdf_mueller_and_deriv(surf, n, mu_i, mu_r, dphi, n_water)[0]


def azimuthal_kernel(n: int, m: int, dphi):
    """[..., n, n]: cos(m dphi) on the (I,Q)x(I,Q) / (U,V)x(U,V) blocks, sin(m dphi) on the cross blocks
    (coxmunk_surface.jl:370-379)."""
    dphi = np.asarray(dphi, dtype=np.float64)
    iq = np.arange(n) <= 1
    same = iq[:, None] == iq[None, :]
    return np.where(same, np.cos(m * dphi)[..., None, None], np.sin(m * dphi)[..., None, None])


NQUAD_PHI = 100


def reflectance_and_deriv(surf: CoxMunkSurface, n: int, mu, m: int, n_water=None):
    """Fourier moment m of the BRDF and of its wind-speed derivative: [Nmu n, Nmu n] each (coxmunk_surface.jl:381-420)."""
    mu = np.asarray(mu, dtype=np.float64)
    Nm = len(mu)
    phi, w = O.gauleg(NQUAD_PHI, 0.0, np.pi)
    M, dM = coxmunk_brdf_mueller_and_deriv(surf, n, mu[:, None, None], mu[None, :, None], phi[None, None, :], n_water)
    az = azimuthal_kernel(n, m, phi)                                    # [phi, n, n]
    R = np.einsum("f,ijfab,fab->iajb", w, M, az).reshape(Nm * n, Nm * n)
    dR = np.einsum("f,ijfab,fab->iajb", w, dM, az).reshape(Nm * n, Nm * n)
    ff = 1.0 if m == 0 else 2.0
    return ff * R / np.pi, ff * dR / np.pi


def reflectance(surf, n, mu, m, n_water=None):
    return reflectance_and_deriv(surf, n, mu, m, n_water)[0]


def fourier_coeff_element(surf, n, mu_i, mu_j, m, n_water=None):
    """coxmunk_surface.jl:553-569 for all (si, sj) at once: [n, n]."""
    phi, w = O.gauleg(NQUAD_PHI, 0.0, np.pi)
    M = coxmunk_brdf_mueller(surf, n, mu_i, mu_j, phi, n_water)
    ff = 1.0 if m == 0 else 2.0
    return ff * np.einsum("f,fab,fab->ab", w, M, azimuthal_kernel(n, m, phi)) / np.pi


def ss_correction_coefficients(surf, pol, vza, vaz, mu0, m_max, n_water=None):
    """The per-geometry factor of apply_ss_correction! (coxmunk_surface.jl:481-545): M_exact[:,1] - M_fourier[:,1], [nV, n]."""
    n = pol.n
    out = np.zeros((len(vza), n))
    for iv in range(len(vza)):
        mu_v = O.cosd(vza[iv])
        dphi = math.radians(vaz[iv])
        Mex = coxmunk_brdf_mueller(surf, n, mu_v, mu0, dphi, n_water)
        Mf = np.zeros((n, n))
        for m in range(m_max + 1):
            wm = 0.5 if m == 0 else 1.0
            Mf += wm * azimuthal_kernel(n, m, dphi) * fourier_coeff_element(surf, n, mu_v, mu0, m, n_water)
        out[iv] = Mex[:, 0] - Mf[:, 0]
    return out


def apply_ss_correction(R_SFI, surf, pol, vza, vaz, mu0, tau_total, m_max, n_water=None):
    """R_SFI [nV, n, S] += mu0 exp(-tau/mu0) (M_exact - M_fourier)[:, 1]  (coxmunk_surface.jl:536-544)."""
    c = ss_correction_coefficients(surf, pol, vza, vaz, mu0, m_max, n_water)
    att = mu0 * np.exp(-np.asarray(tau_total, dtype=np.float64) / mu0)
    R_SFI += (c[:, :, None] * att[None, None, :]).astype(R_SFI.dtype)


# ---- surface layers -------------------------------------------------------------------------------------
def create_surface_layer_brdf(rho_m, added: O.AddedLayer, m, pol, qp, tau_sum, FT):
    """rpv_surface.jl:51-97 with `rho_m` = reflectance(brdf, pol, qp_mu, m)."""
    N = added.r_mp.shape[1]
    n = pol.n
    R_surf = (FT(2) * rho_m if m == 0 else rho_m).astype(FT)
    I0N = np.zeros(N, dtype=FT)
    i0 = n * qp.imu0
    I0N[i0:i0 + n] = pol.I0
    att = np.exp(-np.asarray(tau_sum, dtype=FT) / FT(qp.mu0))
    added.j0_p[...] = I0N[None, :] * att[:, None]
    added.j0_m[...] = (FT(qp.mu0) * (R_surf @ I0N))[None, :] * att[:, None]
    added.r_mp[...] = (R_surf * (qp.qp_muN.astype(FT) * qp.wt_muN.astype(FT))[None, :])[None]
    added.r_pm[...] = 0
    added.t_pp[...] = np.eye(N, dtype=FT)[None]
    added.t_mm[...] = np.eye(N, dtype=FT)[None]


def create_surface_layer_brdf_lin(rho_m, drho_m, added: O.AddedLayer, al: OL.AddedLayerLin, iparam, m, pol, qp, tau_sum,
                                  tau_sum_dot, F0, FT):
    """coxmunk_surface_lin.jl:27-102 with (rho_m, drho_m) = reflectance_and_deriv(surf, pol, qp_mu, m)."""
    S, N = added.j0_p.shape
    n = pol.n
    f = FT(2) if m == 0 else FT(1)
    R_surf, Rd_surf = (f * rho_m).astype(FT), (f * drho_m).astype(FT)
    i0 = n * qp.imu0
    att = np.exp(-np.asarray(tau_sum, dtype=FT) / FT(qp.mu0))
    F0N = np.zeros((S, N), dtype=FT)
    F0N[:, i0:i0 + n] = (np.asarray(F0, dtype=FT) * att[None, :]).T
    for arr in (al.ap_r_mp, al.ap_r_pm, al.ap_t_pp, al.ap_t_mm, al.ap_J0_p, al.ap_J0_m):
        arr[...] = 0
    added.j0_p[...] = 0
    added.j0_m[...] = FT(qp.mu0) * (F0N @ R_surf.T)
    tsd = np.asarray(tau_sum_dot, dtype=FT)
    for p in range(tsd.shape[1]):
        FdN = -F0N * (tsd[:, p] / FT(qp.mu0))[:, None]
        al.ap_J0_m[p] = FT(qp.mu0) * (FdN @ R_surf.T)
    al.ap_J0_m[iparam] = FT(qp.mu0) * (F0N @ Rd_surf.T)
    sc = (qp.qp_muN.astype(FT) * qp.wt_muN.astype(FT))[None, :]
    added.r_mp[...] = (R_surf * sc)[None]
    added.r_pm[...] = 0
    added.t_pp[...] = np.eye(N, dtype=FT)[None]
    added.t_mm[...] = 0
    al.ap_r_mp[iparam] = (Rd_surf * sc)[None]


# ---- drivers ----------------------------------------------------------------------------------------------
def rt_run(model: O.RTModel, surf: CoxMunkSurface, ss_correction: bool = True):
    """rt_run.jl:238-539 with brdf = CoxMunkSurface: the elastic driver of vsm_oracle.rt_run with the BRDF surface layer
    and the TMS correction (rt_run.jl:520-524).  Returns (R_SFI, T_SFI) [nVZA, nStokes, S]."""
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
    added, added_surf, comp = O.make_added_layer(FT, N, S), O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
    mu = qp.qp_mu.astype(np.float64)
    tau_sum_all = None
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = O.construct_core_optical_properties(model, m)
        ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
        for iz in range(L):
            lo = O.expand_optical_properties(lods[iz], FT)
            O.rt_kernel(pol, added, comp, lo, ifaces[iz], tau_sum_all[:, iz].astype(FT), m, qp, iz + 1, F0, FT, model.numerics)
        create_surface_layer_brdf(reflectance(surf, pol.n, mu, m), added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        O.interaction(ifaces[-1], comp, added_surf, FT)
        O.postprocessing_vza(pol, comp, model.vza, model.vaz, qp, m, weight, R_SFI, T_SFI)
    if ss_correction:
        apply_ss_correction(R_SFI, surf, pol, model.vza, model.vaz, FT(qp.mu0), tau_sum_all[:, -1], model.m_max)
    return R_SFI, T_SFI


def rt_run_lin(model: O.RTModel, lin: OL.LinModel, surf: CoxMunkSurface):
    """rt_run_lin.jl:102-326 with brdf = CoxMunkSurface: P = n_layer_params + 1 (wind speed last).  No TMS correction on
    this path (the linearized driver does not call apply_ss_correction!)."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    pl = lin.n_layer_params
    P = pl + 1
    R, T = np.zeros((nV, pol.n, S), dtype=FT), np.zeros((nV, pol.n, S), dtype=FT)
    Rd, Td = np.zeros((nV, pol.n, S, P), dtype=FT), np.zeros((nV, pol.n, S, P), dtype=FT)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S), dtype=FT)
        F0[0, :] = 1
    added, added_s, comp = O.make_added_layer(FT, N, S), O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
    al, als, cl = OL.make_added_layer_lin(FT, P, N, S), OL.make_added_layer_lin(FT, P, N, S), OL.make_composite_layer_lin(FT, P, N, S)
    mu = qp.qp_mu.astype(np.float64)
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = O.construct_core_optical_properties(model, m)
        lins = OL.layer_optics_lin(model, lin, lods)
        ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
        tsd = np.zeros((S, pl, L + 1))
        for iz in range(L):
            tsd[:, :, iz + 1] = tsd[:, :, iz] + lins[iz].tau_dot
        for iz in range(L):
            lo = O.expand_optical_properties(lods[iz], FT)
            dtau, nd = O.get_dtau_ndoubl(lo.tau, lo.varpi, qp, FT, model.numerics)
            expk = np.exp(-dtau / FT(qp.mu0)).astype(FT)
            OL.elemental_lin(pol, tau_sum_all[:, iz].astype(FT), tsd[:, :, iz], dtau, F0, lo.varpi, lo.Zpp, lo.Zmp, lins[iz], m,
                             nd, qp, added, al, FT)
            dall = np.zeros((S, P), dtype=FT)
            dall[:, :pl] = lins[iz].tau_dot / FT(2 ** nd)
            OL.doubling_lin(pol, expk, nd, added, al, dall, qp.mu0, pl, FT)
            if iz == 0:
                O.copy_added_to_composite(comp, added)
                for a, b in ((cl.T_pp, al.ap_t_pp), (cl.T_mm, al.ap_t_mm), (cl.R_mp, al.ap_r_mp), (cl.R_pm, al.ap_r_pm),
                             (cl.J0_p, al.ap_J0_p), (cl.J0_m, al.ap_J0_m)):
                    a[...] = b
            else:
                OL.interaction_lin(ifaces[iz], comp, cl, added, al, FT)
        rho, drho = reflectance_and_deriv(surf, pol.n, mu, m)
        create_surface_layer_brdf_lin(rho, drho, added_s, als, P - 1, m, pol, qp, tau_sum_all[:, -1], tsd[:, :, -1], F0, FT)
        OL.interaction_lin(ifaces[-1], comp, cl, added_s, als, FT)
        n = pol.n
        for i in range(nV):
            imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(O.cosd(model.vza[i])))))
            c_, s_ = O.cosd(m * model.vaz[i]), O.sind(m * model.vaz[i])
            w = weight * np.array([c_, c_, s_, s_][:n])
            R[i] += w[:, None] * comp.J0_m[:, imu * n:(imu + 1) * n].T
            T[i] += w[:, None] * comp.J0_p[:, imu * n:(imu + 1) * n].T
            for p in range(P):
                Rd[i, :, :, p] += w[:, None] * cl.J0_m[p][:, imu * n:(imu + 1) * n].T
                Td[i, :, :, p] += w[:, None] * cl.J0_p[p][:, imu * n:(imu + 1) * n].T
    return R, T, Rd, Td
