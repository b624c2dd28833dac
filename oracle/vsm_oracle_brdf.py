"""CPU ORACLE, kernel-driven BRDF land surfaces (RPV, Ross-Li) -- test infrastructure, NOT the product path.

numpy restatement of
  src/CoreRT/Surfaces/rpv_surface.jl:99-150     reflectance(::rpvSurfaceScalar, n, mu_i, mu_r, dphi) with rpvM / rpvH / rpvF
  src/CoreRT/Surfaces/rossli_surface.jl:1-98    reflectance(::RossLiSurfaceScalar, ...) with the RossThick / LiSparse kernels
  src/CoreRT/Surfaces/rpv_surface.jl:160-190    reflectance(brdf, pol_type, mu, m): Fourier moment by 100-point Gauss-Legendre
                                                quadrature over dphi in [0, pi]
  src/CoreRT/Surfaces/rpv_surface.jl:51-97      create_surface_layer!(::AbstractSurfaceType) (= vsm_oracle_coxmunk.create_surface_layer_brdf)
and the elastic driver with such a surface (rt_run.jl:238-539; no TMS term: rt_run.jl:520 gates it on CoxMunkSurface).

Pinned by exact limits that reduce the BRDF to the golden-pinned Lambertian path (tests/test_oracle_brdf.py):
rpvSurfaceScalar(rho0, rho_c = 1, k = 1, Theta = 0) and RossLiSurfaceScalar(0, 0, fiso) are Lambertian surfaces.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import vsm_oracle as O
from .vsm_oracle_coxmunk import create_surface_layer_brdf


@dataclass
class RPVSurface:
    """rpvSurfaceScalar (types.jl:482-491)."""
    rho0: float
    rho_c: float
    k: float
    Theta: float


@dataclass
class RossLiSurface:
    """RossLiSurfaceScalar (types.jl:505-512): field order fvol, fgeo, fiso."""
    fvol: float
    fgeo: float
    fiso: float


def brdf_value(surf, n: int, mu_i, mu_r, dphi):
    """reflectance(surf, n, mu_i, mu_r, dphi): the I -> I element (n == 1, 1-based Stokes index); zero for n > 1."""
    mu_i, mu_r = np.asarray(mu_i, dtype=np.float64), np.asarray(mu_r, dtype=np.float64)
    if n != 1:
        return np.zeros(np.broadcast(mu_i, mu_r).shape)
    th_i, th_r = np.arccos(mu_i), np.arccos(mu_r)
    if isinstance(surf, RPVSurface):
        cosg = -mu_i * mu_r + np.sin(th_i) * np.sin(th_r) * math.cos(dphi)
        G = np.sqrt(np.tan(th_i) ** 2 + np.tan(th_r) ** 2 + 2 * np.tan(th_i) * np.tan(th_r) * math.cos(dphi))
        M = (mu_i * mu_r) ** (surf.k - 1) / (mu_i + mu_r) ** (1 - surf.k)
        th = -surf.Theta                                             # (rpvF flips the sign: "for RAMI only")
        F = (1 - th ** 2) / (1 + th ** 2 + 2 * th * cosg) ** 1.5
        Hh = 1 + (1 - surf.rho_c) / (1 + G)
        return surf.rho0 * M * F * Hh
    if isinstance(surf, RossLiSurface):
        d = math.pi - dphi
        xi = np.arccos(np.clip(np.cos(th_i) * np.cos(th_r) + np.sin(th_i) * np.sin(th_r) * math.cos(d), -1.0, 1.0))
        K_vol = ((math.pi / 2 - xi) * np.cos(xi) + np.sin(xi)) / (np.cos(th_i) + np.cos(th_r)) - math.pi / 4
        tip, trp = np.arctan(np.tan(th_i) * 1.0), np.arctan(np.tan(th_r) * 1.0)     # b/r = 1
        xip = np.arccos(np.clip(np.cos(tip) * np.cos(trp) + np.sin(tip) * np.sin(trp) * math.cos(d), -1.0, 1.0))
        sec_sum = 1 / np.cos(tip) + 1 / np.cos(trp)
        D = np.sqrt(np.maximum(np.tan(tip) ** 2 + np.tan(trp) ** 2 - 2 * np.tan(tip) * np.tan(trp) * math.cos(d), 0.0))
        ct = np.clip(2.0 * np.sqrt(D ** 2 + (np.tan(tip) * np.tan(trp) * math.sin(d)) ** 2) / sec_sum, -1.0, 1.0)   # h/b = 2
        t = np.arccos(ct)
        K_geo = (1 / math.pi) * (t - np.sin(t) * np.cos(t)) * sec_sum - sec_sum + 0.5 * (1 + np.cos(xip)) / (np.cos(tip) * np.cos(trp))
        return surf.fiso * 1.0 + surf.fvol * K_vol + surf.fgeo * K_geo
    raise TypeError(surf)


def reflectance(surf, n_stokes: int, mu, m: int, nquad: int = 100):
    """reflectance(brdf, pol_type, mu, m) (rpv_surface.jl:160-190): R[n::ns, n::ns] = ff / pi * int_0^pi rho(n, mu_i, mu_j, x) cos(m x) dx,
    ff = 1 (m = 0) or 2; create_surface_layer! doubles the m = 0 block."""
    mu = np.asarray(mu, dtype=np.float64)
    x, w = np.polynomial.legendre.leggauss(nquad)
    phi, wphi = 0.5 * math.pi * (x + 1.0), 0.5 * math.pi * w
    nn = len(mu) * n_stokes
    R = np.zeros((nn, nn))
    ff = 1.0 if m == 0 else 2.0
    for n in range(1, n_stokes + 1):
        c = np.zeros((len(mu), len(mu)))
        for p, wp in zip(phi, wphi):
            c += wp * brdf_value(surf, n, mu[:, None], mu[None, :], p) * math.cos(m * p)
        R[n - 1::n_stokes, n - 1::n_stokes] = c / math.pi
    return ff * R


def rt_run(model: O.RTModel, surf):
    """rt_run.jl:238-539 with brdf = rpvSurfaceScalar / RossLiSurfaceScalar.  Returns (R_SFI, T_SFI) [nVZA, nStokes, S]."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    R_SFI, T_SFI = np.zeros((nV, pol.n, S), dtype=FT), np.zeros((nV, pol.n, S), dtype=FT)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S), dtype=FT)
        F0[0, :] = 1
    added, added_surf, comp = O.make_added_layer(FT, N, S), O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
    mu = qp.qp_mu.astype(np.float64)
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
    return R_SFI, T_SFI
