"""CPU ORACLE, linearized (Jacobian) pass -- test infrastructure, NOT the product path.

numpy restatement of vSmartMOM.jl's `rt_run(model, lin_model, NAer, NGas, NSurf)` hot path:
  src/CoreRT/CoreKernel/elemental_lin.jl:77-206, 456-712   (fused elemental + chain rule, SFI + beam term)
  src/CoreRT/CoreKernel/doubling_lin.jl:216-339, 374-421   (doubling_allparams_helper!, apply_D lin)
  src/CoreRT/CoreKernel/interaction_lin.jl:62-331          (interaction_helper! lin, 4 interface cases)
  src/CoreRT/CoreKernel/rt_kernel_lin.jl:50-180            (rt_kernel! lin)
  src/CoreRT/Surfaces/lambertian_surface_lin.jl:48-162     (Lambertian surface + albedo derivative)
  src/CoreRT/tools/postprocessing_vza_lin.jl:18-48
  src/CoreRT/rt_run_lin.jl:102-326                         (driver)

PARITY UNPINNED: the reference holds no committed numbers for Jacobians that can be reproduced without
Julia (its own tests compare analytic vs finite differences inside Julia).  This module is checked the
same way -- analytic vs central finite differences of the pinned forward oracle
(tests/test_oracle_lin.py) -- and then serves as the checker of the HIP linearized path.

Arrays are batch-first like vsm_oracle: derivative stacks are [P, S, N, N] / [P, S, N]
(the reference's [N,N,S,P] column-major with the parameter axis slowest).
Only the parameters the hot path sees are modelled: per layer (tau_dot, varpi_dot, Z_dot) for the first
`n_layer_params` slots, then surface slots (ParameterLayout, parameter_layout.jl:28-56).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import vsm_oracle as O


@dataclass
class LayerOpticsLin:
    """CoreScatteringOpticalPropertiesLin (types_lin.jl:141-150) for one layer."""
    tau_dot: np.ndarray            # [S, p]   p = number of layer parameters
    varpi_dot: np.ndarray          # [S, p]
    Zpp_dot: Optional[np.ndarray]  # [p, N, N] or [p, S, N, N] or None (= 0)
    Zmp_dot: Optional[np.ndarray]


@dataclass
class AddedLayerLin:
    ap_r_mp: np.ndarray  # [P, S, N, N]
    ap_t_pp: np.ndarray
    ap_r_pm: np.ndarray
    ap_t_mm: np.ndarray
    ap_J0_p: np.ndarray  # [P, S, N]
    ap_J0_m: np.ndarray


@dataclass
class CompositeLayerLin:
    R_mp: np.ndarray
    R_pm: np.ndarray
    T_pp: np.ndarray
    T_mm: np.ndarray
    J0_p: np.ndarray
    J0_m: np.ndarray


def make_added_layer_lin(FT, P, N, S) -> AddedLayerLin:
    z = lambda: np.zeros((P, S, N, N), dtype=FT)
    v = lambda: np.zeros((P, S, N), dtype=FT)
    return AddedLayerLin(z(), z(), z(), z(), v(), v())


def make_composite_layer_lin(FT, P, N, S) -> CompositeLayerLin:
    z = lambda: np.zeros((P, S, N, N), dtype=FT)
    v = lambda: np.zeros((P, S, N), dtype=FT)
    return CompositeLayerLin(z(), z(), z(), z(), v(), v())


def _zdot(Zd, p, S, N, FT):
    if Zd is None:
        return np.zeros((1, N, N), dtype=FT)
    Zd = np.asarray(Zd, dtype=FT)
    return Zd[p][None] if Zd.ndim == 3 else Zd[p]


def elemental_lin(pol, tau_sum, tau_sum_dot, dtau, F0, varpi, Zpp, Zmp, lin: LayerOpticsLin, m, ndoubl, qp,
                  added: O.AddedLayer, added_lin: AddedLayerLin, FT):
    """elemental_lin.jl:77-206 with kernels :456-591 (get_elem_rt_fused!) and :602-712 (get_elem_rt_SFI_fused!).
    tau_sum_dot: [S, p].  Forward fields of `added` are filled as by the forward elemental!."""
    O.elemental(pol, tau_sum, dtau, F0, varpi, Zpp, Zmp, m, ndoubl, qp, added, FT)
    mu = qp.qp_muN.astype(FT)
    N, S = len(mu), len(dtau)
    n = pol.n
    p_layer = lin.tau_dot.shape[1]
    wct02 = FT(0.5) if m == 0 else FT(0.25)
    wct = (qp.wt_muN.astype(FT) / FT(2)) if m == 0 else (qp.wt_muN.astype(FT) / FT(4))
    mi, mj = mu[None, :, None], mu[None, None, :]
    wj = wct[None, None, :]
    d = dtau.astype(FT)[:, None, None]
    w = varpi.astype(FT)[:, None, None]
    Zpp_b = np.broadcast_to(np.asarray(Zpp, dtype=FT), (S, N, N))
    Zmp_b = np.broadcast_to(np.asarray(Zmp, dtype=FT), (S, N, N))
    active = (wct > O.eps(FT))[None, None, :]
    eye = np.eye(N, dtype=bool)[None]
    one = FT(1)
    D = O._dsign(pol, N).astype(FT)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        # forward values in un-signed form (needed for the w-derivatives)
        geo_r = (mj / (mi + mj)) * wj * (-np.expm1(-d * ((one / mi) + (one / mj))))
        r_plain = w * Zmp_b * geo_r
        r_tau = w * Zmp_b * (one / mi) * wj * np.exp(-d * ((one / mi) + (one / mj)))
        r_w = np.where(w == 0, FT(0), r_plain / np.where(w == 0, one, w))
        r_Z = w * geo_r
        same_mu = (mi == mj)
        e_i, e_j = np.exp(-d / mi), np.exp(-d / mj)
        # i == j
        t_dd = e_i * (one + w * Zpp_b * (d / mi) * wj)
        tt_dd = e_i * (one / mi) * (-one + w * Zpp_b * wj * (one - d / mi))
        tw_dd = e_i * Zpp_b * (d / mi) * wj
        tZ_dd = e_i * w * (d / mi) * wj
        # mu_i == mu_j, i != j
        t_sm = e_j * (w * Zpp_b * (d / mi) * wj)
        tt_sm = (e_j * w * Zpp_b / mi) * (one - d / mj) * wj
        tZ_sm = e_j * w * (d / mi) * wj
        # general
        geo_t = (mj / (mi - mj)) * wj
        ed = O.expdiff_neg(d / mi, d / mj)
        t_of = w * Zpp_b * geo_t * ed
        tt_of = -w * Zpp_b * geo_t * (e_i / mi - e_j / mj)
        tZ_of = w * geo_t * ed
        t_plain = np.where(same_mu, np.where(eye, t_dd, t_sm), t_of)
        t_tau = np.where(same_mu, np.where(eye, tt_dd, tt_sm), tt_of)
        t_w_gen = np.where(w == 0, FT(0), t_plain / np.where(w == 0, one, w))
        t_w = np.where(same_mu & eye, tw_dd, t_w_gen)
        t_Z = np.where(same_mu, np.where(eye, tZ_dd, tZ_sm), tZ_of)
    # zero-weight columns: only the diagonal Beer term and its tau-derivative
    r_tau, r_w, r_Z = (np.where(active, x, FT(0)) for x in (r_tau, r_w, r_Z))
    t_tau = np.where(active, t_tau, np.where(eye, -np.exp(-d / mi) / mi * np.ones((1, 1, N), dtype=FT), FT(0)))
    t_w = np.where(active, t_w, FT(0))
    t_Z = np.where(active, t_Z, FT(0))
    sign_r = np.where((ndoubl >= 1) & (D < 0), -one, one)[None, :, None]
    dtau_dot = (lin.tau_dot / FT(2 ** ndoubl)).astype(FT)
    par = (D[:, None] * D[None, :])[None]
    for arr in (added_lin.ap_r_mp, added_lin.ap_t_pp, added_lin.ap_r_pm, added_lin.ap_t_mm, added_lin.ap_J0_p,
                added_lin.ap_J0_m):
        arr[...] = 0
    for p in range(p_layer):
        td = dtau_dot[:, p][:, None, None]
        wd = lin.varpi_dot[:, p].astype(FT)[:, None, None]
        Zpd = _zdot(lin.Zpp_dot, p, S, N, FT)
        Zmd = _zdot(lin.Zmp_dot, p, S, N, FT)
        val_r = r_tau * td + r_w * wd + r_Z * Zmd
        val_t = t_tau * td + t_w * wd + t_Z * Zpd
        added_lin.ap_r_mp[p] = sign_r * val_r
        added_lin.ap_t_pp[p] = val_t
        if ndoubl < 1:
            # d_sign * (.. + r_Z * di*dj*Zdot): di*dj == d_sign, so the Z term carries no net sign
            added_lin.ap_r_pm[p] = par * (r_tau * td + r_w * wd) + r_Z * Zmd
            added_lin.ap_t_mm[p] = par * (t_tau * td + t_w * wd) + t_Z * Zpd
    # ---- SFI source + derivatives ------------------------------------------------------------------
    i_start = n * qp.imu0
    i_end = i_start + n
    F0 = np.asarray(F0, dtype=FT)
    ZFp = np.einsum("sik,ks->si", Zpp_b[:, :, i_start:i_end], F0)
    ZFm = np.einsum("sik,ks->si", Zmp_b[:, :, i_start:i_end], F0)
    mu_s = mu[i_start]
    mi1 = mu[None, :]
    d1 = dtau.astype(FT)[:, None]
    w1 = varpi.astype(FT)[:, None]
    in_sun = np.zeros(N, dtype=bool)
    in_sun[i_start:i_end] = True
    att = np.exp(-np.asarray(tau_sum, dtype=FT) / mu_s)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        jp_sun = wct02 * w1 * ZFp * (d1 / mi1) * np.exp(-d1 / mi1)
        jp_off = wct02 * w1 * ZFp * (mu_s / (mi1 - mu_s)) * O.expdiff_neg(d1 / mi1, d1 / mu_s)
        jp = np.where(in_sun[None, :], jp_sun, jp_off)
        jp_tau_sun = jp_sun * (one / d1 - one / mi1)
        jp_tau_off = -wct02 * w1 * ZFp * (mu_s / (mi1 - mu_s)) * (np.exp(-d1 / mi1) / mi1 - np.exp(-d1 / mu_s) / mu_s)
        jp_tau = np.where(in_sun[None, :], jp_tau_sun, jp_tau_off)
        jp_w = np.where(w1 == 0, FT(0), jp / np.where(w1 == 0, one, w1))
        jp_Z = np.where(ZFp == 0, FT(0), jp / np.where(ZFp == 0, one, ZFp))
        gm = (mu_s / (mi1 + mu_s))
        arg = d1 * ((one / mi1) + (one / mu_s))
        jm = wct02 * w1 * ZFm * gm * (-np.expm1(-arg))
        jm_tau = wct02 * w1 * ZFm * gm * np.exp(-arg) * ((one / mi1) + (one / mu_s))
        jm_w = np.where(w1 == 0, FT(0), jm / np.where(w1 == 0, one, w1))
        jm_Z = np.where(ZFm == 0, FT(0), jm / np.where(ZFm == 0, one, ZFm))
    jp, jm = jp * att, jm * att
    jp_tau, jp_w, jp_Z = jp_tau * att, jp_w * att, jp_Z * att
    jm_tau, jm_w, jm_Z = jm_tau * att, jm_w * att, jm_Z * att
    if ndoubl >= 1:
        jm, jm_tau, jm_w, jm_Z = (x * D[None, :] for x in (jm, jm_tau, jm_w, jm_Z))
    for p in range(p_layer):
        td = dtau_dot[:, p][:, None]
        wd = lin.varpi_dot[:, p].astype(FT)[:, None]
        Zpd = np.broadcast_to(_zdot(lin.Zpp_dot, p, S, N, FT), (S, N, N))
        Zmd = np.broadcast_to(_zdot(lin.Zmp_dot, p, S, N, FT), (S, N, N))
        ZdFp = np.einsum("sik,ks->si", Zpd[:, :, i_start:i_end], F0)
        ZdFm = np.einsum("sik,ks->si", Zmd[:, :, i_start:i_end], F0)
        beam = (-np.asarray(tau_sum_dot, dtype=FT)[:, p] / mu_s)[:, None]
        added_lin.ap_J0_p[p] = jp_tau * td + jp_w * wd + jp_Z * ZdFp + jp * beam
        added_lin.ap_J0_m[p] = jm_tau * td + jm_w * wd + jm_Z * ZdFm + jm * beam
    # (forward j0 of `added` was already written -- with the same D convention -- by O.elemental)


def doubling_lin(pol, expk, ndoubl, added: O.AddedLayer, added_lin: AddedLayerLin, dtau_dot_all, mu0, N_active, FT):
    """doubling_lin.jl:216-339 (doubling_allparams_helper!) + apply_D lin :374-421."""
    if ndoubl == 0:
        return
    r, t = added.r_mp.copy(), added.t_pp.copy()
    jp, jm = added.j0_p.copy(), added.j0_m.copy()
    ar, at = added_lin.ap_r_mp, added_lin.ap_t_pp
    aJp, aJm = added_lin.ap_J0_p, added_lin.ap_J0_m
    P_all = ar.shape[0]
    Np = N_active if N_active > 0 else P_all
    N = r.shape[1]
    I = np.eye(N, dtype=FT)[None]
    expk = expk.astype(FT).copy()
    ek_lin = np.stack([-expk / FT(mu0) * dtau_dot_all[:, p].astype(FT) for p in range(Np)]) if Np else np.zeros((0, len(expk)), dtype=FT)
    mv = O._mv
    for _ in range(ndoubl):
        G = np.linalg.inv(I - r @ r).astype(FT)
        tt = t @ G
        G_lin = [G @ (ar[p] @ r + r @ ar[p]) @ G for p in range(Np)]
        tt_lin = [at[p] @ G + t @ G_lin[p] for p in range(Np)]
        J1p, J1m = jp * expk[:, None], jm * expk[:, None]
        A = J1m + mv(r, jp)
        B = jp + mv(r, J1m)
        for p in range(Np):
            aJ1p = aJp[p] * expk[:, None] + jp * ek_lin[p][:, None]
            aJ1m = aJm[p] * expk[:, None] + jm * ek_lin[p][:, None]
            ek_lin[p] = 2 * expk * ek_lin[p]
            new_m = aJm[p] + mv(tt_lin[p], A) + mv(tt, aJ1m + mv(ar[p], jp) + mv(r, aJp[p]))
            new_p = aJ1p + mv(tt_lin[p], B) + mv(tt, aJp[p] + mv(ar[p], J1m) + mv(r, aJ1m))
            aJm[p], aJp[p] = new_m, new_p
        jm_new = jm + mv(tt, A)
        jp_new = J1p + mv(tt, B)
        jm, jp = jm_new, jp_new
        expk = expk ** 2
        rt = r @ t
        for p in range(Np):
            ar_new = ar[p] + tt_lin[p] @ rt + tt @ (ar[p] @ t + r @ at[p])
            at[p] = tt_lin[p] @ t + tt @ at[p]
            ar[p] = ar_new
        r = r + tt @ rt
        t = tt @ t
    D = O._dsign(pol, N).astype(FT)
    par = (D[:, None] * D[None, :])[None]
    if pol.n == 1:
        added.r_mp[...], added.t_pp[...], added.r_pm[...], added.t_mm[...] = r, t, r, t
        added.j0_p[...], added.j0_m[...] = jp, jm
        added_lin.ap_r_pm[...] = ar
        added_lin.ap_t_mm[...] = at
        return
    r = r * D[None, :, None]
    added.r_mp[...], added.t_pp[...] = r, t
    added.r_pm[...], added.t_mm[...] = par * r, par * t
    added.j0_p[...], added.j0_m[...] = jp, jm * D[None, :]
    ar *= D[None, None, :, None]                     # all parameter slots (size(ṙ⁻⁺,4))
    added_lin.ap_r_pm[...] = par[None] * ar
    added_lin.ap_t_mm[...] = par[None] * at
    aJm *= D[None, None, :]


def interaction_lin(iface, comp: O.CompositeLayer, cl: CompositeLayerLin, add: O.AddedLayer, al: AddedLayerLin, FT):
    """interaction_lin.jl:62-331."""
    mv = O._mv
    r_mp, r_pm, t_pp, t_mm, j0_p, j0_m = add.r_mp, add.r_pm, add.t_pp, add.t_mm, add.j0_p, add.j0_m
    P = cl.T_mm.shape[0]
    N = r_mp.shape[1]
    I = np.eye(N, dtype=FT)[None]
    c = comp
    if iface == "00":
        for p in range(P):
            cl.J0_p[p] = al.ap_J0_p[p] + mv(t_pp, cl.J0_p[p]) + mv(al.ap_t_pp[p], c.J0_p)
            cl.J0_m[p] = cl.J0_m[p] + mv(c.T_mm, al.ap_J0_m[p]) + mv(cl.T_mm[p], j0_m)
        c.J0_p[...] = j0_p + mv(t_pp, c.J0_p)
        c.J0_m[...] = c.J0_m + mv(c.T_mm, j0_m)
        for p in range(P):
            cl.T_mm[p] = al.ap_t_mm[p] @ c.T_mm + t_mm @ cl.T_mm[p]
            cl.T_pp[p] = al.ap_t_pp[p] @ c.T_pp + t_pp @ cl.T_pp[p]
        c.T_mm[...] = t_mm @ c.T_mm
        c.T_pp[...] = t_pp @ c.T_pp
    elif iface == "01":
        for p in range(P):
            cl.J0_m[p] = (cl.J0_m[p] + mv(cl.T_mm[p], mv(r_mp, c.J0_p) + j0_m) +
                          mv(c.T_mm, mv(al.ap_r_mp[p], c.J0_p) + mv(r_mp, cl.J0_p[p]) + al.ap_J0_m[p]))
            cl.J0_p[p] = al.ap_J0_p[p] + mv(al.ap_t_pp[p], c.J0_p) + mv(t_pp, cl.J0_p[p])
        c.J0_m[...] = c.J0_m + mv(c.T_mm, mv(r_mp, c.J0_p) + j0_m)
        c.J0_p[...] = j0_p + mv(t_pp, c.J0_p)
        for p in range(P):
            cl.R_mp[p] = cl.T_mm[p] @ r_mp @ c.T_pp + c.T_mm @ al.ap_r_mp[p] @ c.T_pp + c.T_mm @ r_mp @ cl.T_pp[p]
            cl.R_pm[p] = al.ap_r_pm[p]
            cl.T_pp[p] = al.ap_t_pp[p] @ c.T_pp + t_pp @ cl.T_pp[p]
            cl.T_mm[p] = cl.T_mm[p] @ t_mm + c.T_mm @ al.ap_t_mm[p]
        c.R_mp[...] = c.T_mm @ r_mp @ c.T_pp
        c.R_pm[...] = r_pm
        c.T_pp[...] = t_pp @ c.T_pp
        c.T_mm[...] = c.T_mm @ t_mm
    elif iface == "10":
        for p in range(P):
            cl.J0_p[p] = (al.ap_J0_p[p] + mv(al.ap_t_pp[p], c.J0_p + mv(c.R_pm, j0_m)) +
                          mv(t_pp, cl.J0_p[p] + mv(cl.R_pm[p], j0_m) + mv(c.R_pm, al.ap_J0_m[p])))
            cl.J0_m[p] = cl.J0_m[p] + mv(cl.T_mm[p], j0_m) + mv(c.T_mm, al.ap_J0_m[p])
        c.J0_p[...] = j0_p + mv(t_pp, c.J0_p + mv(c.R_pm, j0_m))
        c.J0_m[...] = c.J0_m + mv(c.T_mm, j0_m)
        for p in range(P):
            cl.T_pp[p] = al.ap_t_pp[p] @ c.T_pp + t_pp @ cl.T_pp[p]
            cl.T_mm[p] = cl.T_mm[p] @ t_mm + c.T_mm @ al.ap_t_mm[p]
            cl.R_pm[p] = al.ap_t_pp[p] @ c.R_pm @ t_mm + t_pp @ cl.R_pm[p] @ t_mm + t_pp @ c.R_pm @ al.ap_t_mm[p]
        c.T_pp[...] = t_pp @ c.T_pp
        c.T_mm[...] = c.T_mm @ t_mm
        c.R_pm[...] = t_pp @ c.R_pm @ t_mm
    elif iface == "11":
        G1 = np.linalg.inv(I - r_mp @ c.R_pm).astype(FT)
        T01 = c.T_mm @ G1
        rT = r_mp @ c.T_pp
        T01_lin, nR_mp, nT_mm = [], [], []
        for p in range(P):
            G1l = G1 @ (al.ap_r_mp[p] @ c.R_pm + r_mp @ cl.R_pm[p]) @ G1
            T01l = cl.T_mm[p] @ G1 + c.T_mm @ G1l
            T01_lin.append(T01l)
            nR_mp.append(cl.R_mp[p] + T01l @ rT + T01 @ (al.ap_r_mp[p] @ c.T_pp + r_mp @ cl.T_pp[p]))
            nT_mm.append(T01l @ t_mm + T01 @ al.ap_t_mm[p])
        A = mv(r_mp, c.J0_p) + j0_m
        nJ0_m = c.J0_m + mv(T01, A)
        nJ0_m_lin = [cl.J0_m[p] + mv(T01_lin[p], A) + mv(T01, mv(al.ap_r_mp[p], c.J0_p) + mv(r_mp, cl.J0_p[p]) + al.ap_J0_m[p])
                     for p in range(P)]
        nR_mp0 = c.R_mp + T01 @ rT
        nT_mm0 = T01 @ t_mm
        G2 = np.linalg.inv(I - c.R_pm @ r_mp).astype(FT)
        T21 = t_pp @ G2
        Rt = c.R_pm @ t_mm
        nT_pp, nR_pm, nJ0_p_lin = [], [], []
        B = c.J0_p + mv(c.R_pm, j0_m)
        for p in range(P):
            G2l = G2 @ (c.R_pm @ al.ap_r_mp[p] + cl.R_pm[p] @ r_mp) @ G2
            T21l = al.ap_t_pp[p] @ G2 + t_pp @ G2l
            nT_pp.append(T21l @ c.T_pp + T21 @ cl.T_pp[p])
            nR_pm.append(al.ap_r_pm[p] + T21l @ Rt + T21 @ (cl.R_pm[p] @ t_mm + c.R_pm @ al.ap_t_mm[p]))
            nJ0_p_lin.append(al.ap_J0_p[p] + mv(T21l, B) + mv(T21, cl.J0_p[p] + mv(cl.R_pm[p], j0_m) + mv(c.R_pm, al.ap_J0_m[p])))
        nJ0_p = j0_p + mv(T21, B)
        nT_pp0 = T21 @ c.T_pp
        nR_pm0 = r_pm + T21 @ Rt
        c.J0_p[...], c.J0_m[...] = nJ0_p, nJ0_m
        c.R_pm[...], c.T_mm[...], c.R_mp[...], c.T_pp[...] = nR_pm0, nT_mm0, nR_mp0, nT_pp0
        for p in range(P):
            cl.J0_p[p], cl.J0_m[p] = nJ0_p_lin[p], nJ0_m_lin[p]
            cl.R_pm[p], cl.T_mm[p], cl.R_mp[p], cl.T_pp[p] = nR_pm[p], nT_mm[p], nR_mp[p], nT_pp[p]
    else:
        raise ValueError(iface)


def create_surface_layer_lambertian_lin(albedo, added: O.AddedLayer, al: AddedLayerLin, iparam, m, pol, qp, tau_sum,
                                        tau_sum_dot, F0, FT):
    """lambertian_surface_lin.jl:48-162.  NOTE (reference quirk): unlike the forward builder this one sets
    j0+ = 0 and t-- = 0, and uses F0 (not pol.I0) for the direct beam."""
    N, S = added.r_mp.shape[1], added.r_mp.shape[0]
    n = pol.n
    I = np.eye(N, dtype=FT)
    for arr in (al.ap_r_mp, al.ap_r_pm, al.ap_t_pp, al.ap_t_mm, al.ap_J0_p, al.ap_J0_m):
        arr[...] = 0
    if m == 0:
        rho = FT(2) * FT(albedo)
        R_surf = np.zeros((N, N), dtype=FT)
        R_surf[0::n, 0::n] = rho
        Rd_surf = np.zeros((N, N), dtype=FT)
        Rd_surf[0::n, 0::n] = FT(2)
        i0 = n * qp.imu0
        att = np.exp(-np.asarray(tau_sum, dtype=FT) / FT(qp.mu0))
        F0N = np.zeros((S, N), dtype=FT)
        F0N[:, i0:i0 + n] = (np.asarray(F0, dtype=FT) * att[None, :]).T
        nparams = tau_sum_dot.shape[1]
        added.j0_p[...] = 0
        added.j0_m[...] = FT(qp.mu0) * (F0N @ R_surf.T)
        for p in range(nparams):
            FdN = -F0N * (np.asarray(tau_sum_dot, dtype=FT)[:, p] / FT(qp.mu0))[:, None]
            al.ap_J0_m[p] = FT(qp.mu0) * (FdN @ R_surf.T)
        al.ap_J0_m[iparam] = FT(qp.mu0) * (F0N @ Rd_surf.T)
        sc = (qp.qp_muN.astype(FT) * qp.wt_muN.astype(FT))[None, :]
        added.r_mp[...] = (R_surf * sc)[None]
        added.r_pm[...] = 0
        added.t_pp[...] = I[None]
        added.t_mm[...] = 0
        al.ap_r_mp[iparam] = (Rd_surf * sc)[None]
    else:
        added.r_mp[...] = 0
        added.t_pp[...] = I[None]
        added.t_mm[...] = 0
        added.j0_p[...] = 0
        added.j0_m[...] = 0


@dataclass
class LinAerosolOptics:
    """lin_aerosol_optics[iaer] as the hot path consumes it (compEffectiveLayerProperties_lin.jl:330-395): derivatives of the
    aerosol's Greek coefficients, single-scattering albedo and truncation factor with respect to the four Mie parameters
    (n_r, n_i, r_m, sigma_r).  Mie theory itself (the producer of these numbers) is upstream of the hot path."""
    greek_dot: List[O.GreekCoefs]   # 4 entries
    ssa_dot: np.ndarray             # [4]
    f_trunc_dot: np.ndarray         # [4]


@dataclass
class LinModel:
    """What the hot path needs from `lin_model`: tau_abs_dot[g][S, L] = d tau_abs / d x_g per layer (NGas slots); for aerosol
    Jacobians tau_aer_dot[iaer][7, L] = d tau_aer / d (tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p) per layer and
    lin_aerosol_optics[iaer]; plus one surface slot (NSurf = 1).  Slot order (parameter_layout.jl:28-56): 7 per aerosol,
    then the gases, then the surface."""
    tau_abs_dot: List[np.ndarray]
    tau_aer_dot: Optional[np.ndarray] = None            # [nAer, 7, L]
    lin_aerosol_optics: Optional[List[LinAerosolOptics]] = None

    @property
    def n_aer(self):
        return 0 if self.tau_aer_dot is None else len(self.tau_aer_dot)

    @property
    def n_layer_params(self):
        return 7 * self.n_aer + len(self.tau_abs_dot)


def create_aero_lin(tau_aer, ao: O.AerosolOptics, tau_aer_dot7, lao: LinAerosolOptics, AZpp_dot4, AZmp_dot4):
    """createAero with derivatives (compEffectiveLayerProperties_lin.jl:330-395): delta-M scaled (tau, varpi) of one aerosol in
    one layer and their derivatives with respect to its 7 sub-parameters; Zdot is non-zero for the Mie slots 2..5 only."""
    f, w = ao.f_trunc, ao.ssa
    wd = np.zeros(7)
    fd = np.zeros(7)
    wd[1:5] = lao.ssa_dot
    fd[1:5] = lao.f_trunc_dot
    tau_mod = (1.0 - f * w) * tau_aer
    varpi_mod = (1.0 - f) * w / (1.0 - f * w)
    td = np.zeros(7)
    vd = np.zeros(7)
    td[0] = (1.0 - f * w) * tau_aer_dot7[0]
    for k in range(1, 5):
        td[k] = (1.0 - f * w) * tau_aer_dot7[k] - (f * wd[k] + w * fd[k]) * tau_aer
        vd[k] = (wd[k] * (1.0 - f) - fd[k] * (w * (1.0 - w))) / (1.0 - f * w) ** 2
    for k in (5, 6):
        td[k] = (1.0 - f * w) * tau_aer_dot7[k]
    N = AZpp_dot4.shape[-1]
    Zpd = np.zeros((7, N, N))
    Zmd = np.zeros((7, N, N))
    Zpd[1:5] = AZpp_dot4
    Zmd[1:5] = AZmp_dot4
    return tau_mod, varpi_mod, td, vd, Zpd, Zmd


def _b(a, S):
    return np.broadcast_to(np.asarray(a, dtype=np.float64), (S,)).copy()


def _zs(Z, S):
    Z = np.asarray(Z, dtype=np.float64)
    return np.broadcast_to(Z, (S,) + Z.shape[-2:]) if Z.ndim == 2 else Z


def _mix_lin(x, xd, y, yd, S):
    """`+` of two scattering layers with derivatives (types_lin.jl:196-298): tau = tau_x + tau_y, varpi = (w_x + w_y) / tau with
    w = tau varpi, Z = (w_x Z_x + w_y Z_y) / (w_x + w_y); the derivative blocks of x and y are concatenated (x first).
    x / y: (tau [S], varpi [S], Zpp [S,N,N], Zmp); xd / yd: None or (tau_dot [S,n], varpi_dot [S,n], Zpp_dot [n,S,N,N], Zmp_dot)."""
    tx, vx, Zxp, Zxm = x
    ty, vy, Zyp, Zym = y
    tau = tx + ty
    wx, wy = tx * vx, ty * vy
    w = wx + wy
    varpi = w / tau
    Zpp = (wx[:, None, None] * Zxp + wy[:, None, None] * Zyp) / w[:, None, None]
    Zmp = (wx[:, None, None] * Zxm + wy[:, None, None] * Zym) / w[:, None, None]
    tds, wds, nums_p, nums_m = [], [], [], []
    for (t_, v_, Zp_, Zm_), d in (((tx, vx, Zxp, Zxm), xd), ((ty, vy, Zyp, Zym), yd)):
        if d is None:
            continue
        td, vd, Zpd, Zmd = d
        wdot = td * v_[:, None] + t_[:, None] * vd                       # d(tau varpi) of this constituent  [S, n]
        tds.append(td)
        wds.append(wdot)
        nums_p.append(wdot.T[:, :, None, None] * Zp_[None] + (t_ * v_)[None, :, None, None] * Zpd)
        nums_m.append(wdot.T[:, :, None, None] * Zm_[None] + (t_ * v_)[None, :, None, None] * Zmd)
    tau_dot = np.concatenate(tds, axis=1)
    w_dot = np.concatenate(wds, axis=1)
    varpi_dot = (w_dot - varpi[:, None] * tau_dot) / tau[:, None]
    tot = tau[:, None] * varpi_dot + tau_dot * varpi[:, None]            # = d(tau varpi) of the mixture
    Zpp_dot = (np.concatenate(nums_p, axis=0) - tot.T[:, :, None, None] * Zpp[None]) / w[None, :, None, None]
    Zmp_dot = (np.concatenate(nums_m, axis=0) - tot.T[:, :, None, None] * Zmp[None]) / w[None, :, None, None]
    return (tau, varpi, Zpp, Zmp), (tau_dot, varpi_dot, Zpp_dot, Zmp_dot)


def _add_absorption_lin(x, xd, tau_abs, tau_abs_dot):
    """`+` of a scattering layer and an absorber with derivatives (types_lin.jl:300-380): tau += tau_abs, varpi = w_x / tau,
    Z unchanged; gas slots are appended after the slots of x."""
    tx, vx, Zp, Zm = x
    tau = tx + tau_abs
    w = tx * vx
    varpi = w / tau
    n2 = tau_abs_dot.shape[1]
    if xd is None:
        return (tau, varpi, Zp, Zm), (tau_abs_dot, -(varpi / tau)[:, None] * tau_abs_dot, None, None)
    td, vd, Zpd, Zmd = xd
    wdot = td * vx[:, None] + tx[:, None] * vd
    tau_dot = np.concatenate([td, tau_abs_dot], axis=1)
    varpi_dot = np.concatenate([(wdot - varpi[:, None] * td) / tau[:, None], -(varpi / tau)[:, None] * tau_abs_dot], axis=1)
    tot = tau[:, None] * varpi_dot + tau_dot * varpi[:, None]
    S, N = len(tau), Zp.shape[-1]
    num_p = np.concatenate([wdot.T[:, :, None, None] * Zp[None] + w[None, :, None, None] * Zpd, np.zeros((n2, S, N, N))], axis=0)
    num_m = np.concatenate([wdot.T[:, :, None, None] * Zm[None] + w[None, :, None, None] * Zmd, np.zeros((n2, S, N, N))], axis=0)
    Zpp_dot = (num_p - tot.T[:, :, None, None] * Zp[None]) / w[None, :, None, None]
    Zmp_dot = (num_m - tot.T[:, :, None, None] * Zm[None]) / w[None, :, None, None]
    return (tau, varpi, Zp, Zm), (tau_dot, varpi_dot, Zpp_dot, Zmp_dot)


def layer_optics_lin(model: O.RTModel, lin: LinModel, lods: List[O.CoreScatteringOpticalProperties], m: int = 0) -> List[LayerOpticsLin]:
    """d(tau, varpi, Z)/dx per layer (constructCoreOpticalProperties with lin_model, compEffectiveLayerProperties_lin.jl:43-197).
    Without aerosol slots: tau = tau_s + tau_abs, varpi = w_s/tau (types.jl:1302-1308) => tau_dot = tau_abs_dot,
    varpi_dot = -varpi/tau * tau_dot, Z_dot = 0.  With aerosol slots the Rayleigh layer (no derivatives) is mixed with every
    aerosol (7 slots each) and the absorbers are added last, each `+` propagating the quotient rule."""
    out = []
    if lin.n_aer == 0:
        for iz, lo in enumerate(lods):
            tau = np.atleast_1d(lo.tau)
            varpi = np.broadcast_to(np.asarray(lo.varpi), tau.shape)
            td = np.stack([g[:, iz] for g in lin.tau_abs_dot], axis=1)
            safe = np.where(tau > 0, tau, 1.0)
            wd = -(varpi / safe)[:, None] * td
            out.append(LayerOpticsLin(td, wd, None, None))
        return out
    assert lin.n_aer == len(model.aerosol_optics), "one derivative block per aerosol of the model"
    mu = model.quad_points.qp_mu.astype(np.float64)
    S, L = model.tau_rayl.shape
    RZpp, RZmp = O.compute_Z_moments(model.pol, mu, model.greek_rayleigh, m)
    aer = []
    for ia, ao in enumerate(model.aerosol_optics):
        AZpp, AZmp = O.compute_Z_moments(model.pol, mu, ao.greek, m)
        Zd = [O.compute_Z_moments(model.pol, mu, g, m) for g in lin.lin_aerosol_optics[ia].greek_dot]   # Z is linear in the Greek coefficients
        aer.append((AZpp, AZmp, np.stack([z[0] for z in Zd]), np.stack([z[1] for z in Zd])))
    for iz in range(L):
        x = (model.tau_rayl[:, iz].astype(np.float64), _b(model.varpi_cabannes, S), _zs(RZpp, S), _zs(RZmp, S))
        xd = None
        for ia, ao in enumerate(model.aerosol_optics):
            AZpp, AZmp, AZpd, AZmd = aer[ia]
            tm, vm, td7, vd7, Zpd7, Zmd7 = create_aero_lin(model.tau_aer[ia, iz], ao, lin.tau_aer_dot[ia][:, iz],
                                                           lin.lin_aerosol_optics[ia], AZpd, AZmd)
            y = (_b(tm, S), _b(vm, S), _zs(AZpp, S), _zs(AZmp, S))
            yd = (np.tile(td7, (S, 1)), np.tile(vd7, (S, 1)), np.broadcast_to(Zpd7[:, None], (7, S) + Zpd7.shape[-2:]),
                  np.broadcast_to(Zmd7[:, None], (7, S) + Zmd7.shape[-2:]))
            x, xd = _mix_lin(x, xd, y, yd, S)
        gd = (np.stack([g[:, iz] for g in lin.tau_abs_dot], axis=1) if lin.tau_abs_dot else np.zeros((S, 0)))
        x, xd = _add_absorption_lin(x, xd, model.tau_abs[:, iz].astype(np.float64), gd)
        # the forward part must reproduce the forward construction
        assert np.allclose(x[0], np.atleast_1d(lods[iz].tau), rtol=1e-13, atol=0) and np.allclose(x[1], lods[iz].varpi, rtol=1e-12, atol=0)
        out.append(LayerOpticsLin(xd[0], xd[1], xd[2], xd[3]))
    return out


def rt_run_lin(model: O.RTModel, lin: LinModel):
    """rt_run_lin.jl:102-326 (noRS, SFI, Lambertian, all layers scattering).  Returns (R, T, Rdot, Tdot) with
    Rdot/Tdot [nVZA, nStokes, S, P], P = n_layer_params + 1 (albedo last)."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    pl = lin.n_layer_params
    P = pl + 1
    R = np.zeros((nV, pol.n, S), dtype=FT)
    T = np.zeros((nV, pol.n, S), dtype=FT)
    Rd = np.zeros((nV, pol.n, S, P), dtype=FT)
    Td = np.zeros((nV, pol.n, S, P), dtype=FT)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S), dtype=FT)
        F0[0, :] = 1
    added, added_s, comp = O.make_added_layer(FT, N, S), O.make_added_layer(FT, N, S), O.make_composite_layer(FT, N, S)
    al, als, cl = make_added_layer_lin(FT, P, N, S), make_added_layer_lin(FT, P, N, S), make_composite_layer_lin(FT, P, N, S)
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = O.construct_core_optical_properties(model, m)
        lins = layer_optics_lin(model, lin, lods, m)
        ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
        tsd = np.zeros((S, pl, L + 1))
        for iz in range(L):
            tsd[:, :, iz + 1] = tsd[:, :, iz] + lins[iz].tau_dot
        for iz in range(L):
            lo = O.expand_optical_properties(lods[iz], FT)
            dtau, nd = O.get_dtau_ndoubl(lo.tau, lo.varpi, qp, FT, model.numerics)
            expk = np.exp(-dtau / FT(qp.mu0)).astype(FT)
            elemental_lin(pol, tau_sum_all[:, iz].astype(FT), tsd[:, :, iz], dtau, F0, lo.varpi, lo.Zpp, lo.Zmp,
                          lins[iz], m, nd, qp, added, al, FT)
            dall = np.zeros((S, P), dtype=FT)
            dall[:, :pl] = lins[iz].tau_dot / FT(2 ** nd)
            doubling_lin(pol, expk, nd, added, al, dall, qp.mu0, pl, FT)
            if iz == 0:
                O.copy_added_to_composite(comp, added)
                for a, b in ((cl.T_pp, al.ap_t_pp), (cl.T_mm, al.ap_t_mm), (cl.R_mp, al.ap_r_mp), (cl.R_pm, al.ap_r_pm),
                             (cl.J0_p, al.ap_J0_p), (cl.J0_m, al.ap_J0_m)):
                    a[...] = b
            else:
                interaction_lin(ifaces[iz], comp, cl, added, al, FT)
        create_surface_layer_lambertian_lin(model.albedo, added_s, als, P - 1, m, pol, qp, tau_sum_all[:, -1],
                                            tsd[:, :, -1], F0, FT)
        interaction_lin(ifaces[-1], comp, cl, added_s, als, FT)
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
