"""CPU ORACLE (test infrastructure -- NOT the product path): rotational-Raman (RRS) pass.

A numpy restatement of the inelastic branch of vSmartMOM.jl's CoreRT hot path
(SURVEY.md 8 row a12): `rt_kernel!(::RRS)` = elemental_inelastic! + elemental! +
doubling_inelastic! + (copy | interaction_helper!(::RRS, ::ScatteringInterface_11)),
the Lambertian surface step and the Raman VZA post-processing.  Every function
cites the reference file:line it follows.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this module.  The shipped package (vsmartmom.jl_amd) never does.

PINNED by the reference's only quantitative RRS fixture, test/reference/phase1b_RRS_sanghavi_q0.jld2 (R, T, ieR,
ieT of test_parameters/Phase1b_RRS_761-764nm.yaml; extracted to tests/golden/phase1b_rrs_sanghavi_q0.json by
tests/golden/make_fixtures.py), at the reference's own gate atol 1e-6 / rtol 0.02
(test/test_forward_raman_phase1b.jl:84-100): observed R 2.4e-4, ieR 9e-4, ieT 1e-2, T 1.4e-2 (the stored Float32 T
carries single-pixel noise).  The scene inputs (N2/O2 molecular constants -> Raman lines, Cabannes optics, Bodhaine
Rayleigh depth, reduced profile) come from the host-side producers vsmartmom.jl_amd/raman_inputs.py.
Also validated by an exact property of the equations it restates (tests/test_oracle_raman.py): the inelastic
recurrences are the first-order perturbation of the elastic ones, so on a spectrally uniform atmosphere with the
Raman phase matrix set to the elastic one, Sum_dn ieJ equals  eps * d(elastic J)/d(varpi)  at every interior
spectral point (central finite differences of the pinned elastic oracle), and out-of-band couplings contribute
exactly zero.

Array conventions (numpy, batch-first): 3-D arrays as in vsm_oracle.py
(A[s, i, j], v[s, i]); inelastic 4-D arrays are A[dn, s, i, j] and v[dn, s, i]
for the reference's A[i, j, n1, dn] (Julia column-major).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import vsm_oracle as O

RT_WEIGHT_TOL = 1e-8   # rt_helpers.jl:56-58
RT_CLOSE_TOL = 1e-8    # rt_helpers.jl:66-68
RT_LOOSE_TOL = 1e-6    # rt_helpers.jl:76


@dataclass
class RRS:
    """The fields of `RRS{FT}` (src/Inelastic/types.jl) that the CoreRT kernels read."""
    i_shift: np.ndarray        # i_lambda1lambda0 [K] (int): n0 = n1 + i_shift[dn]
    varpi_ie: np.ndarray       # varpi_lambda1lambda0 [K]
    greek_raman: O.GreekCoefs  # phase matrix of the Raman lines (get_greek_raman, inelastic_helper.jl:864-882)
    fscatt_rayl: Optional[np.ndarray] = None  # [S] of the current layer (_expand_layer_rayleigh!, rt_run.jl:221-223)
    Zpp_ie: Optional[np.ndarray] = None       # [N,N] of the current Fourier moment (computeRamanZlambda!, :917-924)
    Zmp_ie: Optional[np.ndarray] = None


@dataclass
class AddedLayerRS:
    """src/CoreRT/types.jl:308-335 (inelastic fields; the elastic ones live in O.AddedLayer)."""
    ier_mp: np.ndarray  # [K,S,N,N]
    iet_pp: np.ndarray
    ier_pm: np.ndarray
    iet_mm: np.ndarray
    ieJ0_p: np.ndarray  # [K,S,N]
    ieJ0_m: np.ndarray


@dataclass
class CompositeLayerRS:
    """src/CoreRT/types.jl:278-306."""
    ieR_mp: np.ndarray
    ieR_pm: np.ndarray
    ieT_pp: np.ndarray
    ieT_mm: np.ndarray
    ieJ0_p: np.ndarray
    ieJ0_m: np.ndarray


def make_added_layer_rs(FT, K, N, S) -> AddedLayerRS:
    z = lambda: np.zeros((K, S, N, N), dtype=FT)
    v = lambda: np.zeros((K, S, N), dtype=FT)
    return AddedLayerRS(z(), z(), z(), z(), v(), v())


def make_composite_layer_rs(FT, K, N, S) -> CompositeLayerRS:
    z = lambda: np.zeros((K, S, N, N), dtype=FT)
    v = lambda: np.zeros((K, S, N), dtype=FT)
    return CompositeLayerRS(z(), z(), z(), z(), v(), v())


def get_n0_n1(S: int, delta: int):
    """src/Inelastic/inelastic_helper.jl:19-26, as 0-based slices (n0, n1); empty when no overlap."""
    n1_start = max(1, 1 - delta)
    n1_end = min(S, S - delta)
    if n1_end < n1_start:
        return slice(0, 0), slice(0, 0)
    return slice(n1_start - 1 + delta, n1_end + delta), slice(n1_start - 1, n1_end)


def elemental_inelastic(rs: RRS, pol, tau_sum, dtau, F0, m, ndoubl, qp, add_rs: AddedLayerRS, FT):
    """src/CoreRT/CoreKernel/elemental_inelastic.jl:23-105 with kernels get_elem_rt_RRS! (:117-206),
    get_elem_rt_SFI_RRS! (:479-610) and apply_D_elemental_RRS! (:619-637).
    apply_D_matrix_elemental_SFI!(::RRS) is a no-op by construction (wrapper :745-762 vs kernel :684-692)."""
    mu = qp.qp_muN.astype(FT)
    N = len(mu)
    S = len(dtau)
    n = pol.n
    wct02 = FT(0.5) if m == 0 else FT(0.25)
    wct = (qp.wt_muN.astype(FT) / FT(2)) if m == 0 else (qp.wt_muN.astype(FT) / FT(4))
    Zpp = np.asarray(rs.Zpp_ie, dtype=FT)
    Zmp = np.asarray(rs.Zmp_ie, dtype=FT)
    fs = np.asarray(rs.fscatt_rayl, dtype=FT)
    F0 = np.asarray(F0, dtype=FT)
    dtau = dtau.astype(FT)
    tau_sum = np.asarray(tau_sum, dtype=FT)
    D = O._dsign(pol, N).astype(FT)
    mi = mu[None, :, None]
    mj = mu[None, None, :]
    active = (wct > RT_WEIGHT_TOL)[None, None, :]
    i_start = n * qp.imu0
    in_sun = np.zeros(N, dtype=bool)
    in_sun[i_start:i_start + n] = True
    mu_s = mu[i_start]
    one = FT(1)
    for a in (add_rs.ier_mp, add_rs.iet_pp, add_rs.ier_pm, add_rs.iet_mm, add_rs.ieJ0_p, add_rs.ieJ0_m):
        a[...] = 0
    for dn, (shift, w_ie) in enumerate(zip(rs.i_shift, rs.varpi_ie)):
        n0, n1 = get_n0_n1(S, int(shift))
        if n1.stop <= n1.start:
            continue
        w_ie = FT(w_ie)
        d1 = dtau[n1][:, None, None]
        d0 = dtau[n0][:, None, None]
        f0 = fs[n0][:, None, None]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            ratio = d1 / d0
            r = f0 * w_ie * Zmp[None] * (one / ((mi / mj) + ratio)) * (-np.expm1(-((d1 / mi) + (d0 / mj)))) * wct[None, None, :]
            # mu_i == mu_j
            t_eq_far = w_ie * f0 * Zpp[None] * wct[None, None, :] * O.expdiff_neg(d1 / mi, d0 / mj) / (one - ratio)
            t_eq_near = (d0 / mi) * w_ie * f0 * Zpp[None] * wct[None, None, :] * np.exp(-d0 / mj)
            t_eq = np.where(np.abs(d0 - d1) > RT_LOOSE_TOL, t_eq_far, t_eq_near)
            # mu_i != mu_j
            den = (mi / mj) - ratio
            t_ne_near = (d0 / mi) * w_ie * f0 * Zpp[None] * wct[None, None, :] * np.exp(-d0 / mj)
            t_ne_far = w_ie * f0 * Zpp[None] * (one / den) * wct[None, None, :] * O.expdiff_neg(d1 / mi, d0 / mj)
            t_ne = np.where(np.abs(den) < RT_CLOSE_TOL, t_ne_near, t_ne_far)
            t = np.where(mi == mj, t_eq, t_ne)
        r = np.where(active, r, FT(0)).astype(FT)
        t = np.where(active, t, FT(0)).astype(FT)
        # --- source (get_elem_rt_SFI_RRS!) ---
        ZF_p = np.einsum("ik,ks->si", Zpp[:, i_start:i_start + n], F0[:, n0])
        ZF_m = np.einsum("ik,ks->si", Zmp[:, i_start:i_start + n], F0[:, n0])
        e1 = dtau[n1][:, None]
        e0 = dtau[n0][:, None]
        g0 = fs[n0][:, None]
        mi1 = mu[None, :]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            rat = e1 / e0
            jp_sun_far = w_ie * g0 * ZF_p * wct02 * O.expdiff_neg(e1 / mi1, e0 / mi1) / (one - rat)
            jp_sun_near = (e0 / mi1) * wct02 * w_ie * g0 * ZF_p * np.exp(-e0 / mi1)
            jp_sun = np.where(np.abs(e0 - e1) > RT_CLOSE_TOL, jp_sun_far, jp_sun_near)
            den1 = (mi1 / mu_s) - rat
            jp_off_near = (e0 / mi1) * wct02 * w_ie * g0 * ZF_p * np.exp(-e0 / mu_s)
            jp_off_far = wct02 * w_ie * g0 * ZF_p * (one / den1) * O.expdiff_neg(e1 / mi1, e0 / mu_s)
            jp_off = np.where(np.abs(den1) < RT_CLOSE_TOL, jp_off_near, jp_off_far)
            jp = np.where(in_sun[None, :], jp_sun, jp_off)
            jm = wct02 * w_ie * g0 * ZF_m * (one / ((mi1 / mu_s) + rat)) * (-np.expm1(-((e1 / mi1) + (e0 / mu_s))))
        att = np.exp(-tau_sum[n0] / mu_s)[:, None]
        jp = (jp * att).astype(FT)
        jm = (jm * att).astype(FT)
        if ndoubl >= 1:
            jm = jm * D[None, :]
        add_rs.ieJ0_p[dn, n1] = jp
        add_rs.ieJ0_m[dn, n1] = jm
        add_rs.ier_mp[dn, n1] = r
        add_rs.iet_pp[dn, n1] = t
    # --- apply_D_matrix_elemental!(::RRS) (elemental_inelastic.jl:700-718) ---
    if n == 1:
        add_rs.ier_pm[...] = add_rs.ier_mp
        add_rs.iet_mm[...] = add_rs.iet_pp
    elif ndoubl < 1:
        par = (D[:, None] * D[None, :])[None, None]
        add_rs.ier_pm[...] = par * add_rs.ier_mp
        add_rs.iet_mm[...] = par * add_rs.iet_pp
    else:
        add_rs.ier_mp *= D[None, None, :, None]


def doubling_inelastic(rs: RRS, pol, expk, ndoubl, added: O.AddedLayer, add_rs: AddedLayerRS, FT):
    """src/CoreRT/CoreKernel/doubling_inelastic.jl:13-164 (doubling_helper!(::RRS), SFI = true) with the
    D-matrix kernels apply_D_IE_RRS! (:336-356) and apply_D_SFI_IE_RRS! (:408-416)."""
    if ndoubl == 0:
        return
    r, t = added.r_mp, added.t_pp
    jp, jm = added.j0_p, added.j0_m
    ier, iet = add_rs.ier_mp, add_rs.iet_pp
    ieJp, ieJm = add_rs.ieJ0_p, add_rs.ieJ0_m
    S, N = r.shape[0], r.shape[1]
    I = np.eye(N, dtype=FT)[None]
    expk = expk.astype(FT).copy()
    mv = O._mv
    for _ in range(ndoubl):
        gp = O.batch_inv(I - r @ r).astype(FT)
        ttg = t @ gp
        j1p = jp * expk[:, None]
        j1m = jm * expk[:, None]
        tmp1 = mv(gp, jp + mv(r, j1m))
        tmp2 = mv(gp, j1m + mv(r, jp))
        for dn, shift in enumerate(rs.i_shift):
            n0, n1 = get_n0_n1(S, int(shift))
            if n1.stop <= n1.start:
                continue
            ieJ1p = ieJp[dn, n1] * expk[n0][:, None]
            ieJ1m = ieJm[dn, n1] * expk[n0][:, None]
            X = r[n1] @ ier[dn, n1] + ier[dn, n1] @ r[n0]
            tmp3 = ieJ1p + mv(ttg[n1], ieJp[dn, n1] + mv(r[n1], ieJ1m) + mv(ier[dn, n1], j1m[n0]) + mv(X, tmp1[n0])) \
                + mv(iet[dn, n1], tmp1[n0])
            tmp4 = ieJm[dn, n1] + mv(ttg[n1], ieJ1m + mv(ier[dn, n1], jp[n0]) + mv(r[n1], ieJp[dn, n1])
                                     + mv(X, tmp2[n0])) + mv(iet[dn, n1], tmp2[n0])
            ieJp[dn, n1] = tmp3
            ieJm[dn, n1] = tmp4
        jm_new = jm + mv(ttg, j1m + mv(r, jp))
        jp_new = j1p + mv(ttg, jp + mv(r, j1m))
        jm, jp = jm_new.astype(FT), jp_new.astype(FT)
        expk = expk ** 2
        for dn, shift in enumerate(rs.i_shift):
            n0, n1 = get_n0_n1(S, int(shift))
            if n1.stop <= n1.start:
                continue
            X = ier[dn, n1] @ r[n0] + r[n1] @ ier[dn, n1]
            gt = gp[n0] @ t[n0]
            tmp5 = ttg[n1] @ (iet[dn, n1] + X @ gt) + iet[dn, n1] @ gt
            tmp6 = ier[dn, n1] + (iet[dn, n1] @ gp[n0] @ r[n0] @ t[n0]) \
                + ttg[n1] @ (r[n1] @ iet[dn, n1] + (ier[dn, n1] + X @ gp[n0] @ r[n0]) @ t[n0])
            iet[dn, n1] = tmp5
            ier[dn, n1] = tmp6
        r_new = r + ttg @ r @ t
        t = (ttg @ t).astype(FT)
        r = r_new.astype(FT)
    D = O._dsign(pol, N).astype(FT)
    added.j0_p[...] = jp
    if pol.n == 1:
        added.r_mp[...] = r
        added.t_pp[...] = t
        added.r_pm[...] = r
        added.t_mm[...] = t
        added.j0_m[...] = jm
        add_rs.ier_pm[...] = ier
        add_rs.iet_mm[...] = iet
        return
    r = r * D[None, :, None]
    par = (D[:, None] * D[None, :])[None]
    added.r_mp[...] = r
    added.t_pp[...] = t
    added.r_pm[...] = par * r
    added.t_mm[...] = par * t
    added.j0_m[...] = jm * D[None, :]
    ier *= D[None, None, :, None]
    add_rs.ier_pm[...] = par[None] * ier
    add_rs.iet_mm[...] = par[None] * iet
    ieJm *= D[None, None, :]


def copy_added_to_composite_ie(comp: O.CompositeLayer, comp_rs: CompositeLayerRS, add: O.AddedLayer, add_rs: AddedLayerRS):
    """rt_helpers.jl:222-228."""
    O.copy_added_to_composite(comp, add)
    comp_rs.ieT_pp[...] = add_rs.iet_pp
    comp_rs.ieT_mm[...] = add_rs.iet_mm
    comp_rs.ieR_mp[...] = add_rs.ier_mp
    comp_rs.ieR_pm[...] = add_rs.ier_pm
    comp_rs.ieJ0_p[...] = add_rs.ieJ0_p
    comp_rs.ieJ0_m[...] = add_rs.ieJ0_m


def interaction_inelastic_11(rs: RRS, comp: O.CompositeLayer, crs: CompositeLayerRS, add: O.AddedLayer,
                             ars: AddedLayerRS, FT):
    """src/CoreRT/CoreKernel/interaction_inelastic.jl:319-521 (interaction_helper!(::RRS, ::ScatteringInterface_11),
    SFI = true, no staging).  Every right-hand side uses pre-update values; out-of-band (n1, dn) blocks of the
    composite become zero (the temporaries are zero-initialised, :369-381)."""
    r_mp, r_pm, t_pp, t_mm, j0_p, j0_m = add.r_mp, add.r_pm, add.t_pp, add.t_mm, add.j0_p, add.j0_m
    R_mp, R_pm, T_pp, T_mm, J0_p, J0_m = comp.R_mp, comp.R_pm, comp.T_pp, comp.T_mm, comp.J0_p, comp.J0_m
    S, N = r_mp.shape[0], r_mp.shape[1]
    I = np.eye(N, dtype=FT)[None]
    mv = O._mv
    new = make_composite_layer_rs(FT, len(rs.i_shift), N, S)
    G1 = O.batch_inv(I - r_mp @ R_pm).astype(FT)
    T01 = T_mm @ G1
    v1 = mv(G1, j0_m + mv(r_mp, J0_p))
    grT = G1 @ r_mp @ T_pp
    gt = G1 @ t_mm
    for dn, shift in enumerate(rs.i_shift):
        n0, n1 = get_n0_n1(S, int(shift))
        if n1.stop <= n1.start:
            continue
        Y = T01[n1] @ (ars.ier_mp[dn, n1] @ R_pm[n0] + r_mp[n1] @ crs.ieR_pm[dn, n1]) + crs.ieT_mm[dn, n1]
        new.ieJ0_m[dn, n1] = crs.ieJ0_m[dn, n1] + mv(T01[n1], mv(ars.ier_mp[dn, n1], J0_p[n0])
                                                     + mv(r_mp[n1], crs.ieJ0_p[dn, n1]) + ars.ieJ0_m[dn, n1]) \
            + mv(Y, v1[n0])
        new.ieR_mp[dn, n1] = crs.ieR_mp[dn, n1] + T01[n1] @ (ars.ier_mp[dn, n1] @ T_pp[n0] + r_mp[n1] @ crs.ieT_pp[dn, n1]) \
            + Y @ grT[n0]
        new.ieT_mm[dn, n1] = T01[n1] @ ars.iet_mm[dn, n1] + Y @ gt[n0]
    J0_m_new = J0_m + mv(T01, mv(r_mp, J0_p) + j0_m)
    R_mp_new = R_mp + T01 @ r_mp @ T_pp
    T_mm_new = T01 @ t_mm
    G2 = O.batch_inv(I - R_pm @ r_mp).astype(FT)
    T21 = t_pp @ G2
    v2 = mv(G2, J0_p + mv(R_pm, j0_m))
    gT = G2 @ T_pp
    gRt = G2 @ R_pm @ t_mm
    for dn, shift in enumerate(rs.i_shift):
        n0, n1 = get_n0_n1(S, int(shift))
        if n1.stop <= n1.start:
            continue
        Y = T21[n1] @ (crs.ieR_pm[dn, n1] @ r_mp[n0] + R_pm[n1] @ ars.ier_mp[dn, n1]) + ars.iet_pp[dn, n1]
        new.ieJ0_p[dn, n1] = ars.ieJ0_p[dn, n1] + mv(T21[n1], crs.ieJ0_p[dn, n1] + mv(crs.ieR_pm[dn, n1], j0_m[n0])
                                                     + mv(R_pm[n1], ars.ieJ0_m[dn, n1])) + mv(Y, v2[n0])
        new.ieT_pp[dn, n1] = T21[n1] @ crs.ieT_pp[dn, n1] + Y @ gT[n0]
        new.ieR_pm[dn, n1] = ars.ier_pm[dn, n1] + T21[n1] @ (crs.ieR_pm[dn, n1] @ t_mm[n0] + R_pm[n1] @ ars.iet_mm[dn, n1]) \
            + Y @ gRt[n0]
    J0_p_new = j0_p + mv(T21, J0_p + mv(R_pm, j0_m))
    T_pp_new = T21 @ T_pp
    R_pm_new = r_pm + T21 @ R_pm @ t_mm
    comp.J0_m[...] = J0_m_new
    comp.R_mp[...] = R_mp_new
    comp.T_mm[...] = T_mm_new
    comp.J0_p[...] = J0_p_new
    comp.T_pp[...] = T_pp_new
    comp.R_pm[...] = R_pm_new
    for f in ("ieJ0_m", "ieJ0_p", "ieT_mm", "ieR_mp", "ieT_pp", "ieR_pm"):
        getattr(crs, f)[...] = getattr(new, f)


def interaction_inelastic(iface, rs: RRS, comp: O.CompositeLayer, crs: CompositeLayerRS, add: O.AddedLayer, ars: AddedLayerRS, FT):
    """interaction!(RS_type::RRS, scattering_interface, ...) (interaction_inelastic.jl:523-539) for the four interface tags.
    _00 / _01 / _10 follow the statements of interaction_inelastic.jl:74-101, 103-154, 215-262 over the in-band (n1, dn) pairs
    (their loop headers name an undefined `ieJ1+` upstream, so the branches cannot run there as committed; out-of-band blocks of
    the composite become / stay zero as in the _11 pass).  Inelastic right-hand sides use the pre-update elastic composite."""
    if iface == "11":
        return interaction_inelastic_11(rs, comp, crs, add, ars, FT)
    S = add.r_mp.shape[0]
    mv = O._mv
    if iface == "00":
        for f in ("ieJ0_m", "ieJ0_p", "ieT_mm", "ieR_mp", "ieT_pp", "ieR_pm"):
            getattr(crs, f)[...] = 0
    elif iface == "01":
        new = make_composite_layer_rs(FT, len(rs.i_shift), add.r_mp.shape[1], S)
        for dn, shift in enumerate(rs.i_shift):
            n0, n1 = get_n0_n1(S, int(shift))
            if n1.stop <= n1.start:
                continue
            new.ieJ0_m[dn, n1] = mv(comp.T_mm[n1], mv(ars.ier_mp[dn, n1], comp.J0_p[n0]) + ars.ieJ0_m[dn, n1])
            new.ieJ0_p[dn, n1] = ars.ieJ0_p[dn, n1] + mv(ars.iet_pp[dn, n1], comp.J0_p[n0])
            new.ieR_mp[dn, n1] = comp.T_mm[n1] @ ars.ier_mp[dn, n1] @ comp.T_pp[n0]
            new.ieR_pm[dn, n1] = ars.ier_pm[dn, n1]
            new.ieT_pp[dn, n1] = ars.iet_pp[dn, n1] @ comp.T_pp[n0]
            new.ieT_mm[dn, n1] = comp.T_mm[n1] @ ars.iet_mm[dn, n1]
        for f in ("ieJ0_m", "ieJ0_p", "ieT_mm", "ieR_mp", "ieT_pp", "ieR_pm"):
            getattr(crs, f)[...] = getattr(new, f)
    elif iface == "10":
        for dn, shift in enumerate(rs.i_shift):
            n0, n1 = get_n0_n1(S, int(shift))
            if n1.stop <= n1.start:
                continue
            Jp = mv(add.t_pp[n1], crs.ieJ0_p[dn, n1] + mv(crs.ieR_pm[dn, n1], add.j0_m[n0]))
            Jm = crs.ieJ0_m[dn, n1] + mv(crs.ieT_mm[dn, n1], add.j0_m[n0])
            Tpp = add.t_pp[n1] @ crs.ieT_pp[dn, n1]
            Tmm = crs.ieT_mm[dn, n1] @ add.t_mm[n0]
            Rpm = add.t_pp[n1] @ crs.ieR_pm[dn, n1] @ add.t_mm[n0]
            crs.ieJ0_p[dn, n1], crs.ieJ0_m[dn, n1] = Jp, Jm
            crs.ieT_pp[dn, n1], crs.ieT_mm[dn, n1], crs.ieR_pm[dn, n1] = Tpp, Tmm, Rpm
    else:
        raise ValueError(iface)
    O.interaction(iface, comp, add, FT)


def rt_kernel_rrs(rs: RRS, pol, added, add_rs, comp, comp_rs, props, tau_sum, m, qp, iz, F0, FT,
                  numerics: O.Numerics = O.Numerics(), trace=None, iface="11"):
    """src/CoreRT/CoreKernel/rt_kernel.jl:352-391 (scatter hard-wired to true, :365)."""
    tau, varpi = props.tau, props.varpi
    dtau, ndoubl = O.get_dtau_ndoubl(tau, varpi, qp, FT, numerics)
    expk = np.exp(-dtau / FT(qp.mu0)).astype(FT)
    elemental_inelastic(rs, pol, tau_sum, dtau, F0, m, ndoubl, qp, add_rs, FT)
    O.elemental(pol, tau_sum, dtau, F0, varpi, props.Zpp, props.Zmp, m, ndoubl, qp, added, FT)
    doubling_inelastic(rs, pol, expk, ndoubl, added, add_rs, FT)
    if trace is not None:
        trace.append(dict(iz=iz, m=m, ndoubl=ndoubl))
    if iz == 1:
        copy_added_to_composite_ie(comp, comp_rs, added, add_rs)
    else:
        interaction_inelastic(iface, rs, comp, comp_rs, added, add_rs, FT)   # dispatched on the tag (rt_kernel.jl:385)


def postprocessing_vza_rs(pol, comp, comp_rs, vza, vaz, qp, m, weight, R_SFI, T_SFI, ieR_SFI, ieT_SFI):
    """src/CoreRT/tools/postprocessing_vza.jl:117-151 (SFI branch)."""
    O.postprocessing_vza(pol, comp, vza, vaz, qp, m, weight, R_SFI, T_SFI)
    n = pol.n
    sJm = comp_rs.ieJ0_m.sum(axis=0)
    sJp = comp_rs.ieJ0_p.sum(axis=0)
    for i in range(len(vza)):
        imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(O.cosd(vza[i])))))
        istart = imu * n
        c, s = O.cosd(m * vaz[i]), O.sind(m * vaz[i])
        w = weight * np.array([c, c, s, s][:n])
        ieR_SFI[i] += w[:, None] * sJm[:, istart:istart + n].T
        ieT_SFI[i] += w[:, None] * sJp[:, istart:istart + n].T


def rt_run_rrs(model: O.RTModel, rs: RRS, fscatt=None, trace=None, per_m=None):
    """src/CoreRT/rt_run.jl:238-535 for RS_type::RRS (SFI, Lambertian surface).  `model.greek_rayleigh` is the
    elastic (Cabannes) phase matrix and model.varpi_cabannes the Rayleigh single-scattering albedo
    (compEffectiveLayerProperties.jl:36-41).  fscatt [S, L]: per-layer fScattRayleigh (:56); default tau_rayl /
    (tau_rayl + sum of the delta-M scaled aerosol taus).  Returns (R_SFI, T_SFI, ieR_SFI, ieT_SFI)."""
    FT = model.FT
    pol, qp = model.pol, model.quad_points
    S, L = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    K = len(rs.i_shift)
    nV = len(model.vza)
    out = [np.zeros((nV, pol.n, S), dtype=FT) for _ in range(4)]
    R_SFI, T_SFI, ieR_SFI, ieT_SFI = out
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S), dtype=FT)
        F0[0, :] = 1
    if fscatt is None:
        fscatt = default_fscatt(model)
    added = O.make_added_layer(FT, N, S)
    added_surf = O.make_added_layer(FT, N, S)
    comp = O.make_composite_layer(FT, N, S)
    add_rs = make_added_layer_rs(FT, K, N, S)
    surf_rs = make_added_layer_rs(FT, K, N, S)   # stays zero: the surface has no inelastic part
    comp_rs = make_composite_layer_rs(FT, K, N, S)
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        rs.Zpp_ie, rs.Zmp_ie = O.compute_Z_moments(pol, qp.qp_mu, rs.greek_raman, m)
        lods = O.construct_core_optical_properties(model, m)
        ifaces, tau_sum_all = O.extract_effective_props(lods, FT)
        for iz in range(L):
            rs.fscatt_rayl = np.asarray(fscatt[:, iz], dtype=FT)
            lo = O.expand_optical_properties(lods[iz], FT)
            rt_kernel_rrs(rs, pol, added, add_rs, comp, comp_rs, lo, tau_sum_all[:, iz].astype(FT), m, qp, iz + 1, F0, FT,
                          model.numerics, trace, ifaces[iz])
        O.create_surface_layer_lambertian(model.albedo, added_surf, m, pol, qp, tau_sum_all[:, -1], FT)
        interaction_inelastic(ifaces[-1], rs, comp, comp_rs, added_surf, surf_rs, FT)
        if per_m is not None:
            per_m.append(dict(m=m, J0_m=comp.J0_m.copy(), J0_p=comp.J0_p.copy(), ieJ0_m=comp_rs.ieJ0_m.copy(),
                              ieJ0_p=comp_rs.ieJ0_p.copy(), weight=weight))
        postprocessing_vza_rs(pol, comp, comp_rs, model.vza, model.vaz, qp, m, weight, R_SFI, T_SFI, ieR_SFI, ieT_SFI)
    return R_SFI, T_SFI, ieR_SFI, ieT_SFI


def default_fscatt(model: O.RTModel):
    """fScattRayleigh = tau_rayl / tau(rayl + aerosols) per layer (compEffectiveLayerProperties.jl:56)."""
    tau_sc = model.tau_rayl.astype(np.float64).copy()
    for a, ao in enumerate(model.aerosol_optics):
        tau_sc = tau_sc + ((1 - ao.f_trunc * ao.ssa) * model.tau_aer[a])[None, :]
    return model.tau_rayl / tau_sc
