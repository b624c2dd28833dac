"""Host-side producers of the rotational-Raman (RRS) inputs of the hot path and of the Rayleigh/Cabannes layer optics.

Mirror of the pieces of `src/Inelastic` and `src/CoreRT/tools` that turn (band grid, temperature profile) into what the
RRS kernels read -- `i_lambda1lambda0`, `varpi_lambda1lambda0`, `greek_raman`, `varpi_Cabannes`, the Cabannes Greek
coefficients and the Rayleigh optical depth per layer.  Plain numpy, FP64, runs once per band:

  MolecularConstants / get_molecular_constants   src/Inelastic/src/molecular_constructors.jl:2-143 (N2, O2 literature
                                                 constants: polarizability tensor, Dunham Y_kl, nuclear-spin weights)
  compute_effective_coefficients                 src/Inelastic/src/inelastic_cross_section.jl:33-60
  compute_energy_levels                          :162-182
  compute_sigma_rayl_coeff                       :69-76
  compute_sigma_rot_raman_coeff                  :222-273,370-384 (the pure-rotational J -> J-2 / J+2 part)
  get_raman_atmo_constants                       src/Inelastic/inelastic_helper.jl:28-47
  compute_varpi_cabannes / _mol                  :228-258, :298-321
  compute_gamma_mol_cabannes, compute_gamma_air_cabannes, compute_gamma_air_rayleigh   :384-450
  apply_gridlines, compute_optical_rs            :543-660
  get_raman_ss_prop                              src/Inelastic/raman_atmo_prop.jl:80-99 (the (RS, lambda, grid) method)
  normalize_raman_weights                        src/Inelastic/types.jl:625-629
  compute_atmos_profile_fields / reduce_profile / rayleigh_layer_optical_depth
                                                 src/CoreRT/tools/atmo_prof.jl:36-107,110-167,238-262
  rrs_band_setup                                 src/CoreRT/tools/model_from_parameters.jl:270-302 + the caller pattern of
                                                 test/test_forward_raman_phase1b.jl:41-66
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

# CGS constants of InelasticScattering.jl:29-33
C_LIGHT = 2.99792458e10
H_PLANCK = 6.62607015e-27
K_BOLTZ = 1.380649e-16
NM_PER_CM = 1.0e7


@dataclass
class MolecularConstants:
    vmr: float
    alpha00: float          # mean polarizability [cm^3]
    alpha00_prime: float
    omega0: float
    alpha_b: float
    alpha_c: float
    gamma00: float          # anisotropy [cm^3]
    gamma00_prime: float
    Y: np.ndarray           # Dunham coefficients Y[k,l] (1-based in the reference)
    gs: Sequence[int]       # nuclear-spin weights {odd J, even J}
    # effective coefficients
    alpha: float = 0.0
    gamma: float = 0.0
    eps: float = 0.0
    gamma_C_Rayl: float = 0.0
    gamma_C_RotRaman: float = 0.75
    rho_depol_Rayl: float = 0.0
    rho_depol_RotRaman: float = 0.0
    sigma_Rayl_coeff: float = 0.0
    E_vJ: Optional[np.ndarray] = None               # [v, J]
    sigma_RoRaman_JtoJm2: Optional[np.ndarray] = None   # [J]
    sigma_RoRaman_JtoJp2: Optional[np.ndarray] = None
    dnu_RoRaman_JtoJm2: Optional[np.ndarray] = None
    dnu_RoRaman_JtoJp2: Optional[np.ndarray] = None


def get_molecular_constants(name: str, vmr: float) -> MolecularConstants:
    Y = np.zeros((5, 5))
    if name == "N2":
        Y[0, 1], Y[0, 2], Y[1, 0], Y[1, 1], Y[2, 0], Y[3, 0] = 1.99824, -5.76e-6, 2358.57, -0.017318, -14.324, -2.26e-3
        return MolecularConstants(vmr=vmr, alpha00=1.7406e-24, alpha00_prime=1.86e-24, omega0=2.6049e16, alpha_b=1.8e-6,
                                  alpha_c=0.0, gamma00=0.71e-24, gamma00_prime=2.23e-24, Y=Y, gs=(3, 6))
    if name == "O2":
        Y[0, 1], Y[0, 2], Y[1, 0], Y[1, 1], Y[2, 0], Y[3, 0] = 1.4376766, -4.839e-6, 1580.19, -0.01590, -11.98, 0.0
        return MolecularConstants(vmr=vmr, alpha00=1.5658e-24, alpha00_prime=1.76e-24, omega0=2.1801e16,
                                  alpha_b=-2.369e-6, alpha_c=8.687e-9, gamma00=1.080e-24, gamma00_prime=3.19e-24, Y=Y,
                                  gs=(1, 0))
    raise ValueError(f"unknown molecule {name}")


def compute_effective_coefficients(nu_eff: float, T: float, mol: MolecularConstants) -> None:
    mol.alpha = mol.alpha00 * (1 + mol.alpha_b * T + mol.alpha_c * T * T) / (1 - (C_LIGHT * nu_eff / mol.omega0) ** 2)
    mol.gamma = mol.gamma00
    mol.eps = mol.alpha / mol.gamma
    mol.gamma_C_Rayl = 3 / (45 * mol.eps ** 2 + 4)
    mol.gamma_C_RotRaman = 0.75
    mol.rho_depol_Rayl = 2 * mol.gamma_C_Rayl / (1 + mol.gamma_C_Rayl)
    mol.rho_depol_RotRaman = 2 * mol.gamma_C_RotRaman / (1 + mol.gamma_C_RotRaman)


def compute_energy_levels(mol: MolecularConstants, vmax: int = 2, Jmax: int = 30) -> None:
    E = np.zeros((vmax + 1, Jmax + 1))
    for v in range(vmax + 1):
        for J in range(Jmax + 1):
            e1 = float(J * (J + 1))
            for l in range(5):
                for k in range(5):
                    E[v, J] += e1 ** l * (v + 0.5) ** k * mol.Y[k, l]
    mol.E_vJ = E


def compute_sigma_rayl_coeff(mol: MolecularConstants) -> None:
    g = mol.gamma_C_Rayl
    mol.sigma_Rayl_coeff = 128 * math.pi ** 5 * mol.alpha ** 2 * (1 + 2 * g) / (3 - 4 * g)


def compute_sigma_rot_raman_coeff(T: float, mol: MolecularConstants, Jmax: int = 30) -> None:
    kv = (256 / 27) * math.pi ** 5
    E = mol.E_vJ
    hck = H_PLANCK * C_LIGHT / (K_BOLTZ * T)
    sm2, sp2 = np.zeros(Jmax + 1), np.zeros(Jmax + 1)
    dm2, dp2 = np.zeros(Jmax + 1), np.zeros(Jmax + 1)
    Z_pf = 0.0
    for J in range(Jmax + 1):
        b_m2 = 3 * J * (J - 1) / (2 * (2 * J + 1) * (2 * J - 1))
        b_p2 = 3 * (J + 1) * (J + 2) / (2 * (2 * J + 1) * (2 * J + 3))
        g_N = mol.gs[1] if J % 2 == 0 else mol.gs[0]
        Ni = math.exp(-hck * E[0, J])
        Z_pf += g_N * (2 * J + 1) * (math.exp(-hck * E[0, J]) + math.exp(-hck * E[1, J]))
        if J - 2 >= 0:
            sm2[J] = kv * g_N * (2 * J + 1) * b_m2 * Ni * mol.gamma ** 2
            dm2[J] = -(E[0, J - 2] - E[0, J])
        if J + 2 <= Jmax:
            sp2[J] = kv * g_N * (2 * J + 1) * b_p2 * Ni * mol.gamma ** 2
            dp2[J] = -(E[0, J + 2] - E[0, J])
    mol.sigma_RoRaman_JtoJm2, mol.sigma_RoRaman_JtoJp2 = sm2 / Z_pf, sp2 / Z_pf
    mol.dnu_RoRaman_JtoJm2, mol.dnu_RoRaman_JtoJp2 = dm2, dp2


def get_raman_atmo_constants(nu: float, T: float, vmr_n2: float = 0.8, vmr_o2: float = 0.2):
    out = []
    for name, vmr in (("N2", vmr_n2), ("O2", vmr_o2)):
        mol = get_molecular_constants(name, vmr)
        compute_effective_coefficients(nu, T, mol)
        compute_energy_levels(mol)
        compute_sigma_rayl_coeff(mol)
        compute_sigma_rot_raman_coeff(T, mol)
        out.append(mol)
    return out[0], out[1]


def _sigma_rrs_total(nu0: float, mol: MolecularConstants) -> float:
    return float(np.dot((nu0 + mol.dnu_RoRaman_JtoJp2) ** 4, mol.sigma_RoRaman_JtoJp2)
                 + np.dot((nu0 + mol.dnu_RoRaman_JtoJm2) ** 4, mol.sigma_RoRaman_JtoJm2))


def compute_varpi_cabannes(lambda0_nm: float, n2: MolecularConstants, o2: MolecularConstants) -> float:
    nu0 = NM_PER_CM / lambda0_nm
    s_rayl = (n2.vmr * n2.sigma_Rayl_coeff + o2.vmr * o2.sigma_Rayl_coeff) * nu0 ** 4
    s_rrs = n2.vmr * _sigma_rrs_total(nu0, n2) + o2.vmr * _sigma_rrs_total(nu0, o2)
    return 1.0 - s_rrs / s_rayl


def compute_varpi_cabannes_mol(lambda0_nm: float, mol: MolecularConstants) -> float:
    nu0 = NM_PER_CM / lambda0_nm
    return 1.0 - _sigma_rrs_total(nu0, mol) / (mol.sigma_Rayl_coeff * nu0 ** 4)


def compute_gamma_mol_cabannes(lambda0_nm: float, mol: MolecularConstants) -> float:
    w = compute_varpi_cabannes_mol(lambda0_nm, mol)
    t1 = 1 + 2 * mol.gamma_C_Rayl
    return 0.5 * (t1 * (2 + 3 * w) - 5) / (t1 * (1 - w) + 5)


def _gamma_air(parts):
    tmp1 = sum(s * v for s, v, g in parts)
    tmp2 = sum(s * v * g / (3 - 4 * g) for s, v, g in parts)
    return 3 / (4 + tmp1 / tmp2)


def compute_gamma_air_cabannes(lambda0_nm: float, n2: MolecularConstants, o2: MolecularConstants) -> float:
    parts = []
    for mol in (n2, o2):
        g = compute_gamma_mol_cabannes(lambda0_nm, mol)
        w = compute_varpi_cabannes_mol(lambda0_nm, mol)
        parts.append((w * mol.sigma_Rayl_coeff * (3 - 4 * g) / (1 + 2 * g), mol.vmr, g))
    return _gamma_air(parts)


def compute_gamma_air_rayleigh(n2: MolecularConstants, o2: MolecularConstants) -> float:
    parts = [(mol.sigma_Rayl_coeff * (3 - 4 * mol.gamma_C_Rayl) / (1 + 2 * mol.gamma_C_Rayl), mol.vmr, mol.gamma_C_Rayl)
             for mol in (n2, o2)]
    return _gamma_air(parts)


def apply_gridlines(dnu_lines, sigma_lines, lambda0_nm: float, nu_grid: np.ndarray) -> np.ndarray:
    """Each transition inside the band is split half/half between its two neighbouring grid points."""
    nu0 = NM_PER_CM / lambda0_nm
    dgrid = nu_grid - nu0
    out = np.zeros_like(dgrid)
    gmin, gmax = dgrid.min(), dgrid.max()
    for dn, sg in zip(dnu_lines, sigma_lines):
        if gmin < dn < gmax:
            S = sg * (dn + nu0) ** 4
            i = int(np.argmin(np.abs(dn - dgrid)))
            lo, hi = (i, i + 1) if dgrid[i] < dn else (i - 1, i)
            out[lo] += S / 2
            out[hi] += S / 2
    return out


def compute_optical_rs(nu_grid: np.ndarray, lambda0_nm: float, n2: MolecularConstants, o2: MolecularConstants):
    tot = np.zeros(len(nu_grid))
    for mol in (n2, o2):
        tot += mol.vmr * apply_gridlines(mol.dnu_RoRaman_JtoJp2, mol.sigma_RoRaman_JtoJp2, lambda0_nm, nu_grid)
        tot += mol.vmr * apply_gridlines(mol.dnu_RoRaman_JtoJm2, mol.sigma_RoRaman_JtoJm2, lambda0_nm, nu_grid)
    idx = np.nonzero(tot > 0)[0]
    sig = tot[idx]
    nu0 = NM_PER_CM / lambda0_nm
    if nu_grid[0] < nu0 < nu_grid[-1]:
        idx = idx - int(np.argmin(np.abs(nu_grid - nu0)))
    return idx.astype(np.int64), sig


def greek_raman_coefficients(n2: MolecularConstants) -> dict:
    """get_greek_raman (inelastic_helper.jl:864-882): Rayleigh-type coefficients at the rotational-Raman depolarization."""
    depol = n2.rho_depol_RotRaman
    p = (1 - depol) / (1 + depol / 2)
    r = (1 - 2 * depol) / (1 - depol)
    return dict(alpha=[0.0, 0.0, 3 * p], beta=[1.0, 0.0, 0.5 * p], gamma=[0.0, 0.0, p * math.sqrt(1.5)],
                delta=[0.0, p * r * 1.5, 0.0], epsilon=[0.0, 0.0, 0.0], zeta=[0.0, 0.0, 0.0])


@dataclass
class RamanSSProp:
    i_shift: np.ndarray       # i_lambda1lambda0: donor index n0 = n1 + i_shift[k]
    varpi_ie: np.ndarray      # varpi_lambda1lambda0 (before normalize_raman_weights)
    varpi_cabannes: float     # RS_type.varpi_Cabannes (at the effective temperature)
    greek_raman: dict
    i_ref: int


def get_raman_ss_prop(lambda0_nm: float, nu_grid: np.ndarray, n2: MolecularConstants, o2: MolecularConstants) -> RamanSSProp:
    nu0 = NM_PER_CM / lambda0_nm
    s_rayl = (n2.vmr * n2.sigma_Rayl_coeff + o2.vmr * o2.sigma_Rayl_coeff) * nu0 ** 4
    idx, sig = compute_optical_rs(nu_grid, lambda0_nm, n2, o2)
    return RamanSSProp(i_shift=idx, varpi_ie=sig[::-1] / s_rayl, varpi_cabannes=compute_varpi_cabannes(lambda0_nm, n2, o2),
                       greek_raman=greek_raman_coefficients(n2), i_ref=int(np.argmin(np.abs(nu_grid - nu0))))


def normalize_raman_weights(varpi_ie: np.ndarray, model_varpi_cabannes: float) -> np.ndarray:
    return varpi_ie * (1 - model_varpi_cabannes) / varpi_ie.sum()


# ---------------------------------------------------------------------------
# atmospheric profile (atmo_prof.jl)
# ---------------------------------------------------------------------------
N_AVOGADRO = 6.02214179e+23
DRY_MASS, WET_MASS, G0 = 28.9644e-3, 18.01534e-3, 9.8032465


@dataclass
class AtmosphericProfile:
    T: np.ndarray
    p_full: np.ndarray
    q: np.ndarray
    p_half: np.ndarray
    vmr_h2o: np.ndarray
    vcd_dry: np.ndarray
    vcd_h2o: np.ndarray


def _vcds(p_half, vmr_h2o):
    dp = np.diff(p_half)
    M = (1 - vmr_h2o) * DRY_MASS + vmr_h2o * WET_MASS
    vcd = N_AVOGADRO * dp / (M * G0 * 100 ** 2) * 100
    return (1 - vmr_h2o) * vcd, vmr_h2o * vcd


def compute_atmos_profile_fields(T, p_half, q) -> AtmosphericProfile:
    T, p_half, q = (np.asarray(x, dtype=np.float64) for x in (T, p_half, q))
    q = q[:len(T)]   # the reference loops over length(T) layers (atmo_prof.jl:58)
    p_full = (p_half[1:] + p_half[:-1]) / 2
    vmr_h2o = q / (1 - q) * (DRY_MASS / WET_MASS)
    vcd_dry, vcd_h2o = _vcds(p_half, vmr_h2o)
    return AtmosphericProfile(T, p_full, q, p_half, vmr_h2o, vcd_dry, vcd_h2o)


def reduce_profile(n: int, prof: AtmosphericProfile) -> AtmosphericProfile:
    """Linear interpolation onto n uniform pressure layers; the reference interpolates on a UNIFORM grid between the
    extreme full-level pressures (atmo_prof.jl:132-136), reproduced here."""
    assert n < len(prof.T)
    p_half = np.linspace(prof.p_half[0], prof.p_half[-1], n + 1)
    p_full = (p_half[:-1] + p_half[1:]) / 2

    def interp(data):
        grid = np.linspace(prof.p_full.min(), prof.p_full.max(), len(data))
        return np.interp(p_full, grid, data)

    T, q, vmr_h2o = interp(prof.T), interp(prof.q), interp(prof.vmr_h2o)
    vcd_dry, vcd_h2o = _vcds(p_half, vmr_h2o)
    return AtmosphericProfile(T, p_full, q, p_half, vmr_h2o, vcd_dry, vcd_h2o)


def rayleigh_layer_optical_depth(psurf: float, lambda_um: np.ndarray, depol: float, vcd_dry: np.ndarray) -> np.ndarray:
    """Bodhaine 1999 Eq. 30, rescaled from its implicit depolarization 0.0279 to `depol`; [nSpec, nLayers]."""
    lam = np.atleast_1d(np.asarray(lambda_um, dtype=np.float64))
    tau = 0.002152 * (1.0455996 - 341.29061 * lam ** -2 - 0.90230850 * lam ** 2) / \
        (1 + 0.0027059889 * lam ** -2 - 85.968563 * lam ** 2)
    tau = tau * (psurf / 1013.25)
    rho0 = 0.0279
    tau = tau * (6 - 7 * rho0) * (6 + 3 * depol) / ((6 + 3 * rho0) * (6 - 7 * depol))
    return np.outer(tau / vcd_dry.sum(), vcd_dry)


@dataclass
class RRSBandSetup:
    """Everything `rt_run(RS_type::RRS, model, iBand)` reads for a Rayleigh + RRS band."""
    nu: np.ndarray
    tau_rayl: np.ndarray          # [S, L]
    depol_cabannes: float         # -> greek_cabannes (elastic phase matrix)
    depol_rayleigh: float
    varpi_cabannes_rs: float      # RS_type.varpi_Cabannes: single-scattering albedo of the elastic Rayleigh layers
    varpi_cabannes_model: float   # model.varpi_Cabannes (300 K): normalisation of the Raman weights
    i_shift: np.ndarray
    varpi_ie: np.ndarray          # normalised
    greek_raman: dict
    i_ref: int
    eff_T: float
    profile: AtmosphericProfile = field(repr=False, default=None)


def rrs_band_setup(nu: np.ndarray, T, p_half, q, profile_reduction_n: int = -1, depol: float = -1.0) -> RRSBandSetup:
    nu = np.asarray(nu, dtype=np.float64)
    prof = compute_atmos_profile_fields(T, p_half, q)
    if profile_reduction_n != -1:
        prof = reduce_profile(profile_reduction_n, prof)
    nu_m = 0.5 * (nu[0] + nu[-1])
    lam_m = NM_PER_CM / nu_m
    n2m, o2m = get_raman_atmo_constants(nu_m, 300.0)
    w_model = compute_varpi_cabannes(lam_m, n2m, o2m)
    g_cab = compute_gamma_air_cabannes(lam_m, n2m, o2m)
    g_ray = compute_gamma_air_rayleigh(n2m, o2m)
    dep_cab = 2 * g_cab / (1 + g_cab) if depol < 0 else depol
    dep_ray = 2 * g_ray / (1 + g_ray) if depol < 0 else depol
    tau_rayl = rayleigh_layer_optical_depth(prof.p_half[-1], 1e4 / nu, dep_ray, prof.vcd_dry)
    eff_T = float(np.dot(prof.vcd_dry, prof.T) / prof.vcd_dry.sum())
    nu_bar = float(nu.mean())
    n2, o2 = get_raman_atmo_constants(nu_bar, eff_T)
    ss = get_raman_ss_prop(NM_PER_CM / nu_bar, nu, n2, o2)
    return RRSBandSetup(nu=nu, tau_rayl=tau_rayl, depol_cabannes=dep_cab, depol_rayleigh=dep_ray,
                        varpi_cabannes_rs=ss.varpi_cabannes, varpi_cabannes_model=w_model, i_shift=ss.i_shift,
                        varpi_ie=normalize_raman_weights(ss.varpi_ie, w_model), greek_raman=ss.greek_raman,
                        i_ref=ss.i_ref, eff_T=eff_T, profile=prof)
