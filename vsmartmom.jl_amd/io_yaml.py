"""YAML -> parameters -> RTModel for the Rayleigh + Lambertian / Cox-Munk subset of the reference's scene files.

Host-side mirror of `parameters_from_yaml` (src/IO/Parameters.jl:1021-1075, the new `nstreams` schema :1102-1175) and of
the parts of `model_from_parameters` (src/CoreRT/tools/model_from_parameters.jl:211-302) that such scenes exercise:
quadrature from `nstreams`, profile fields / reduction, the depolarization rule (`depol < 0`: from the N2/O2 molecular
constants), Bodhaine Rayleigh optical depth.  `config/quickstart.yaml` and `config/lambertian_land.yaml` of the reference
are of this kind.  Blocks that need components outside this backend (absorption -> HITRAN tables, scattering -> Mie,
surfaces other than Lambertian / Cox-Munk) raise NotImplementedError instead of being silently ignored.  `config/ocean_coxmunk.yaml`
(BASELINE config C3) parses to a CoxMunkSurface model with the Fourier bound of the reference's trait aggregator (m <= 21).
"""
from __future__ import annotations

import ast
import operator
import re
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import host_model as H
from . import raman_inputs as RI


@dataclass
class vSmartMOM_Parameters:
    spec_bands: List[np.ndarray]
    albedo: List[float]               # NaN for non-Lambertian surfaces
    nstreams: int
    polarization_type: str
    depol: float
    float_type: type
    architecture: str
    sza: float
    vza: List[float]
    vaz: List[float]
    obs_alt: float
    T: List[float]
    p: List[float]
    q: List[float]
    profile_reduction_n: int = -1
    brdf: List[object] = field(default_factory=list)   # params.brdf: one surface per band
    l_trunc: int = field(init=False)
    max_m: int = field(init=False)

    def __post_init__(self):
        if self.nstreams < 3:   # Parameters.jl:1131-1137
            raise ValueError("radiative_transfer.nstreams = %d; must be >= 3 for solar/scattering scenes" % self.nstreams)
        self.l_trunc = 2 * self.nstreams - 1    # stream_l_cap (Parameters.jl:1141-1156)
        self.max_m = self.l_trunc + 1


_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow, ast.USub: operator.neg, ast.UAdd: operator.pos}


def _num(expr: str) -> float:
    """Arithmetic on literals only (the scene files write e.g. `1e7/765`)."""
    def ev(n):
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return float(n.value)
        if isinstance(n, ast.BinOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.operand))
        raise ValueError("unsupported expression %r" % expr)
    return ev(ast.parse(expr.strip().replace("^", "**"), mode="eval").body)


def parse_spec_band(s: str) -> np.ndarray:
    """`[a b c]` (Julia vector literal) or `start:step:stop` (Julia range, stop inclusive when hit)."""
    s = str(s).strip()
    if s.startswith("["):
        return np.array([_num(x) for x in re.split(r"[\s,;]+", s.strip("[] ")) if x], dtype=np.float64)
    parts = []
    depth, cur = 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == ":" and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    if len(parts) == 2:
        a, st, b = _num(parts[0]), 1.0, _num(parts[1])
    elif len(parts) == 3:
        a, st, b = (_num(x) for x in parts)
    else:
        raise ValueError("cannot parse spec_band %r" % s)
    n = int(np.floor((b - a) / st + 1e-12)) + 1
    return a + st * np.arange(n)


def _split_args(body: str):
    """Top-level comma split of a constructor argument list; `k=v` items go to the keyword dict."""
    args, kw, depth, cur = [], {}, 0, ""
    for ch in body + ",":
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            item = cur.strip()
            cur = ""
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=", 1)
                kw[k.strip()] = v.strip()
            else:
                args.append(item)
        else:
            cur += ch
    return args, kw


def _complex(v: str) -> complex:
    m = re.fullmatch(r"(?:complex|Complex)\((.*),(.*)\)", v.replace(" ", ""))
    if m:
        return complex(_num(m.group(1)), _num(m.group(2)))
    m = re.fullmatch(r"(.+?)([+-].+)im", v.replace(" ", ""))
    if m:
        return complex(_num(m.group(1)), _num(m.group(2)))
    return complex(_num(v), 0.0)


def parse_surface(s: str):
    """`parse_surface` (src/IO/Parameters.jl:145-156,327-372): a constructor call string, without eval.
    Built here: LambertianSurfaceScalar(albedo), rpvSurfaceScalar(rho0, rho_c, k, Theta), RossLiSurfaceScalar(fvol, fgeo, fiso),
    CoxMunkSurface(U) / CoxMunkSurface(wind_speed=U, n_water=, whitecap_albedo=,
    include_whitecaps=, shadowing=)."""
    m = re.fullmatch(r"(\w+)(?:\{\w+\})?\((.*)\)", str(s).strip())
    if not m:
        raise ValueError("cannot parse surface %r" % s)
    name, (args, kw) = m.group(1), _split_args(m.group(2))
    if name == "LambertianSurfaceScalar":
        if len(args) != 1 or kw:
            raise ValueError("LambertianSurfaceScalar expects 1 argument (albedo)")
        return H.LambertianSurfaceScalar(_num(args[0]))
    if name == "LambertianSurfaceLegendre":
        if len(args) != 1 or kw:
            raise ValueError("LambertianSurfaceLegendre expects 1 argument (coefficient vector)")
        return H.LambertianSurfaceLegendre([_num(x) for x in re.split(r"[\s,;]+", args[0].strip("[] ")) if x])
    if name == "CoxMunkSurface":
        if kw:
            if args or "wind_speed" not in kw:
                raise ValueError("CoxMunkSurface keyword arguments require wind_speed.")
            unknown = set(kw) - {"wind_speed", "n_water", "whitecap_albedo", "include_whitecaps", "shadowing"}
            if unknown:
                raise ValueError("CoxMunkSurface: unknown keyword(s) %s" % sorted(unknown))
            b = lambda k, d: d if k not in kw else {"true": True, "false": False}[kw[k].lower()]
            return H.CoxMunkSurface(wind_speed=_num(kw["wind_speed"]),
                                    n_water=None if kw.get("n_water", "nothing") == "nothing" else _complex(kw["n_water"]),
                                    whitecap_albedo=_num(kw.get("whitecap_albedo", "0.22")),
                                    include_whitecaps=b("include_whitecaps", True), shadowing=b("shadowing", True))
        if len(args) != 1:
            raise ValueError("CoxMunkSurface expects 1 argument (wind_speed)")
        return H.CoxMunkSurface(wind_speed=_num(args[0]))
    if name == "rpvSurfaceScalar":
        if len(args) != 4 or kw:
            raise ValueError("rpvSurfaceScalar expects 4 arguments (rho0, rho_c, k, Theta)")
        return H.rpvSurfaceScalar(*[_num(a) for a in args])
    if name == "RossLiSurfaceScalar":
        if len(args) != 3 or kw:
            raise ValueError("RossLiSurfaceScalar expects 3 arguments (fvol, fgeo, fiso)")
        return H.RossLiSurfaceScalar(*[_num(a) for a in args])
    raise NotImplementedError("surface %r: LambertianSurfaceScalar / LambertianSurfaceLegendre, CoxMunkSurface, rpvSurfaceScalar and "
                              "RossLiSurfaceScalar are built in this backend" % s)


def _surface_albedo(s: str) -> float:
    surf = parse_surface(s)
    return surf.albedo if isinstance(surf, H.LambertianSurfaceScalar) else float("nan")


def parameters_from_dict(d: dict) -> vSmartMOM_Parameters:
    for blk in ("absorption", "scattering"):
        if d.get(blk):
            raise NotImplementedError("the `%s` block needs components outside this backend (SURVEY 8 out of scope)" % blk)
    rt, geo, atm = d["radiative_transfer"], d["geometry"], d["atmospheric_profile"]
    if rt.get("quadrature_type") not in (None, "GaussLegQuad()", "GaussLegQuad"):
        raise NotImplementedError("quadrature_type %r (GaussLegQuad only)" % rt.get("quadrature_type"))
    pol = re.sub(r"[()\s]", "", str(rt["polarization_type"])).replace("Stokes_", "")
    ft = {"Float64": np.float64, "Float32": np.float32}[str(rt.get("float_type", "Float64"))]
    T = [float(x) for x in atm["T"]]
    q = [float(x) for x in atm.get("q", [0.0] * len(T))]
    return vSmartMOM_Parameters(
        spec_bands=[parse_spec_band(b) for b in rt["spec_bands"]], albedo=[_surface_albedo(s) for s in rt["surface"]],
        nstreams=int(rt.get("nstreams", 8)), polarization_type=pol, depol=float(rt["depol"]), float_type=ft,
        architecture=str(rt.get("architecture", "default_architecture")), sza=float(geo["sza"]),
        vza=[float(x) for x in geo["vza"]], vaz=[float(x) for x in geo["vaz"]], obs_alt=float(geo.get("obs_alt", 0.0)),
        T=T, p=[float(x) for x in atm["p"]], q=q, profile_reduction_n=int(atm.get("profile_reduction", -1)),
        brdf=[parse_surface(s) for s in rt["surface"]])


def parameters_from_yaml(path_or_text: str) -> vSmartMOM_Parameters:
    import os
    import yaml
    if os.path.exists(path_or_text):
        with open(path_or_text, encoding="utf-8") as f:
            return parameters_from_dict(yaml.safe_load(f))
    return parameters_from_dict(yaml.safe_load(path_or_text))


def model_from_parameters(params: vSmartMOM_Parameters, architecture, iBand: int = 1) -> H.RTModel:
    """One band (rt_run(model) of this backend runs one band per call, like the reference's iBand = 1 default)."""
    nu = params.spec_bands[iBand - 1]
    prof = RI.compute_atmos_profile_fields(params.T, params.p, params.q)
    if params.profile_reduction_n != -1:
        prof = RI.reduce_profile(params.profile_reduction_n, prof)
    nu_m = 0.5 * (nu[0] + nu[-1])
    if params.depol < 0:
        n2, o2 = RI.get_raman_atmo_constants(nu_m, 300.0)
        g = RI.compute_gamma_air_rayleigh(n2, o2)
        depol = 2 * g / (1 + g)
    else:
        depol = params.depol
    tau_rayl = RI.rayleigh_layer_optical_depth(prof.p_half[-1], 1e4 / nu, depol, prof.vcd_dry)
    # Fourier bound (component_m_max.jl:60-131, model_from_parameters.jl:109-131): the maximum over the band's components --
    # Rayleigh 2, Lambertian / solar beam 0, Cox-Munk user_l_cap = min(2 nstreams - 1, max_m - 1, l_trunc) -- clamped to
    # user_l_cap (>= 5 here, so it never binds for Rayleigh + Lambertian).
    surface = params.brdf[iBand - 1]
    user_l_cap = min(2 * params.nstreams - 1, params.max_m - 1, params.l_trunc)
    brdf = isinstance(surface, (H.CoxMunkSurface, H.rpvSurfaceScalar, H.RossLiSurfaceScalar))   # component_m_max.jl:72-74
    m_max = min(max(2, user_l_cap if brdf else 0), user_l_cap)
    alb = surface.albedo if isinstance(surface, H.LambertianSurfaceScalar) else 0.0
    if isinstance(surface, H.LambertianSurfaceLegendre) and len(surface.legendre_coeff) < 2:
        raise ValueError("LambertianSurfaceLegendre needs at least two coefficients")
    return H.model_from_arrays(architecture, params.polarization_type, params.l_trunc, params.sza, params.vza, params.vaz,
                               tau_rayl, depol=depol, albedo=alb, m_max=m_max, float_type=params.float_type, surface=surface)
