"""Rotational-Raman (RRS) CoreRT host layer: rt_run(RS_type::RRS, model, iBand) (src/CoreRT/rt_run.jl:238-535)
on the MI355X, through the C ABI (vsm_*_inelastic_rrs_*, vsmartmom.jl_amd/csrc/vsm_raman.hip).

Mirrors: RRS{FT} (src/Inelastic/types.jl; the fields the kernels read), AddedLayerRS / CompositeLayerRS
(src/CoreRT/types.jl:278-335), rt_kernel!(::RRS) (CoreKernel/rt_kernel.jl:352-391), elemental_inelastic!,
doubling_inelastic!, interaction!(::RRS, ::ScatteringInterface_11), copy_added_to_composite_ie!,
postprocessing_vza!(::RRS).  Operator level this round: one launch per batched operator over ALL
(spectral point, Raman offset) pairs.  The producers of the Raman inputs (getRamanSSProp!, N2/O2 constants)
are outside the hot path; `RRS` takes them as arrays.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import host_model as H
from . import core_rt as CR
from .architectures import array_type, devi, synchronize_if_gpu, to_host


@dataclass
class RRS:
    """RS_type::RRS as the CoreRT kernels see it."""
    i_lambda1lambda0: np.ndarray      # i_λ₁λ₀ [K] int: donor index n0 = n1 + i_λ₁λ₀[Δn]
    varpi_lambda1lambda0: np.ndarray  # ϖ_λ₁λ₀ [K]
    greek_raman: H.GreekCoefs         # get_greek_raman (inelastic_helper.jl:864-882)
    fscattRayl: Optional[np.ndarray] = None   # [S, Nz] fScattRayleigh per layer (compEffectiveLayerProperties.jl:56)


class AddedLayerRS:
    """types.jl:308-335 (inelastic fields; the elastic ones are a CR.AddedLayer)."""

    def __init__(self, FT, arch, K, N, nSpec):
        dev, dt = devi(arch), CR._torch_dtype(FT)
        z = lambda: torch.zeros((K, nSpec, N, N), dtype=dt, device=dev)
        v = lambda: torch.zeros((K, nSpec, N), dtype=dt, device=dev)
        self.ier_mp, self.iet_pp, self.ier_pm, self.iet_mm = z(), z(), z(), z()
        self.ieJ0_p, self.ieJ0_m = v(), v()
        self.K, self.N, self.nSpec, self.dtype = K, N, nSpec, dt

    def cstruct(self):
        return _lib.vsm_added_rs(self.ier_mp.data_ptr(), self.iet_pp.data_ptr(), self.ier_pm.data_ptr(),
                                 self.iet_mm.data_ptr(), self.ieJ0_p.data_ptr(), self.ieJ0_m.data_ptr(), self.K, 0)


class CompositeLayerRS:
    """types.jl:278-306."""

    def __init__(self, FT, arch, K, N, nSpec):
        dev, dt = devi(arch), CR._torch_dtype(FT)
        z = lambda: torch.zeros((K, nSpec, N, N), dtype=dt, device=dev)
        v = lambda: torch.zeros((K, nSpec, N), dtype=dt, device=dev)
        self.ieR_mp, self.ieR_pm, self.ieT_pp, self.ieT_mm = z(), z(), z(), z()
        self.ieJ0_p, self.ieJ0_m = v(), v()
        self.K, self.N, self.nSpec, self.dtype = K, N, nSpec, dt

    def cstruct(self):
        return _lib.vsm_composite_rs(self.ieR_mp.data_ptr(), self.ieR_pm.data_ptr(), self.ieT_pp.data_ptr(),
                                     self.ieT_mm.data_ptr(), self.ieJ0_p.data_ptr(), self.ieJ0_m.data_ptr(), self.K, 0)


@dataclass
class DeviceRRS:
    """Device copies of the RRS fields + the per-layer / per-moment inputs the kernels take."""
    shift: torch.Tensor       # int32 [K]
    varpi_ie: torch.Tensor    # [K]
    K: int
    fscatt: Optional[torch.Tensor] = None  # [S] current layer
    Zpp: Optional[torch.Tensor] = None     # (1,N,N) layout tensor, current Fourier moment
    Zmp: Optional[torch.Tensor] = None

    def cstruct(self):
        return _lib.vsm_rrs(self.shift.data_ptr(), self.varpi_ie.data_ptr(), self.fscatt.data_ptr(), self.Zpp.data_ptr(),
                            self.Zmp.data_ptr())


def device_rrs(rs: RRS, arch, FT) -> DeviceRRS:
    dev = devi(arch)
    shift = torch.tensor(np.asarray(rs.i_lambda1lambda0, dtype=np.int32), dtype=torch.int32, device=dev)
    return DeviceRRS(shift, array_type(arch)(np.asarray(rs.varpi_lambda1lambda0, dtype=FT)), int(shift.numel()))


_work = {}


def _workbuf(name, elems, dtype, device):
    key = (name, str(device), dtype)
    w = _work.get(key)
    if w is None or w.numel() < elems:
        w = _lib.poison(torch.empty(max(int(elems), 1), dtype=dtype, device=device))
        _work[key] = w
    return w


def elemental_inelastic_(drs: DeviceRRS, tau_sum, dtau, F0, m, ndoubl, dq: CR.DeviceQuad, added_rs: AddedLayerRS):
    """elemental_inelastic!(::RRS) (elemental_inelastic.jl:23-105)."""
    q, a, r = dq.cstruct(), added_rs.cstruct(), drs.cstruct()
    _lib.call("vsm_elemental_inelastic_rrs", added_rs.dtype, C.byref(q), added_rs.nSpec, m, ndoubl, CR._ptr(dtau),
              CR._ptr(tau_sum), CR._ptr(F0), C.byref(r), C.byref(a), CR._stream_ptr())


def doubling_inelastic_(drs: DeviceRRS, pol, expk, ndoubl, added: CR.AddedLayer, added_rs: AddedLayerRS):
    """doubling_inelastic! (doubling_inelastic.jl:13-164, 313-328)."""
    N, S, K = added.N, added.nSpec, added_rs.K
    work = _workbuf("dbl", _lib.lib().vsm_doubling_inelastic_work_elems(N, S, K), added.dtype, added.r_mp.device)
    a, ar = added.cstruct(), added_rs.cstruct()
    _lib.call("vsm_doubling_inelastic_rrs", added.dtype, N, pol.n, S, ndoubl, CR._ptr(expk), CR._ptr(drs.shift), C.byref(a),
              C.byref(ar), CR._ptr(work), CR._stream_ptr())


def interaction_inelastic_(drs: DeviceRRS, scattering_interface: str, comp: CR.CompositeLayer, comp_rs: CompositeLayerRS,
                           added: CR.AddedLayer, added_rs: AddedLayerRS):
    """interaction!(RS_type::RRS, ...) (interaction_inelastic.jl:683-700 -> :319-521)."""
    N, S, K = comp.N, comp.nSpec, comp_rs.K
    work = _workbuf("ia", _lib.lib().vsm_interaction_inelastic_work_elems(N, S, K), comp.dtype, comp.R_mp.device)
    a, ar, c, cr = added.cstruct(), added_rs.cstruct(), comp.cstruct(), comp_rs.cstruct()
    _lib.call("vsm_interaction_inelastic_rrs", comp.dtype, CR.IFACE[scattering_interface], N, S, CR._ptr(drs.shift),
              C.byref(c), C.byref(cr), C.byref(a), C.byref(ar), CR._ptr(work), CR._stream_ptr())


def copy_added_to_composite_ie_(comp, comp_rs: CompositeLayerRS, added, added_rs: AddedLayerRS):
    """copy_added_to_composite_ie! (rt_helpers.jl:222-228)."""
    CR.copy_added_to_composite_(comp, added)
    ar, cr = added_rs.cstruct(), comp_rs.cstruct()
    _lib.call("vsm_copy_added_to_composite_ie", comp.dtype, comp.N, comp.nSpec, C.byref(ar), C.byref(cr), CR._stream_ptr())


def rt_kernel_rrs_(drs: DeviceRRS, pol, added, added_rs, comp, comp_rs, props: CR.DeviceLayerOptics, tau_sum, m, dq, arch, iz,
                   F0, FT, numerics, dtau=None, ndoubl=None, expk=None, trace=None):
    """rt_kernel!(::RRS, ...) (rt_kernel.jl:352-391): scatter is hard-wired to true (:365).
    dtau / ndoubl / expk may be passed pre-computed (sharded runs derive them from the full spectral axis)."""
    if dtau is None:
        dtau_h, ndoubl = H.get_dtau_ndoubl(props.tau_h, props.varpi_h, dq.host, FT, numerics)
        dtau = array_type(arch)(dtau_h)
        expk = array_type(arch)(np.exp(-dtau_h / FT(dq.host.mu0)).astype(FT))   # arr_type(exp.(-dτ/μ₀)) (rt_kernel.jl:367)
    elemental_inelastic_(drs, tau_sum, dtau, F0, m, ndoubl, dq, added_rs)
    CR.elemental_(pol, tau_sum, dtau, F0, props, m, ndoubl, dq, added)
    doubling_inelastic_(drs, pol, expk, ndoubl, added, added_rs)
    if trace is not None:
        trace.append(dict(iz=iz, m=m, ndoubl=ndoubl))
    if iz == 1:
        copy_added_to_composite_ie_(comp, comp_rs, added, added_rs)
    else:
        interaction_inelastic_(drs, "11", comp, comp_rs, added, added_rs)


def postprocessing_vza_rs_(pol, comp, comp_rs: CompositeLayerRS, vza, vaz, qp, m, weight, R_SFI, T_SFI, ieR_SFI, ieT_SFI):
    """postprocessing_vza!(::RRS, ...) (postprocessing_vza.jl:117-151), SFI branch."""
    CR.postprocessing_vza_(pol, comp, vza, vaz, qp, m, weight, R_SFI, T_SFI)
    n, nV = pol.n, len(vza)
    row0 = (C.c_int * nV)()
    ctype = C.c_double if comp.dtype == torch.float64 else C.c_float
    w = (ctype * (nV * n))()
    for v in range(nV):
        imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(H.cosd(vza[v])))))
        row0[v] = imu * n
        c, s = H.cosd(m * vaz[v]), H.sind(m * vaz[v])
        ws = [c, c, s, s][:n]
        for k in range(n):
            w[v + nV * k] = weight * ws[k]
    _lib.call("vsm_postprocess_vza_ie", comp.dtype, comp.N, n, comp.nSpec, comp_rs.K, nV, row0, w, CR._ptr(comp_rs.ieJ0_m),
              CR._ptr(comp_rs.ieJ0_p), CR._ptr(ieR_SFI), CR._ptr(ieT_SFI), CR._stream_ptr())


def default_fscatt(model: H.RTModel) -> np.ndarray:
    """fScattRayleigh = τ_rayl / τ(Rayleigh + δ-M scaled aerosols) (compEffectiveLayerProperties.jl:56)."""
    tau_sc = model.tau_rayl.astype(np.float64).copy()
    for a, ao in enumerate(model.aerosol_optics):
        tau_sc = tau_sc + ((1 - ao.f_trunc * ao.ssa) * model.tau_aer[a])[None, :]
    return model.tau_rayl / tau_sc


def rt_run(RS_type: RRS, model: H.RTModel, iBand: int = 1, trace: Optional[list] = None, spec_slice: Optional[slice] = None,
           device_out: bool = False):
    """rt_run(RS_type::RRS, model, iBand) (rt_run.jl:238-535): returns (R_SFI, T_SFI, ieR_SFI, ieT_SFI) as host arrays
    [nVZA, nStokes, nSpec].  `model.greek_rayleigh` must hold the Cabannes phase matrix and `model.varpi_Cabannes`
    the elastic Rayleigh single-scattering albedo (compEffectiveLayerProperties.jl:36-41).

    `spec_slice` = the recipient points this rank owns (multi-GPU): the run covers the slice extended
    by a halo of max|i_λ₁λ₀| donor points on each side (`parallel.raman_halo_slices`), ndoubl comes from the FULL
    spectral axis, and only the owned points are returned -- no exchange step is needed.
    `device_out=True` returns the four (S_local, nStokes, nVZA) device tensors instead (for the gather)."""
    from . import parallel
    arch, FT = model.architecture, model.float_type
    CR._require_gpu(arch)
    if iBand != 1:
        raise _lib.VSMError("single-band models only (iBand = 1)")
    pol, qp = model.polarization_type, model.quad_points
    S_full, Nz = model.tau_rayl.shape
    if spec_slice is None:
        ext, crop = slice(0, S_full), slice(0, S_full)
    else:
        ext, crop = parallel.raman_halo_slices(S_full, spec_slice, RS_type.i_lambda1lambda0)
    S = ext.stop - ext.start
    N = qp.Nquad * pol.n
    conv = array_type(arch)
    up = lambda x: conv(np.ascontiguousarray(np.asarray(x, dtype=FT)))
    dq = CR.device_quad(qp, pol, arch, FT)
    drs = device_rrs(RS_type, arch, FT)
    K = drs.K
    fscatt = RS_type.fscattRayl if RS_type.fscattRayl is not None else default_fscatt(model)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S_full))
        F0[0, :] = 1.0
    F0d = up(np.asarray(F0)[:, ext].T)
    dt, dev = CR._torch_dtype(FT), devi(arch)
    nV = len(model.vza)
    out = [torch.zeros((S, pol.n, nV), dtype=dt, device=dev) for _ in range(4)]
    R_SFI, T_SFI, ieR_SFI, ieT_SFI = out
    if S > 0:
        added = CR.make_added_layer(FT, arch, (N, N), S)
        added_surface = CR.make_added_layer(FT, arch, (N, N), S, shared=True)
        comp = CR.make_composite_layer(FT, arch, (N, N), S)
        added_rs = AddedLayerRS(FT, arch, K, N, S)
        surf_rs = AddedLayerRS(FT, arch, K, N, S)      # stays zero: the surface has no inelastic part
        comp_rs = CompositeLayerRS(FT, arch, K, N, S)
    for m in range(model.m_max + 1 if S > 0 else 0):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        Zpp_ie, Zmp_ie = H.compute_Z_moments(pol, qp.qp_mu, RS_type.greek_raman, m)   # computeRamanZλ! (:917-924)
        drs.Zpp, drs.Zmp = CR.to_device_matrix(Zpp_ie, arch, FT), CR.to_device_matrix(Zmp_ie, arch, FT)
        lods = H.constructCoreOpticalProperties(model, m)
        tags, tau_sum_all = H.extractEffectiveProps(lods, FT)
        # the reference dispatches interaction!(RS_type, scattering_interfaces_all[iz], ...) on these tags; only the
        # ScatteringInterface_11 inelastic method is built here (rt_kernel!(::RRS) hard-wires scatter = true, rt_kernel.jl:365)
        if any(t != "11" for t in tags):
            raise _lib.VSMError("rt_run(::RRS): a layer with max(tau*varpi) <= 2 eps gives interface tags %s; only "
                                "ScatteringInterface_11 is implemented for the inelastic interaction" % sorted(set(tags)))
        for iz, lo in enumerate(lods):
            tau_full = np.atleast_1d(lo.tau).astype(FT)
            varpi_full = np.broadcast_to(np.asarray(lo.varpi, dtype=FT), tau_full.shape)
            dtau_full, nd = H.get_dtau_ndoubl(tau_full, varpi_full, qp, FT, model.numerics)   # batch-global ndoubl
            Zpp, Zmp = np.asarray(lo.Zpp), np.asarray(lo.Zmp)
            if Zpp.ndim == 3:
                Zpp, Zmp = Zpp[ext], Zmp[ext]
            props = CR.DeviceLayerOptics(up(tau_full[ext]), up(varpi_full[ext]), CR.to_device_matrix(Zpp, arch, FT),
                                         CR.to_device_matrix(Zmp, arch, FT), float(np.max(tau_full * varpi_full)), tau_full,
                                         np.asarray(varpi_full))
            drs.fscatt = up(np.asarray(fscatt)[ext, iz])                                      # _expand_layer_rayleigh!
            expk = up(np.exp(-dtau_full[ext] / FT(qp.mu0)))                                   # rt_kernel.jl:367
            rt_kernel_rrs_(drs, pol, added, added_rs, comp, comp_rs, props, up(tau_sum_all[ext, iz]), m, dq, arch, iz + 1, F0d,
                           FT, model.numerics, dtau=up(dtau_full[ext]), ndoubl=nd, expk=expk, trace=trace)
        CR.create_surface_layer_(model.albedo, added_surface, m, dq, up(tau_sum_all[ext, -1]))
        interaction_inelastic_(drs, tags[-1], comp, comp_rs, added_surface, surf_rs)
        postprocessing_vza_rs_(pol, comp, comp_rs, model.vza, model.vaz, qp, m, float(weight), R_SFI, T_SFI, ieR_SFI, ieT_SFI)
    if device_out:
        return tuple(t[crop].contiguous() for t in out)
    synchronize_if_gpu()
    return tuple(to_host(t[crop]).transpose(2, 1, 0).copy() for t in out)


def rt_run_sharded(RS_type: RRS, model: H.RTModel, rank: int = 0, world: int = 1, dst: int = 0, executor=None):
    """rt_run(RS_type, model) over this rank's block of recipient points + one gather of R/T/ieR/ieT on `dst`.
    `executor(RS_type, model, spec_slice)` must return four (S_local, nStokes, nVZA) tensors (default: the HIP engine)."""
    from . import parallel
    S = model.tau_rayl.shape[0]
    sl = parallel.shard_slice(S, rank, world)
    if executor is None:
        executor = lambda rs, mdl, s: rt_run(rs, mdl, 1, spec_slice=s, device_out=True)
    parts = executor(RS_type, model, sl)
    gathered = [parallel.gather_spectral(t, S, rank, world, dst) for t in parts]
    if rank != dst:
        return (None,) * 4
    return tuple(g.detach().cpu().numpy().transpose(2, 1, 0).copy() for g in gathered)
