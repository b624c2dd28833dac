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
import dataclasses
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


REDUCE_M0 = True   # A/B switch (module attribute): False = the moment m = 0 with all Stokes components of the model

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
                   F0, FT, numerics, dtau=None, ndoubl=None, expk=None, trace=None, iface: str = "11"):
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
        interaction_inelastic_(drs, iface, comp, comp_rs, added, added_rs)


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


class SceneRRS:
    """Everything rt_run(RS_type::RRS, model, iBand) needs, resident in HBM (the Raman twin of CoreRT.Scene): the forward layer
    optics (tau, varpi, dtau, tau_sum, the component phase matrices and their per-point weights) come from CoreRT.Scene's device
    pass over the raw optical depths, uploaded once; the Raman phase matrices Z_λ₁λ₀(m) are computed on the device from
    `greek_raman` (computeRamanZλ!, inelastic_helper.jl:917-924 -> vsm_compute_Z_moments), fScattRayleigh [nSpec, Nz] is
    uploaded once, exp(-dtau/mu0) is formed per layer on the device.  `run()` only launches kernels.

    `spec_slice` = the recipient points this rank owns (multi-GPU): the scene covers the slice extended by a
    halo of max|i_λ₁λ₀| donor points on each side (`parallel.raman_halo_slices`), ndoubl comes from the FULL spectral axis,
    and only the owned points are returned -- no exchange step is needed."""

    def __init__(self, RS_type: RRS, model: H.RTModel, iBand: int = 1, spec_slice: Optional[slice] = None, _reduce_m0: bool = True):
        from . import parallel
        arch, FT = model.architecture, model.float_type
        CR._require_gpu(arch)
        if iBand != 1:
            raise _lib.VSMError("single-band models only (iBand = 1)")
        self.model, self.rs, self.arch, self.FT = model, RS_type, arch, FT
        pol, qp = model.polarization_type, model.quad_points
        S_full, Nz = model.tau_rayl.shape
        if spec_slice is None:
            ext, crop = slice(0, S_full), slice(0, S_full)
        else:
            ext, crop = parallel.raman_halo_slices(S_full, spec_slice, RS_type.i_lambda1lambda0)
        self.ext, self.crop = ext, crop
        # the surface of the Raman driver is the Lambertian albedo (rt_run.jl:455-463 with the model's brdf; only the scalar
        # Lambertian builder is wired here, like before)
        self.fwd = CR.Scene(model, ext, full_added_layer=True)
        self.fwd.compute_hdrf = False
        fwd = self.fwd
        S, N = fwd.S, fwd.N
        self.S, self.N = S, N
        conv = array_type(arch)
        dt, dev = CR._torch_dtype(FT), devi(arch)
        self.dt = dt
        self.drs = device_rrs(RS_type, arch, FT)
        K = self.drs.K
        fscatt = RS_type.fscattRayl if RS_type.fscattRayl is not None else default_fscatt(model)
        self.fscatt = conv(np.ascontiguousarray(np.asarray(fscatt, dtype=FT)[ext].T))          # (Nz, S): _expand_layer_rayleigh!
        g = RS_type.greek_raman
        tab = np.stack([np.asarray(getattr(g, k), dtype=np.float64) for k in ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")])
        gd = conv(np.ascontiguousarray(tab))
        q = fwd.dq.cstruct()
        self.Zie = []
        for m in range(model.m_max + 1):
            Zp = torch.empty((1, N, N), dtype=dt, device=dev)
            Zm = torch.empty_like(Zp)
            _lib.call("vsm_compute_Z_moments", dt, C.byref(q), m, tab.shape[1], CR._ptr(gd), CR._ptr(Zp), CR._ptr(Zm), CR._stream_ptr())
            self.Zie.append((Zp, Zm))
        md = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("vsm_stokes_coupling", dt, N, pol.n, 1, CR._ptr(self.Zie[0][0]), CR._ptr(self.Zie[0][1]), CR._ptr(md), CR._stream_ptr())
        zie_mask0 = int(md.cpu()[0])
        nV = len(model.vza)
        self.out = [fwd.R_SFI, fwd.T_SFI] + [torch.zeros((S, pol.n, nV), dtype=dt, device=dev) for _ in range(2)]
        self.expk = torch.empty(max(S, 1), dtype=dt, device=dev)
        if S > 0:
            self.added_rs = AddedLayerRS(FT, arch, K, N, S)
            self.surf_rs = AddedLayerRS(FT, arch, K, N, S)      # stays zero: the surface has no inelastic part
            self.comp_rs = CompositeLayerRS(FT, arch, K, N, S)
        # The Fourier moment m = 0 as a Stokes_IQ run.  No phase matrix -- Cabannes, Raman, aerosol -- couples (I,Q) with (U,V) at
        # m = 0 (compute_Z_matrices.jl:26-110: the T_l^m functions carry a factor m; CoreRT.Scene reads the exact zeros off the
        # device, `coupling`), the Lambertian surface reflects into I only, and a beam without U / V components drives nothing in the
        # (U,V) block: its source vectors stay zero through every recurrence, elastic and inelastic, so R / T / ieR / ieT of
        # m = 0 are those of the same model carried with two Stokes components (the reference's own Stokes_IQ), U = V = 0 -- at
        # (2/3)^3 of the products and (2/3)^2 of the bytes for Stokes_IQU.
        self.sub0 = None
        F0 = model.F0
        if (_reduce_m0 and REDUCE_M0 and pol.n >= 3 and S > 0 and (F0 is None or not np.any(np.asarray(F0)[2:] != 0))
                and fwd.coupling is not None and zie_mask0 is not None
                and not any((fwd.coupling[0] | zie_mask0) >> (4 * a + b) & 1 or (fwd.coupling[0] | zie_mask0) >> (4 * b + a) & 1
                            for a in (0, 1) for b in range(2, pol.n))):
            qi = H.QuadPoints(qp.mu0, qp.imu0, qp.qp_mu, qp.wt_mu, np.repeat(qp.qp_mu, 2), np.repeat(qp.wt_mu, 2), qp.Nquad, qp.Nstreams)
            m_iq = dataclasses.replace(model, polarization_type=H.Stokes_IQ(), quad_points=qi, m_max=0,
                                       F0=None if F0 is None else np.ascontiguousarray(np.asarray(F0)[:2]))
            self.sub0 = SceneRRS(RS_type, m_iq, iBand, spec_slice, _reduce_m0=False)

    def run(self, trace: Optional[list] = None):
        """rt_run.jl:383-517 for RS_type::RRS: Fourier loop -> rt_kernel!(::RRS) per layer -> surface -> interaction ->
        postprocessing_vza!(::RRS).  Returns the four (S_ext, nStokes, nVZA) device tensors R, T, ieR, ieT."""
        model, fwd, FT, dt = self.model, self.fwd, self.FT, self.dt
        pol, qp, drs = fwd.pol, fwd.qp, self.drs
        for t in self.out:
            t.zero_()
        if self.S == 0:
            return tuple(self.out)
        R_SFI, T_SFI, ieR_SFI, ieT_SFI = self.out
        added, comp = fwd.added, fwd.composite
        mu0 = C.c_double(qp.mu0) if dt == torch.float64 else C.c_float(qp.mu0)
        reduced = self.sub0 is not None and trace is None
        if reduced:      # m = 0 first, like the Fourier loop: its I, Q rows are copied in, the later moments accumulate on top
            for full, part in zip(self.out, self.sub0.run()):
                full[:, :2, :].copy_(part)
        for mom in fwd.moments:
            m = mom["m"]
            if m == 0 and reduced:
                continue           # (below: the Stokes_IQ run of this moment)
            weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
            drs.Zpp, drs.Zmp = self.Zie[m]
            # the reference dispatches interaction!(RS_type, scattering_interfaces_all[iz], ...) on the tags of
            # extractEffectiveProps (rt_kernel!(::RRS) itself hard-wires scatter = true, rt_kernel.jl:365)
            for iz, ly in enumerate(mom["layers"]):
                drs.fscatt = self.fscatt[iz]
                _lib.call("vsm_layer_expk", dt, self.S, CR._ptr(ly["dtau"]), mu0, CR._ptr(self.expk), CR._stream_ptr())   # rt_kernel.jl:367
                rt_kernel_rrs_(drs, pol, added, self.added_rs, comp, self.comp_rs, ly["props"].materialize(), ly["tau_sum"], m, fwd.dq,
                               self.arch, iz + 1, fwd.F0, FT, model.numerics, dtau=ly["dtau"], ndoubl=ly["nd"], expk=self.expk,
                               trace=trace, iface=ly["iface"])
            CR.create_surface_layer_(model.albedo, fwd.added_surface, m, fwd.dq, mom["tau_sum_surface"])
            interaction_inelastic_(drs, mom["iface_surface"], comp, self.comp_rs, fwd.added_surface, self.surf_rs)
            postprocessing_vza_rs_(pol, comp, self.comp_rs, model.vza, model.vaz, qp, m, float(weight), R_SFI, T_SFI, ieR_SFI, ieT_SFI)
        return tuple(self.out)

    def results_device(self):
        """The owned points' (S_local, nStokes, nVZA) tensors (for the gather)."""
        return tuple(t[self.crop].contiguous() for t in self.out)

    def results_host(self):
        return tuple(to_host(t[self.crop]).transpose(2, 1, 0).copy() for t in self.out)


def raman_bytes_per_point(N: int, K: int, itemsize: int) -> int:
    """Device bytes one spectral point of a SceneRRS holds: the 12 + 10 four-dimensional inelastic arrays of the added / surface /
    composite layers and of the kernels' work space (N x N x K each), the elastic layers and their work space."""
    return int(itemsize * (22 * K * N * N + 8 * K * N + 40 * N * N))


def rt_run(RS_type: RRS, model: H.RTModel, iBand: int = 1, trace: Optional[list] = None, spec_slice: Optional[slice] = None,
           device_out: bool = False, max_points: Optional[int] = None):
    """rt_run(RS_type::RRS, model, iBand) (rt_run.jl:238-535): returns (R_SFI, T_SFI, ieR_SFI, ieT_SFI) as host arrays
    [nVZA, nStokes, nSpec].  `model.greek_rayleigh` must hold the Cabannes phase matrix and `model.varpi_Cabannes`
    the elastic Rayleigh single-scattering albedo (compEffectiveLayerProperties.jl:36-41).  `spec_slice`: see SceneRRS;
    `device_out=True` returns the four (S_local, nStokes, nVZA) device tensors instead (for the gather).

    Footprint: the reference pages its N x N x nSpec x nRaman arrays through the host (interaction_inelastic.jl:16-35,427-438).
    Here a run whose arrays would not fit the device (or `max_points` recipient points per pass, if given) walks the recipient
    axis in blocks, each a SceneRRS over the block extended by the halo of max|shift| donor points -- the same construction as
    the multi-GPU shards (no exchange: a recipient reads only ELASTIC fields of its donors), bit-identical to the one-pass run,
    at the price of recomputing 2 max|shift| points per block."""
    S_full = model.tau_rayl.shape[0]
    lo, hi, _ = (spec_slice or slice(0, S_full)).indices(S_full)
    n_own = max(hi - lo, 0)
    if max_points is None and n_own > 0:
        N = model.quad_points.Nquad * model.polarization_type.n
        K = len(np.asarray(RS_type.i_lambda1lambda0).ravel())
        per = raman_bytes_per_point(N, K, np.dtype(model.float_type).itemsize)
        if REDUCE_M0 and model.polarization_type.n >= 3:   # + the Stokes_IQ scene of the moment m = 0
            per += raman_bytes_per_point(2 * model.quad_points.Nquad, K, np.dtype(model.float_type).itemsize)
        free, _total = torch.cuda.mem_get_info()
        halo = max([abs(int(x)) for x in np.asarray(RS_type.i_lambda1lambda0).ravel() if abs(int(x)) < S_full] or [0])
        fit = int(0.8 * free // per) - 2 * halo
        if fit < n_own:
            if fit < 1:
                raise _lib.VSMError("rt_run (RRS): not even one block of recipient points fits the device (%d bytes per point, "
                                    "halo %d)" % (per, halo))
            max_points = fit
    if max_points is None or max_points >= n_own or trace is not None:
        scene = SceneRRS(RS_type, model, iBand, spec_slice)
        scene.run(trace)
        if device_out:
            return scene.results_device()
        synchronize_if_gpu()
        _lib.check_device_status("rt_run (RRS)")
        return scene.results_host()
    parts = []
    for b0 in range(lo, hi, int(max_points)):
        scene = SceneRRS(RS_type, model, iBand, slice(b0, min(b0 + int(max_points), hi)))
        scene.run()
        parts.append(scene.results_device() if device_out else scene.results_host())
        del scene
        torch.cuda.empty_cache()
    if device_out:
        return tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(4))
    return tuple(np.concatenate([p[i] for p in parts], axis=2) for i in range(4))


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
