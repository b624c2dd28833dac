"""Linearized (Jacobian) CoreRT host layer: rt_run(model, lin_model, NAer, NGas, NSurf)
(src/CoreRT/rt_run_lin.jl:72-78, :102-326) on the MI355X, through the C ABI.

Mirrors: make_added_layer / make_composite_layer (LinMode) (tools/rt_helper_functions_lin.jl:15-80),
elemental! / doubling_allparams! / interaction! (lin) (CoreKernel/*_lin.jl), create_surface_layer! (lin),
postprocessing_vza! (lin).  This round the linearized kernels are operator-level (batched MFMA products
over (spectral point, parameter)); inverses are shared by all parameters like in the reference.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import host_model as H
from . import core_rt as CR
from .architectures import array_type, devi, synchronize_if_gpu, to_host


class AddedLayerLin:
    """types_lin.jl:20-60 AddedLayerLin (`ap_*` all-parameter fields)."""

    def __init__(self, FT, arch, P, N, nSpec, shared=False):
        dev, dt = devi(arch), CR._torch_dtype(FT)
        sm = 1 if shared else nSpec
        z = lambda: torch.zeros((P, sm, N, N), dtype=dt, device=dev)
        self.ap_r_mp, self.ap_t_pp, self.ap_r_pm, self.ap_t_mm = z(), z(), z(), z()
        self.ap_J0_p = torch.zeros((P, nSpec, N), dtype=dt, device=dev)
        self.ap_J0_m = torch.zeros((P, nSpec, N), dtype=dt, device=dev)
        self.P, self.N, self.nSpec, self.shared, self.dtype = P, N, nSpec, shared, dt

    def cstruct(self):
        return _lib.vsm_added_lin(self.ap_r_mp.data_ptr(), self.ap_t_pp.data_ptr(), self.ap_r_pm.data_ptr(),
                                  self.ap_t_mm.data_ptr(), self.ap_J0_p.data_ptr(), self.ap_J0_m.data_ptr(), self.P, 0,
                                  0 if self.shared else self.N * self.N)


class CompositeLayerLin:
    """types_lin.jl:62-97 CompositeLayerLin."""

    def __init__(self, FT, arch, P, N, nSpec):
        dev, dt = devi(arch), CR._torch_dtype(FT)
        z = lambda: torch.zeros((P, nSpec, N, N), dtype=dt, device=dev)
        self.R_mp, self.R_pm, self.T_pp, self.T_mm = z(), z(), z(), z()
        self.J0_p = torch.zeros((P, nSpec, N), dtype=dt, device=dev)
        self.J0_m = torch.zeros((P, nSpec, N), dtype=dt, device=dev)
        self.P, self.N, self.nSpec, self.dtype = P, N, nSpec, dt

    def cstruct(self):
        return _lib.vsm_composite_lin(self.R_mp.data_ptr(), self.R_pm.data_ptr(), self.T_pp.data_ptr(),
                                      self.T_mm.data_ptr(), self.J0_p.data_ptr(), self.J0_m.data_ptr(), self.P, 0)


def to_device_sp(x: np.ndarray, arch, FT) -> torch.Tensor:
    """[S, p] host array -> device tensor in the reference's [S, p] column-major order (shape (p, S))."""
    return array_type(arch)(np.ascontiguousarray(np.asarray(x, dtype=FT).T))


def to_device_zdot(Zd: Optional[np.ndarray], arch, FT):
    """[p,N,N] or [p,S,N,N] math order -> layout tensor + (stride_s, stride_p) in elements."""
    if Zd is None:
        return None, 0, 0
    Zd = np.asarray(Zd, dtype=FT)
    if Zd.ndim == 3:
        t = array_type(arch)(np.ascontiguousarray(Zd.transpose(0, 2, 1)))
        N = Zd.shape[-1]
        return t, 0, N * N
    t = array_type(arch)(np.ascontiguousarray(Zd.transpose(0, 1, 3, 2)))
    P, S, N, _ = Zd.shape
    return t, N * N, N * N * S


def elemental_lin_(pol, tau_sum, tau_sum_dot, dtau, dtau_dot, F0, props: CR.DeviceLayerOptics, varpi_dot, Zpp_dot,
                   Zmp_dot, zd_strides, p_layer, m, ndoubl, dq: CR.DeviceQuad, added: CR.AddedLayer, added_lin: AddedLayerLin):
    """elemental! (lin): elemental_lin.jl:77-206."""
    q, a, al = dq.cstruct(), added.cstruct(), added_lin.cstruct()
    _lib.call("vsm_elemental_lin", added.dtype, C.byref(q), added.nSpec, m, ndoubl, CR._ptr(dtau), CR._ptr(props.varpi),
              CR._ptr(tau_sum), CR._ptr(F0), CR._ptr(props.Zpp), CR._ptr(props.Zmp), props.z_stride, p_layer,
              CR._ptr(dtau_dot), CR._ptr(varpi_dot), CR._ptr(tau_sum_dot), CR._ptr(Zpp_dot), CR._ptr(Zmp_dot),
              zd_strides[0], zd_strides[1], C.byref(a), C.byref(al), CR._stream_ptr())


_lin_work = {}


def _work(kind, n, dtype, device):
    key = (kind, dtype, str(device))
    w = _lin_work.get(key)
    if w is None or w.numel() < n:
        w = _lib.poison(torch.empty(int(n), dtype=dtype, device=device))
        _lin_work[key] = w
    return w


def doubling_allparams_(pol, expk, ndoubl, added: CR.AddedLayer, added_lin: AddedLayerLin, dtau_dot_all, mu0, N_active):
    """doubling_allparams! (doubling_lin.jl:216-339)."""
    n = _lib.lib().vsm_doubling_lin_work_elems(added.N, added.nSpec, max(N_active, 1) if N_active else added_lin.P)
    work = _work("dbl", max(int(n), 1), added.dtype, added.r_mp.device)
    a, al = added.cstruct(), added_lin.cstruct()
    mu = C.c_double(mu0) if added.dtype == torch.float64 else C.c_float(mu0)
    _lib.call("vsm_doubling_lin", added.dtype, added.N, pol.n, added.nSpec, ndoubl, CR._ptr(expk), CR._ptr(dtau_dot_all),
              mu, N_active, C.byref(a), C.byref(al), CR._ptr(work), CR._stream_ptr())


def interaction_lin_(scattering_interface, comp: CR.CompositeLayer, comp_lin: CompositeLayerLin, added: CR.AddedLayer,
                     added_lin: AddedLayerLin):
    """interaction! (lin) (interaction_lin.jl:337-351)."""
    n = _lib.lib().vsm_interaction_lin_work_elems(comp.N, comp.nSpec, comp_lin.P)
    work = _work("ia", int(n), comp.dtype, comp.R_mp.device)
    c, cl, a, al = comp.cstruct(), comp_lin.cstruct(), added.cstruct(), added_lin.cstruct()
    _lib.call("vsm_interaction_lin", comp.dtype, CR.IFACE[scattering_interface], comp.N, comp.nSpec, C.byref(c),
              C.byref(cl), C.byref(a), C.byref(al), CR._ptr(work), CR._stream_ptr())


def rt_run_lin(model: H.RTModel, lin_model: H.LinModel, NAer: int, NGas: int, NSurf: int):
    """rt_run(model, lin_model, NAer, NGas, NSurf) -> (R, T, Rdot, Tdot); Rdot/Tdot: [nVZA, nStokes, nSpec, Nparams].
    Supported this round: NAer = 0, NGas = len(lin_model.tau_abs_dot), NSurf = 1 (Lambertian albedo)."""
    if NAer != 0 or NSurf != 1 or NGas != len(lin_model.tau_abs_dot):
        raise _lib.VSMError("rt_run (linearized): only NAer=0, NGas=len(lin_model.tau_abs_dot), NSurf=1 are wired up")
    arch, FT = model.architecture, model.float_type
    CR._require_gpu(arch)
    pol, qp = model.polarization_type, model.quad_points
    layout = H.ParameterLayout(n_aerosols=NAer, n_gases=NGas, n_surface=NSurf)
    P, pl = layout.n_total, layout.n_layer_params
    S, Nz = model.tau_rayl.shape
    N = qp.Nquad * pol.n
    nV = len(model.vza)
    conv = array_type(arch)
    dt, dev = CR._torch_dtype(FT), devi(arch)
    dq = CR.device_quad(qp, pol, arch, FT)
    F0 = model.F0
    if F0 is None:
        F0 = np.zeros((pol.n, S))
        F0[0, :] = 1.0
    F0d = conv(np.ascontiguousarray(np.asarray(F0, dtype=FT).T))
    added, added_s = CR.AddedLayer(FT, arch, N, S), CR.AddedLayer(FT, arch, N, S, shared=True)
    comp = CR.CompositeLayer(FT, arch, N, S)
    al, als = AddedLayerLin(FT, arch, P, N, S), AddedLayerLin(FT, arch, P, N, S, shared=True)
    cl = CompositeLayerLin(FT, arch, P, N, S)
    R = torch.zeros((S, pol.n, nV), dtype=dt, device=dev)
    T = torch.zeros_like(R)
    Rd = torch.zeros((P, S, pol.n, nV), dtype=dt, device=dev)
    Td = torch.zeros_like(Rd)
    for m in range(model.m_max + 1):
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        lods = H.constructCoreOpticalProperties(model, m)
        lins = H.constructCoreOpticalPropertiesLin(model, lin_model, lods)
        tags, tau_sum_all = H.extractEffectiveProps(lods, FT)
        if any(t != "11" for t in tags):
            raise _lib.VSMError("rt_run (linearized): every layer must scatter (rt_kernel_lin.jl:87 hard-codes scatter=true)")
        tsd = np.zeros((S, pl, Nz + 1))
        for iz in range(Nz):
            tsd[:, :, iz + 1] = tsd[:, :, iz] + lins[iz].tau_dot
        for iz in range(Nz):
            props = CR.expandOpticalProperties(lods[iz], arch, FT)
            dtau_h, nd = H.get_dtau_ndoubl(props.tau_h, props.varpi_h, qp, FT, model.numerics)
            dtau = conv(dtau_h)
            expk = conv(np.exp(-dtau_h / FT(qp.mu0)).astype(FT))
            dtd = lins[iz].tau_dot / FT(2 ** nd)
            zpd, zs_, zp_ = to_device_zdot(lins[iz].Zpp_dot, arch, FT)
            zmd, _, _ = to_device_zdot(lins[iz].Zmp_dot, arch, FT)
            elemental_lin_(pol, conv(tau_sum_all[:, iz].astype(FT)), to_device_sp(tsd[:, :, iz], arch, FT), dtau,
                           to_device_sp(dtd, arch, FT), F0d, props, to_device_sp(lins[iz].varpi_dot, arch, FT), zpd, zmd,
                           (zs_, zp_), pl, m, nd, dq, added, al)
            dall = np.zeros((S, P))
            dall[:, :pl] = dtd
            doubling_allparams_(pol, expk, nd, added, al, to_device_sp(dall, arch, FT), qp.mu0, pl)
            if iz == 0:
                CR.copy_added_to_composite_(comp, added)
                a_, c_ = al.cstruct(), cl.cstruct()
                _lib.call("vsm_copy_added_to_composite_lin", dt, N, S, C.byref(a_), C.byref(c_), CR._stream_ptr())
            else:
                interaction_lin_(tags[iz], comp, cl, added, al)
        # surface
        q_, a_, al_ = dq.cstruct(), added_s.cstruct(), als.cstruct()
        alb = C.c_double(model.albedo) if dt == torch.float64 else C.c_float(model.albedo)
        ts_surf = conv(tau_sum_all[:, -1].astype(FT))           # keep the device buffers alive across the launch
        tsd_surf = to_device_sp(tsd[:, :, -1], arch, FT)
        _lib.call("vsm_lambertian_surface_lin", dt, C.byref(q_), S, m, alb, layout.surface_index(0),
                  CR._ptr(ts_surf), CR._ptr(tsd_surf), pl, CR._ptr(F0d), C.byref(a_), C.byref(al_), CR._stream_ptr())
        interaction_lin_(tags[-1], comp, cl, added_s, als)
        CR.postprocessing_vza_(pol, comp, model.vza, model.vaz, qp, m, float(weight), R, T)
        n = pol.n
        row0 = (C.c_int * nV)()
        ctype = C.c_double if dt == torch.float64 else C.c_float
        w = (ctype * (nV * n))()
        for v in range(nV):
            imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(H.cosd(model.vza[v])))))
            row0[v] = imu * n
            c0, s0 = H.cosd(m * model.vaz[v]), H.sind(m * model.vaz[v])
            for k in range(n):
                w[v + nV * k] = float(weight) * [c0, c0, s0, s0][k]
        _lib.call("vsm_postprocess_vza_lin", dt, N, n, S, nV, P, row0, w, CR._ptr(cl.J0_m), CR._ptr(cl.J0_p), CR._ptr(Rd),
                  CR._ptr(Td), CR._stream_ptr())
    synchronize_if_gpu()
    tr = lambda t: to_host(t).transpose(2, 1, 0).copy()
    tr4 = lambda t: to_host(t).transpose(3, 2, 1, 0).copy()
    return tr(R), tr(T), tr4(Rd), tr4(Td)
