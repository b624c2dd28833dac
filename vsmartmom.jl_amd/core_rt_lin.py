"""Linearized (Jacobian) CoreRT host layer: rt_run(model, lin_model, NAer, NGas, NSurf)
(src/CoreRT/rt_run_lin.jl:72-78, :102-326) on the MI355X, through the C ABI.

Mirrors: make_added_layer / make_composite_layer (LinMode) (tools/rt_helper_functions_lin.jl:15-80),
elemental! / doubling_allparams! / interaction! (lin) (CoreKernel/*_lin.jl), create_surface_layer! (lin),
postprocessing_vza! (lin).  FP64 with 32 < N <= 60 runs the fused column-strip kernels (vsm_striplin.hip: one launch
per doubling step, two per interaction); other shapes run operator level (batched MFMA products over (spectral point,
parameter)); inverses are shared by all parameters like in the reference.  `SceneLin` keeps every input in HBM.
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


def _pp_args(pol, qp, vza, vaz, m, weight, dtype):
    """row0 / weights of postprocessing_vza! (postprocessing_vza.jl:23-94) as ctypes arrays."""
    n, nV = pol.n, len(vza)
    row0 = (C.c_int * nV)()
    ctype = C.c_double if dtype == torch.float64 else C.c_float
    w = (ctype * (nV * n))()
    for v in range(nV):
        imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(H.cosd(vza[v])))))
        row0[v] = imu * n
        c0, s0 = H.cosd(m * vaz[v]), H.sind(m * vaz[v])
        for k in range(n):
            w[v + nV * k] = float(weight) * [c0, c0, s0, s0][k]
    return row0, w


class SceneLin:
    """Everything rt_run(model, lin_model, NAer, NGas, NSurf) needs, resident in HBM (the linearized twin of
    CoreRT.Scene): per Fourier moment and layer tau, varpi, dtau, exp(-dtau/mu0), tau_sum and their parameter
    derivatives, Z (shared) and Zdot, interface tags and ndoubl -- built once from host numpy, then `run()` only
    launches kernels.  `spec_slice` selects this rank's spectral shard; ndoubl and the tags always come from the FULL
    spectral axis (rt_kernel_lin.jl:87-95 uses batch-global maxima like the forward kernel).

    Parameter slots (parameter_layout.jl:28-56): 7 per aerosol (tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p; their optics
    derivatives are inputs: LinModel.tau_aer_dot / lin_aerosol_optics), the gases, then ONE surface slot -- the Lambertian albedo
    (lambertian_surface_lin.jl:48-162) or the Cox-Munk wind speed (coxmunk_surface_lin.jl:27-102)."""

    def __init__(self, model: H.RTModel, lin_model: H.LinModel, NAer: int, NGas: int, NSurf: int,
                 spec_slice: Optional[slice] = None):
        if NAer != lin_model.n_aer or NAer != len(model.aerosol_optics) or NSurf != 1 or NGas != len(lin_model.tau_abs_dot):
            raise _lib.VSMError("rt_run (linearized): NAer must equal the aerosols of model and lin_model, NGas = "
                                "len(lin_model.tau_abs_dot), NSurf = 1")
        if not isinstance(model.surface, (H.LambertianSurfaceScalar, H.CoxMunkSurface)):
            raise _lib.VSMError("rt_run (linearized): surface %r has no linearized builder here" % (model.surface,))
        arch, FT = model.architecture, model.float_type
        CR._require_gpu(arch)
        self.model, self.arch, self.FT = model, arch, FT
        pol, qp = model.polarization_type, model.quad_points
        self.pol, self.qp = pol, qp
        self.layout = H.ParameterLayout(n_aerosols=NAer, n_gases=NGas, n_surface=NSurf)
        P, pl = self.layout.n_total, self.layout.n_layer_params
        self.P, self.pl = P, pl
        S_full, Nz = model.tau_rayl.shape
        self.sl = spec_slice if spec_slice is not None else slice(0, S_full)
        sl = self.sl
        S = self.S = len(range(*sl.indices(S_full)))
        N = self.N = qp.Nquad * pol.n
        nV = len(model.vza)
        conv = array_type(arch)
        dt, dev = CR._torch_dtype(FT), devi(arch)
        self.dt = dt
        self.dq = CR.device_quad(qp, pol, arch, FT)
        F0 = model.F0
        if F0 is None:
            F0 = np.zeros((pol.n, S_full))
            F0[0, :] = 1.0
        self.F0 = conv(np.ascontiguousarray(np.asarray(F0, dtype=FT)[:, sl].T))
        cut = lambda x: np.ascontiguousarray(np.asarray(x)[sl])
        self.moments = []
        shared = None     # tau/varpi/derivative tensors do not depend on m when no aerosol is mixed in: upload once
        for m in range(model.m_max + 1):
            lods = H.constructCoreOpticalProperties(model, m)
            lins = H.constructCoreOpticalPropertiesLin(model, lin_model, lods, m)
            tags, tau_sum_all = H.extractEffectiveProps(lods, FT)
            # rt_kernel_lin.jl:87 hard-codes scatter = true: a layer with tau*varpi <= 2 eps still goes through elemental! and
            # doubling! (ndoubl = 0), only its interaction follows the 00 / 01 / 10 tag of extractEffectiveProps
            tsd = np.zeros((S_full, pl, Nz + 1))
            for iz in range(Nz):
                tsd[:, :, iz + 1] = tsd[:, :, iz] + lins[iz].tau_dot
            reuse = shared is not None and not model.aerosol_optics
            layers = []
            for iz in range(Nz):
                lo = lods[iz]
                tau_full = np.atleast_1d(lo.tau).astype(FT)
                varpi_full = np.broadcast_to(np.asarray(lo.varpi, dtype=FT), tau_full.shape)
                Zpp, Zmp = CR.to_device_matrix(lo.Zpp, arch, FT), CR.to_device_matrix(lo.Zmp, arch, FT)
                if Zpp.shape[0] != 1:
                    Zpp, Zmp = Zpp[sl].contiguous(), Zmp[sl].contiguous()
                if reuse:
                    ly = dict(shared[iz])
                    ly["props"] = CR.DeviceLayerOptics(ly["props"].tau, ly["props"].varpi, Zpp, Zmp, ly["props"].max_tau_varpi,
                                                       tau_full, np.asarray(varpi_full))
                    layers.append(ly)
                    continue
                dtau_h, nd = H.get_dtau_ndoubl(tau_full, varpi_full, qp, FT, model.numerics)
                dtd = lins[iz].tau_dot / FT(2 ** nd)
                dall = np.zeros((S_full, P))
                dall[:, :pl] = dtd
                zpd, zs_, zp_ = to_device_zdot(lins[iz].Zpp_dot, arch, FT)
                zmd, _, _ = to_device_zdot(lins[iz].Zmp_dot, arch, FT)
                props = CR.DeviceLayerOptics(conv(cut(tau_full)), conv(cut(varpi_full)), Zpp, Zmp,
                                             float(np.max(tau_full * varpi_full)), tau_full, np.asarray(varpi_full))
                layers.append(dict(props=props, nd=nd, iface=tags[iz], dtau=conv(cut(dtau_h)),
                                   expk0=conv(cut(np.exp(-dtau_h / FT(qp.mu0)).astype(FT))),
                                   dtau_dot=to_device_sp(cut(dtd), arch, FT), dall=to_device_sp(cut(dall), arch, FT),
                                   varpi_dot=to_device_sp(cut(lins[iz].varpi_dot), arch, FT),
                                   tau_sum=conv(cut(tau_sum_all[:, iz].astype(FT))),
                                   tau_sum_dot=to_device_sp(cut(tsd[:, :, iz]), arch, FT), zpd=zpd, zmd=zmd, zds=(zs_, zp_)))
            if shared is None:
                shared = layers
            rho = drho = None
            if isinstance(model.surface, H.CoxMunkSurface):
                rho, drho = CR.reflectance(model.surface, self.dq, m, arch, FT, deriv=True)
            surf = (shared_surf if reuse else
                    dict(tau_sum=conv(cut(tau_sum_all[:, -1].astype(FT))), tau_sum_dot=to_device_sp(cut(tsd[:, :, -1]), arch, FT)))
            shared_surf = surf
            self.moments.append(dict(m=m, layers=layers, iface_surface=tags[-1], rho=rho, drho=drho, **surf))
        self.added, self.added_s = CR.AddedLayer(FT, arch, N, S), CR.AddedLayer(FT, arch, N, S, shared=True)
        self.comp = CR.CompositeLayer(FT, arch, N, S)
        self.al, self.als = AddedLayerLin(FT, arch, P, N, S), AddedLayerLin(FT, arch, P, N, S, shared=True)
        self.cl = CompositeLayerLin(FT, arch, P, N, S)
        self.expk = torch.empty(max(S, 1), dtype=dt, device=dev)
        self.R = torch.zeros((S, pol.n, nV), dtype=dt, device=dev)
        self.T = torch.zeros_like(self.R)
        self.Rd = torch.zeros((P, S, pol.n, nV), dtype=dt, device=dev)
        self.Td = torch.zeros_like(self.Rd)

    def run(self):
        """The device-resident part of rt_run_lin.jl:200-322: Fourier loop -> layers -> surface -> post-processing."""
        model, pol, qp, FT, dt = self.model, self.pol, self.qp, self.FT, self.dt
        N, S, P, pl = self.N, self.S, self.P, self.pl
        for t in (self.R, self.T, self.Rd, self.Td):
            t.zero_()
        if S == 0:
            return self.R, self.T, self.Rd, self.Td
        added, al, comp, cl = self.added, self.al, self.comp, self.cl
        isurf = self.layout.surface_index(0)
        for mom in self.moments:
            m = mom["m"]
            weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
            for iz, ly in enumerate(mom["layers"]):
                elemental_lin_(pol, ly["tau_sum"], ly["tau_sum_dot"], ly["dtau"], ly["dtau_dot"], self.F0, ly["props"],
                               ly["varpi_dot"], ly["zpd"], ly["zmd"], ly["zds"], pl, m, ly["nd"], self.dq, added, al)
                self.expk[:S].copy_(ly["expk0"])      # doubling! squares exp(-dtau/mu0) in place
                doubling_allparams_(pol, self.expk, ly["nd"], added, al, ly["dall"], qp.mu0, pl)
                if iz == 0:
                    CR.copy_added_to_composite_(comp, added)
                    a_, c_ = al.cstruct(), cl.cstruct()
                    _lib.call("vsm_copy_added_to_composite_lin", dt, N, S, C.byref(a_), C.byref(c_), CR._stream_ptr())
                else:
                    interaction_lin_(ly["iface"], comp, cl, added, al)
            q_, a_, al_ = self.dq.cstruct(), self.added_s.cstruct(), self.als.cstruct()
            if isinstance(model.surface, H.CoxMunkSurface):
                _lib.call("vsm_brdf_surface_lin", dt, C.byref(q_), S, m, CR._ptr(mom["rho"]), CR._ptr(mom["drho"]), isurf,
                          CR._ptr(mom["tau_sum"]), CR._ptr(mom["tau_sum_dot"]), pl, CR._ptr(self.F0), C.byref(a_), C.byref(al_),
                          CR._stream_ptr())
            else:
                alb = C.c_double(model.surface.albedo) if dt == torch.float64 else C.c_float(model.surface.albedo)
                _lib.call("vsm_lambertian_surface_lin", dt, C.byref(q_), S, m, alb, isurf, CR._ptr(mom["tau_sum"]),
                          CR._ptr(mom["tau_sum_dot"]), pl, CR._ptr(self.F0), C.byref(a_), C.byref(al_), CR._stream_ptr())
            interaction_lin_(mom["iface_surface"], comp, cl, self.added_s, self.als)
            CR.postprocessing_vza_(pol, comp, model.vza, model.vaz, qp, m, float(weight), self.R, self.T)
            row0, w = _pp_args(pol, qp, model.vza, model.vaz, m, weight, dt)
            _lib.call("vsm_postprocess_vza_lin", dt, N, pol.n, S, len(model.vza), P, row0, w, CR._ptr(cl.J0_m), CR._ptr(cl.J0_p),
                      CR._ptr(self.Rd), CR._ptr(self.Td), CR._stream_ptr())
        return self.R, self.T, self.Rd, self.Td

    def results_host(self):
        """(R, T, Rdot, Tdot) as the reference returns them: [nVZA, nStokes, nSpec] and [nVZA, nStokes, nSpec, Nparams]."""
        tr = lambda t: to_host(t).transpose(2, 1, 0).copy()
        tr4 = lambda t: to_host(t).transpose(3, 2, 1, 0).copy()
        return tr(self.R), tr(self.T), tr4(self.Rd), tr4(self.Td)

    def flops_per_point(self) -> float:
        """ALGORITHMIC flops per spectral point: forward + per parameter nd (24N^3+16N^2) per layer for
        the layer parameters and (48N^3+16N^2) per interaction for every parameter."""
        N = float(self.N)
        tot = 0.0
        for mom in self.moments:
            for iz, ly in enumerate(mom["layers"]):
                tot += ly["nd"] * ((12 + 24 * self.pl) * N ** 3 + (8 + 16 * self.pl) * N ** 2)
                if iz > 0:
                    tot += (24 + 48 * self.P) * N ** 3 + (8 + 16 * self.P) * N ** 2
            tot += (24 + 48 * self.P) * N ** 3 + (8 + 16 * self.P) * N ** 2
        return tot


def prepare_scene_lin(model, lin_model, NAer, NGas, NSurf, spec_slice: Optional[slice] = None) -> SceneLin:
    return SceneLin(model, lin_model, NAer, NGas, NSurf, spec_slice)


def rt_run_lin(model: H.RTModel, lin_model: H.LinModel, NAer: int, NGas: int, NSurf: int):
    """rt_run(model, lin_model, NAer, NGas, NSurf) (rt_run_lin.jl:72-78 -> :102-326) -> (R, T, Rdot, Tdot);
    Rdot/Tdot: [nVZA, nStokes, nSpec, Nparams].  Supported: NAer = number of aerosols of the model (7 slots each, optics
    derivatives supplied in lin_model), NGas = len(lin_model.tau_abs_dot), NSurf = 1
    (Lambertian albedo or Cox-Munk wind speed, by the model's surface).  Like the reference's linearized driver this
    path applies no TMS correction."""
    scene = SceneLin(model, lin_model, NAer, NGas, NSurf)
    scene.run()
    synchronize_if_gpu()
    return scene.results_host()
