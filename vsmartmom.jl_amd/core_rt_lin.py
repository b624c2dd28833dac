"""Linearized (Jacobian) CoreRT host layer: rt_run(model, lin_model, NAer, NGas, NSurf)
(src/CoreRT/rt_run_lin.jl:72-78, :102-326) on the MI355X, through the C ABI.

Mirrors: make_added_layer / make_composite_layer (LinMode) (tools/rt_helper_functions_lin.jl:15-80),
elemental! / doubling_allparams! / interaction! (lin) (CoreKernel/*_lin.jl), create_surface_layer! (lin),
postprocessing_vza! (lin).  FP64 with 8 <= N <= 60 runs the fused column-strip kernels (vsm_striplin.hip: one launch
per doubling step, two per interaction); other shapes run operator level (batched MFMA products over (spectral point,
parameter)); inverses are shared by all parameters like in the reference.  `SceneLin` keeps every input in HBM.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import dataclasses
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


class _CompositeAsAdded:
    """A composite of LOWER layers handed to interaction! as its added layer: the adding equations are associative and the sources
    of every layer carry their absolute attenuation from the top of the atmosphere, so composites of sub-columns combine like layers
    (SceneLin: the interactions of a small batch as a tree).  All four matrices are in memory (no D-symmetry is assumed)."""

    def __init__(self, comp: CR.CompositeLayer):
        self.comp, self.N, self.nSpec, self.dtype = comp, comp.N, comp.nSpec, comp.dtype

    def cstruct(self):
        c = self.comp.cstruct()
        return _lib.vsm_added(c.R_mp, c.T_pp, c.R_pm, c.T_mm, c.J0_p, c.J0_m, self.N * self.N, 0, 0)


class _CompositeLinAsAdded:
    def __init__(self, cl: "CompositeLayerLin"):
        self.cl, self.P = cl, cl.P

    def cstruct(self):
        c = self.cl
        return _lib.vsm_added_lin(c.R_mp.data_ptr(), c.T_pp.data_ptr(), c.R_pm.data_ptr(), c.T_mm.data_ptr(), c.J0_p.data_ptr(),
                                  c.J0_m.data_ptr(), c.P, 0, c.N * c.N)


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
                   Zmp_dot, zd_strides, p_layer, m, ndoubl, dq: CR.DeviceQuad, added: CR.AddedLayer, added_lin: AddedLayerLin,
                   n_m0: int = 0):
    """elemental! (lin): elemental_lin.jl:77-206.  n_m0 > 0: a batch with Fourier moments folded into it whose first n_m0 points are
    (moment 0, point) pairs and the others pairs of moments m > 0 (vsm_elemental_lin_fold)."""
    q, a, al = dq.cstruct(), added.cstruct(), added_lin.cstruct()
    if n_m0:
        _lib.call("vsm_elemental_lin_fold", added.dtype, C.byref(q), added.nSpec, m, int(n_m0), ndoubl, CR._ptr(dtau),
                  CR._ptr(props.varpi), CR._ptr(tau_sum), CR._ptr(F0), CR._ptr(props.Zpp), CR._ptr(props.Zmp), props.z_stride, p_layer,
                  CR._ptr(dtau_dot), CR._ptr(varpi_dot), CR._ptr(tau_sum_dot), C.byref(a), C.byref(al), CR._stream_ptr())
        return
    _lib.call("vsm_elemental_lin", added.dtype, C.byref(q), added.nSpec, m, ndoubl, CR._ptr(dtau), CR._ptr(props.varpi),
              CR._ptr(tau_sum), CR._ptr(F0), CR._ptr(props.Zpp), CR._ptr(props.Zmp), props.z_stride, p_layer,
              CR._ptr(dtau_dot), CR._ptr(varpi_dot), CR._ptr(tau_sum_dot), CR._ptr(Zpp_dot), CR._ptr(Zmp_dot),
              zd_strides[0], zd_strides[1], C.byref(a), C.byref(al), CR._stream_ptr())


_lin_work = {}
_lane = 0   # the moment lane (stream) the calling thread launches on: work buffers are per lane


def _work(kind, n, dtype, device):
    key = (kind, dtype, str(device), _lane)
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
                     added_lin: AddedLayerLin, p_range=None):
    """interaction! (lin) (interaction_lin.jl:337-351).  `p_range = (p_lo, p_hi)`: only these parameter slots (the caller knows
    the others to be zero in both operands, where the reference's loop over all Nparams computes exact zeros)."""
    n = _lib.lib().vsm_interaction_lin_work_elems(comp.N, comp.nSpec, comp_lin.P)
    work = _work("ia", int(n), comp.dtype, comp.R_mp.device)
    c, cl, a, al = comp.cstruct(), comp_lin.cstruct(), added.cstruct(), added_lin.cstruct()
    if p_range is not None and (p_range[0] > 0 or p_range[1] < comp_lin.P) and p_range[1] > p_range[0]:
        _lib.call("vsm_interaction_lin_range", comp.dtype, CR.IFACE[scattering_interface], comp.N, comp.nSpec, C.byref(c),
                  C.byref(cl), C.byref(a), C.byref(al), int(p_range[0]), int(p_range[1]), CR._ptr(work), CR._stream_ptr())
        return
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


REDUCE_M0 = True   # A/B switch (module attribute): False = the moment m = 0 with all Stokes components of the model


class SceneLin:
    """Everything rt_run(model, lin_model, NAer, NGas, NSurf) needs, resident in HBM (the linearized twin of CoreRT.Scene).
    The raw inputs -- tau_rayl, tau_abs, lin_model.tau_abs_dot [nSpec, Nz] and the small aerosol tables (tau_aer,
    tau_aer_dot, ssa / f_trunc and their Mie derivatives, Greek coefficients and their derivatives) -- are uploaded ONCE; the
    forward layer optics come from CoreRT.Scene's device pass (vsm_layer_optics, vsm_compute_Z_moments, vsm_layer_dtau) and
    their parameter derivatives from vsm_layer_optics_lin (tau_dot, varpi_dot, tau_sum_dot and the COEFFICIENTS of Z_dot over
    the component phase matrices: constructCoreOpticalProperties with lin_model, compEffectiveLayerProperties_lin.jl:43-197,
    types_lin.jl:196-380).  Z_dot[N, N, nSpec, P] is never materialised: elemental! (lin) forms it from the coefficients
    (vsm_elemental_lin_mix).  `run()` only launches kernels; exp(-dtau / mu0) is refreshed on the device per layer.

    `spec_slice` selects this rank's spectral shard; ndoubl and the tags always come from the FULL spectral axis
    (rt_kernel_lin.jl:87-95 uses batch-global maxima like the forward kernel).  `host_optics=True` builds the same inputs with
    the host mirror (host_model.constructCoreOpticalPropertiesLin, Z_dot materialised on the host) -- kept as a cross-check.

    Parameter slots (parameter_layout.jl:28-56): 7 per aerosol (tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p; their optics
    derivatives are inputs: LinModel.tau_aer_dot / lin_aerosol_optics), the gases, then ONE surface slot -- the Lambertian albedo
    (lambertian_surface_lin.jl:48-162) or the Cox-Munk wind speed (coxmunk_surface_lin.jl:27-102)."""

    def __init__(self, model: H.RTModel, lin_model: H.LinModel, NAer: int, NGas: int, NSurf: int,
                 spec_slice: Optional[slice] = None, host_optics: bool = False, _reduce_m0: bool = True):
        if NAer != lin_model.n_aer or NAer != len(model.aerosol_optics) or NSurf != 1 or NGas != len(lin_model.tau_abs_dot):
            raise _lib.VSMError("rt_run (linearized): NAer must equal the aerosols of model and lin_model, NGas = "
                                "len(lin_model.tau_abs_dot), NSurf = 1")
        if not isinstance(model.surface, (H.LambertianSurfaceScalar, H.CoxMunkSurface)):
            raise _lib.VSMError("rt_run (linearized): surface %r has no linearized builder here" % (model.surface,))
        arch, FT = model.architecture, model.float_type
        CR._require_gpu(arch)
        # vsm_elemental_lin_mix forms Z_dot from at most 16 component blocks (1 + 5 NAer: VSM_LIN_CT_MAX of vsm_lin.hip), i.e. three
        # aerosols; the reference has no such limit, so beyond it the scene takes the host-optics path (Z_dot materialised per
        # layer and moment, vsm_elemental_lin) from the start instead of failing inside run()
        if 1 + 5 * NAer > 16:
            host_optics = True
        self.model, self.lin_model, self.arch, self.FT = model, lin_model, arch, FT
        pol, qp = model.polarization_type, model.quad_points
        self.pol, self.qp = pol, qp
        self.layout = H.ParameterLayout(n_aerosols=NAer, n_gases=NGas, n_surface=NSurf)
        P, pl = self.layout.n_total, self.layout.n_layer_params
        self.P, self.pl, self.nAer, self.nGas = P, pl, NAer, NGas
        if P > 64:
            raise _lib.VSMError("rt_run (linearized): %d parameter slots (limit 64)" % P)
        # forward optics, quadrature, F0, composite / added / surface layers and R, T: the forward scene's device state
        # (rt_kernel_lin.jl:87 hard-codes scatter = true: a layer with tau*varpi <= 2 eps still goes through elemental! and
        # doubling! with ndoubl = 0 -- Scene.prepare gives such a layer ndoubl = 0 as well -- and only its interaction
        # follows the 00 / 01 / 10 tag of extractEffectiveProps)
        self.fwd = CR.Scene(model, spec_slice, host_optics=host_optics, full_added_layer=True)
        self.fwd.compute_hdrf = False
        fwd = self.fwd
        self.sl, self.S, self.N, self.dq, self.F0 = fwd.sl, fwd.S, fwd.N, fwd.dq, fwd.F0
        S, N, L = self.S, self.N, fwd.Nz
        dt, dev = CR._torch_dtype(FT), devi(arch)
        self.dt = dt
        nV = len(model.vza)
        self.C_, self.CT = 1 + NAer, 1 + NAer + 4 * NAer
        z = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
        # layouts: the reference's column-major [nSpec, p] per layer -> tensors (L, p, S)
        self.dtau_dot_all, self.varpi_dot, self.tau_sum_dot = z(L, P, S), z(L, max(pl, 1), S), z(L + 1, max(pl, 1), S)
        self.fz = z(L, S, self.C_) if NAer else None
        self.zdcoef = z(L, S, pl, self.CT) if NAer else None
        self.Zall = None          # per moment ((CT, N, N), (CT, N, N)): component blocks of Z and of the Greek derivatives
        self.host_zdot = None     # host_optics: per moment and layer the materialised Z_dot
        self.upload()
        if host_optics:
            self._prepare_host()
        else:
            self.prepare()
        self.surf = []
        for m in range(model.m_max + 1):
            rho = drho = None
            if isinstance(model.surface, H.CoxMunkSurface):
                rho, drho = CR.reflectance(model.surface, self.dq, m, arch, FT, deriv=True)
            self.surf.append((rho, drho))
        self.added, self.added_s, self.comp = fwd.added, fwd.added_surface, fwd.composite
        self.al, self.als = AddedLayerLin(FT, arch, P, N, S), AddedLayerLin(FT, arch, P, N, S, shared=True)
        self.cl = CompositeLayerLin(FT, arch, P, N, S)
        self.expk = torch.empty(max(S, 1), dtype=dt, device=dev)
        self.R, self.T = fwd.R_SFI, fwd.T_SFI
        self.Rd = torch.zeros((P, S, pol.n, nV), dtype=dt, device=dev)
        self.Td = torch.zeros_like(self.Rd)
        self._lanes = []
        self._fold = {}
        # The composite above the surface does not depend on a surface parameter: in the atmospheric layers the surface slots of
        # the added layer's AND of the composite's derivative stacks are zero, and the reference's loop over all Nparams
        # (interaction_lin.jl:242,291) computes exact zeros for them.  The layer interactions therefore run on the layer slots
        # [0, n_layer_params) only; the surface interaction on all of them.
        self._layer_slots = (0, pl) if 0 < pl < P else None
        # The Fourier moment m = 0 as a Stokes_IQ run (large batches; models without aerosol Jacobian slots over a Lambertian
        # surface).  At m = 0 no phase matrix couples (I,Q) with (U,V) (compute_Z_matrices.jl:26-110; the forward scene reads the
        # exact zeros off the device: `coupling`), the Lambertian surface reflects into I only, the gas and albedo derivatives
        # inherit the structure, and a beam without U / V components drives nothing in the (U,V) block: R, T, Rdot, Tdot of m = 0
        # are those of the same model carried with two Stokes components, U = V = 0.
        self.sub0 = None
        F0m = model.F0
        c0 = fwd.coupling[0] if getattr(fwd, "coupling", None) else None
        if (_reduce_m0 and REDUCE_M0 and pol.n >= 3 and S >= self.LANE_POINTS and NAer == 0 and not host_optics and c0 is not None
                and isinstance(model.surface, H.LambertianSurfaceScalar) and (F0m is None or not np.any(np.asarray(F0m)[2:] != 0))
                and not any(c0 >> (4 * a + b) & 1 or c0 >> (4 * b + a) & 1 for a in (0, 1) for b in range(2, pol.n))):
            self.sub0 = SceneLin(self._sub0_model(), lin_model, NAer, NGas, NSurf, spec_slice, _reduce_m0=False)

    def _sub0_model(self):
        """The Stokes_IQ model of the moment m = 0, derived from the parent's model AS IT IS NOW (a step rebinds model.tau_abs,
        scene.lin_model, ... and re-uploads: the sub-scene must see the new arrays, not the constructor-time snapshot)."""
        model, qp = self.model, self.model.quad_points
        F0m = model.F0
        qi = H.QuadPoints(qp.mu0, qp.imu0, qp.qp_mu, qp.wt_mu, np.repeat(qp.qp_mu, 2), np.repeat(qp.wt_mu, 2), qp.Nquad, qp.Nstreams)
        return dataclasses.replace(model, polarization_type=H.Stokes_IQ(), quad_points=qi, m_max=0,
                                   F0=None if F0m is None else np.ascontiguousarray(np.asarray(F0m)[:2]))

    # -- inputs -----------------------------------------------------------------------------------------------------------
    def upload(self):
        """H2D of what the linearization adds to Scene.upload(): lin_model.tau_abs_dot [nSpec, Nz, nGas] (FP64) and the
        aerosol derivative tables."""
        model, lin = self.model, self.lin_model
        conv = array_type(self.arch)
        S_full, L = model.tau_rayl.shape
        if getattr(self, "sub0", None) is not None:   # (a step re-uploads: the Stokes_IQ scene of m = 0 follows -- with the
            # parent's CURRENT model and lin_model: dataclasses.replace() is a shallow snapshot, so it is taken again here)
            self.sub0.model = self.sub0.fwd.model = self._sub0_model()
            self.sub0.lin_model = self.lin_model
            self.sub0.fwd.upload()
            self.sub0.upload()
        if self.nGas:
            g = np.stack([np.asarray(t, dtype=np.float64) for t in lin.tau_abs_dot])            # (nGas, S, L)
            if g.shape != (self.nGas, S_full, L):
                raise _lib.VSMError("lin_model.tau_abs_dot: %d arrays of [nSpec, Nz] = [%d, %d] expected" % (self.nGas, S_full, L))
            self.tau_abs_dot_d = conv(g).transpose(1, 2).contiguous()                           # (nGas, L, S)
        else:
            self.tau_abs_dot_d = None
        if self.nAer:
            tad = np.asarray(lin.tau_aer_dot, dtype=np.float64)                                 # [nAer, 7, L]
            if tad.shape != (self.nAer, 7, L):
                raise _lib.VSMError("lin_model.tau_aer_dot: [nAer, 7, Nz] expected")
            self.tau_aer_dot_d = conv(np.ascontiguousarray(tad.transpose(2, 0, 1)))             # (L, nAer, 7)
            self.ssa_dot_d = conv(np.ascontiguousarray(np.stack([np.asarray(l_.ssa_dot, dtype=np.float64)
                                                                  for l_ in lin.lin_aerosol_optics])))     # (nAer, 4)
            self.ftr_dot_d = conv(np.ascontiguousarray(np.stack([np.asarray(l_.f_trunc_dot, dtype=np.float64)
                                                                  for l_ in lin.lin_aerosol_optics])))
            self.greek_dot_dev = []
            for l_ in lin.lin_aerosol_optics:
                for g in l_.lin_greek_coefs:
                    tab = np.stack([np.asarray(getattr(g, k), dtype=np.float64) for k in
                                    ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")])
                    self.greek_dot_dev.append((conv(np.ascontiguousarray(tab)), tab.shape[1]))
        else:
            self.tau_aer_dot_d = self.ssa_dot_d = self.ftr_dot_d = None
            self.greek_dot_dev = []

    # -- device optics ----------------------------------------------------------------------------------------------------
    def prepare(self):
        model, fwd, dt = self.model, self.fwd, self.dt
        self._fold = {}   # (the folded per-layer inputs are copies of the optics)
        if getattr(self, "sub0", None) is not None:
            self.sub0.fwd.prepare()
            self.sub0.prepare()
        S_full, L = model.tau_rayl.shape
        _lib.call("vsm_layer_optics_lin", dt, S_full, fwd.lo, self.S, L, self.nAer, self.nGas, self.P, CR._ptr(fwd.tau_rayl_d),
                  CR._ptr(fwd.tau_abs_d), C.c_double(float(model.varpi_Cabannes)), CR._ptr(fwd.tau_aer_d), CR._ptr(fwd.ssa_d),
                  CR._ptr(fwd.ftr_d), CR._ptr(self.tau_abs_dot_d), CR._ptr(self.tau_aer_dot_d), CR._ptr(self.ssa_dot_d),
                  CR._ptr(self.ftr_dot_d), CR._ptr(fwd.nd_dev), CR._ptr(self.dtau_dot_all), CR._ptr(self.varpi_dot),
                  CR._ptr(self.tau_sum_dot), CR._ptr(self.fz), CR._ptr(self.zdcoef), CR._stream_ptr())
        if self.nAer:
            q = self.dq.cstruct()
            N = self.N
            self.Zall = []
            for m in range(model.m_max + 1):
                Zp = torch.empty((self.CT, N, N), dtype=dt, device=devi(self.arch))
                Zm = torch.empty_like(Zp)
                for k, (gd, lmax) in enumerate(fwd.greek_dev + self.greek_dot_dev):
                    _lib.call("vsm_compute_Z_moments", dt, C.byref(q), m, lmax, CR._ptr(gd), CR._ptr(Zp[k]), CR._ptr(Zm[k]),
                              CR._stream_ptr())
                self.Zall.append((Zp, Zm))

    def _prepare_host(self):
        """The same derivative inputs from the host mirror (numpy, Z_dot materialised per layer and moment)."""
        model, lin, FT, fwd = self.model, self.lin_model, self.FT, self.fwd
        S_full, L = model.tau_rayl.shape
        sl, pl = self.sl, self.pl
        conv = array_type(self.arch)
        self.host_zdot = []
        for m in range(model.m_max + 1):
            lods = H.constructCoreOpticalProperties(model, m)
            lins = H.constructCoreOpticalPropertiesLin(model, lin, lods, m)
            zd = []
            tsd = np.zeros((S_full, pl))
            for iz in range(L):
                if m == 0:
                    nd = fwd.moments[0]["layers"][iz]["nd"]
                    dall = np.zeros((S_full, self.P))
                    dall[:, :pl] = lins[iz].tau_dot / FT(2 ** nd)
                    self.dtau_dot_all[iz].copy_(to_device_sp(dall[sl], self.arch, FT))
                    if pl:
                        self.varpi_dot[iz].copy_(to_device_sp(lins[iz].varpi_dot[sl], self.arch, FT))
                        self.tau_sum_dot[iz].copy_(to_device_sp(tsd[sl], self.arch, FT))
                    tsd = tsd + lins[iz].tau_dot
                Zpd, Zmd = lins[iz].Zpp_dot, lins[iz].Zmp_dot
                if Zpd is not None and Zpd.ndim == 4:      # per-point Z_dot: this rank's spectral block
                    Zpd, Zmd = Zpd[:, sl], Zmd[:, sl]
                zpd, zs_, zp_ = to_device_zdot(Zpd, self.arch, FT)
                zmd, _, _ = to_device_zdot(Zmd, self.arch, FT)
                zd.append((zpd, zmd, (zs_, zp_)))
            if m == 0 and pl:
                self.tau_sum_dot[L].copy_(to_device_sp(tsd[sl], self.arch, FT))
            self.host_zdot.append(zd)

    # Small spectral batches are latency bound: a linearized layer step of ONE workgroup is ~0.4 ms and the 33 layers of a moment
    # are a dependent chain (C3: 2 points, 22 moments x 33 layers = 0.3 s moment by moment).  The Fourier moments are independent
    # until post-processing (rt_run_lin.jl:200-322), so below LANE_POINTS spectral points they run on up to LANES HIP streams, each
    # lane with its own added / composite layers (+ derivatives), work buffers and R / T / Rdot / Tdot accumulators, summed at the
    # end.  (The sum over moments is then taken lane by lane: equal to the sequential walk up to the rounding of the reordering.)
    LANES = 8
    LANE_POINTS = 64
    # A folded group of at most PARALLEL_LAYER_POINTS (moment, point) pairs doubles all its layers side by side on the lane streams
    # (each layer into an added layer of its own: 12 (1 + P) N^2 elements per pair and layer) before the interactions walk the
    # column (_run_folded)
    PARALLEL_LAYERS = True
    PARALLEL_LAYER_POINTS = 128
    MERGE_M0 = True          # the folded walk takes m = 0 and the moments m > 0 as one batch (False: two batches, two chains)
    TREE_INTERACTIONS = True  # the interactions of a pre-doubled column whose tags are all 11 as a tree over the lane streams

    def _lane_state(self, k):
        """Workspace of moment lane k (lane 0: the scene's own buffers)."""
        while len(self._lanes) <= k:
            if not self._lanes:
                self._lanes.append(dict(added=self.added, al=self.al, comp=self.comp, cl=self.cl, added_s=self.added_s, als=self.als,
                                        expk=self.expk, R=self.R, T=self.T, Rd=self.Rd, Td=self.Td, stream=None))
                continue
            FT, arch, P, N, S = self.FT, self.arch, self.P, self.N, self.S
            self._lanes.append(dict(
                added=CR.make_added_layer(FT, arch, (N, N), S), al=AddedLayerLin(FT, arch, P, N, S),
                comp=CR.make_composite_layer(FT, arch, (N, N), S), cl=CompositeLayerLin(FT, arch, P, N, S),
                added_s=self._new_surface_added(), als=AddedLayerLin(FT, arch, P, N, S, shared=True),
                expk=torch.empty_like(self.expk), R=torch.zeros_like(self.R), T=torch.zeros_like(self.T),
                Rd=torch.zeros_like(self.Rd), Td=torch.zeros_like(self.Td), stream=torch.cuda.Stream()))
        return self._lanes[k]

    def _new_surface_added(self):
        a0 = self.added_s
        return CR.AddedLayer(self.FT, self.arch, self.N, self.S, a0.shared, a0.d_symmetric)

    def run(self, lanes: Optional[int] = None, fold: Optional[bool] = None, graph: Optional[bool] = None):
        """The device-resident part of rt_run_lin.jl:200-322: Fourier loop -> layers -> surface -> post-processing.
        `lanes`: number of concurrent moment lanes (default: LANES for batches below LANE_POINTS points, else 1);
        `fold`: walk the layers with the Fourier moments folded into the spectral axis (_run_folded; default: for such small
        batches when the scene has no aerosol Jacobian slots);
        `graph`: replay the pass from a HIP graph (default: `self.graph_replay`, which a caller that steps the SAME scene many
        times -- a retrieval loop, the bench -- switches on; the first such call costs a warm-up pass and the capture).  A small
        batch is bound by the HOST: a pass of the C3 scene is ~ 2500 launches and library calls of ~ 15 us each, issued by one
        thread, so its two layer chains never overlap on the device (`tools/trace_timeline.py`: chain after chain, 14 ms each, with
        5 ms of launch-bound input folding in front of either).  The launch sequence depends only on the layer structure (ndoubl,
        interface tags, lanes / fold) -- the graph is captured once per such key, on buffers the scene owns, and reads the
        optics where `prepare()` puts them."""
        if graph is None:
            graph = bool(getattr(self, "graph_replay", False)) and self.S > 0 and not torch.cuda.is_current_stream_capturing()
        if graph:
            return self._run_graph(lanes, fold)
        return self._run_eager(lanes, fold)

    def _run_graph(self, lanes, fold):
        key = (lanes, fold, tuple((ly["nd"], ly["iface"]) for ly in self.fwd.moments[0]["layers"]), len(self.fwd.moments),
               self.fwd.moments[0]["iface_surface"])
        if getattr(self, "_graph_key", None) != key:
            self._graph = None
            # The library's scratch is keyed by (device, stream) and may not grow under a capture: the warm-up pass runs on the
            # very stream the capture uses (and on the same lane streams), so every buffer the captured launches need exists.
            if getattr(self, "_graph_stream", None) is None:
                self._graph_stream = torch.cuda.Stream()
            gs = self._graph_stream
            gs.wait_stream(torch.cuda.current_stream())
            self._fold = {}                       # (the folded per-layer inputs are copies of the optics: they must be formed
            with torch.cuda.stream(gs):           #  INSIDE the graph, not come out of the cache of an earlier pass)
                self._run_eager(lanes, fold)
            torch.cuda.synchronize()
            self._fold = {}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=gs):
                self._run_eager(lanes, fold)
            self._graph, self._graph_key = g, key
        self._graph.replay()
        return self.R, self.T, self.Rd, self.Td

    def _run_eager(self, lanes: Optional[int] = None, fold: Optional[bool] = None):
        global _lane
        S = self.S
        nm = len(self.fwd.moments)
        small = 0 < S < self.LANE_POINTS
        if lanes is None:
            lanes = self.LANES if small else 1
        lanes = max(1, min(lanes, nm))
        can_fold = self.Zall is None and self.host_zdot is None and nm > 2 and S > 0
        if fold is None:
            fold = small and can_fold and lanes > 1
        if fold and not can_fold:
            raise _lib.VSMError("SceneLin.run(fold=True): needs a scene without aerosol Jacobian slots and more than two moments")
        st = [self._lane_state(k) for k in range(lanes)]
        torch._foreach_zero_([t for w in st for t in (w["R"], w["T"], w["Rd"], w["Td"])])
        if S == 0:
            return self.R, self.T, self.Rd, self.Td
        if fold:
            return self._run_folded(st)
        if lanes == 1:
            reduced = self.sub0 is not None
            if reduced:   # m = 0 first, like the Fourier loop: its I, Q rows are copied in, the later moments accumulate on top
                r0, t0, rd0, td0 = self.sub0.run(lanes=1, fold=False)
                self.R[:, :2, :].copy_(r0)
                self.T[:, :2, :].copy_(t0)
                self.Rd[:, :, :2, :].copy_(rd0)
                self.Td[:, :, :2, :].copy_(td0)
            for mom in self.fwd.moments:
                if reduced and mom["m"] == 0:
                    continue
                self._run_moment(mom, st[0])
            return self.R, self.T, self.Rd, self.Td
        main = torch.cuda.current_stream()
        for w in st[1:]:
            w["stream"].wait_stream(main)       # (the zeroing above and whatever produced the inputs)
        try:
            for i, mom in enumerate(self.fwd.moments):
                k = i % lanes
                _lane = k
                if k == 0:
                    self._run_moment(mom, st[0])
                else:
                    with torch.cuda.stream(st[k]["stream"]):
                        self._run_moment(mom, st[k])
        finally:
            _lane = 0
        for w in st[1:]:
            main.wait_stream(w["stream"])
        for w in st[1:]:
            self.R += w["R"]
            self.T += w["T"]
            self.Rd += w["Rd"]
            self.Td += w["Td"]
        return self.R, self.T, self.Rd, self.Td

    # ---- moments folded into the spectral axis -------------------------------------------------------------------------------
    # The layer kernels are batched over the spectral axis with per-point dtau, varpi, Z (stride N^2) and derivatives; the Fourier
    # moment enters elemental! (lin) only through m == 0 (the weight factor 1/2 vs 1/4, elemental_lin.jl:77-206).  For a small
    # batch the moments m >= 1 therefore walk the layers as ONE batch of (moment, point) pairs -- and m = 0 as another --: two
    # dependent chains of layer steps instead of one per moment (C3: 22).  Each moment's block of the folded composite is then
    # copied to a lane and finished there (surface layer, its interaction, post-processing: these depend on m).
    def _fold_group(self, gi, group):
        """Workspace and per-layer inputs of a folded group (cached until the optics change)."""
        key = (gi, len(group))
        g = self._fold.get(key)
        if g is not None:
            return g
        FT, arch, P, N, S = self.FT, self.arch, self.P, self.N, self.S
        n = len(group)
        Sf = n * S
        g = dict(added=CR.make_added_layer(FT, arch, (N, N), Sf), al=AddedLayerLin(FT, arch, P, N, Sf),
                 comp=CR.make_composite_layer(FT, arch, (N, N), Sf), cl=CompositeLayerLin(FT, arch, P, N, Sf),
                 expk=torch.empty(Sf, dtype=self.dt, device=self.expk.device), F0=self.F0.repeat(n, 1).contiguous(), layers=[])
        # per-layer inputs of the folded walk: the optics of a layer do not depend on the moment (repeated along the folded axis, all
        # layers in one operation each), its phase matrices do (vsm_mix_Z_moments: every moment's Z of a layer in one launch)
        fwd = self.fwd
        lo, hi, L = fwd.lo, fwd.hi, fwd.Nz
        tau_f, varpi_f = fwd.tau[:L, lo:hi].repeat(1, n), fwd.varpi[:L, lo:hi].repeat(1, n)
        dtau_f, tsum_f = fwd.dtau[:L, lo:hi].repeat(1, n), fwd.tau_sum[:L, lo:hi].repeat(1, n)
        dtd_f, vd_f = self.dtau_dot_all.repeat(1, 1, n), self.varpi_dot.repeat(1, 1, n)
        tsd_f = self.tau_sum_dot.repeat(1, 1, n)
        Zf = torch.empty((2, L, Sf, N, N), dtype=self.dt, device=self.expk.device)
        for iz in range(L):
            props_m = [mom["layers"][iz]["props"] for mom in group]
            p0 = props_m[0]
            mixed = p0.fcomp is not None
            single = 0
            if not mixed:   # (one scatterer: the props hold the one-block views Zc[m][k : k + 1])
                single = int(fwd.zcomp[iz][1])
            zp = (C.c_void_p * n)(*[fwd.Zc[mom["m"]][0].data_ptr() for mom in group])
            zm = (C.c_void_p * n)(*[fwd.Zc[mom["m"]][1].data_ptr() for mom in group])
            _lib.call("vsm_mix_Z_moments", self.dt, N, S, int(p0.fcomp.shape[1]) if mixed else 0, n, zp, zm, single,
                      CR._ptr(p0.fcomp) if mixed else None, CR._ptr(Zf[0, iz]), CR._ptr(Zf[1, iz]), CR._stream_ptr())
            props = CR.DeviceLayerOptics(tau_f[iz], varpi_f[iz], Zf[0, iz], Zf[1, iz], p0.max_tau_varpi, p0.tau_h, p0.varpi_h)
            g["layers"].append(dict(props=props, dtau=dtau_f[iz], tau_sum=tsum_f[iz], dtd=dtd_f[iz], vd=vd_f[iz], tsd=tsd_f[iz]))
        if self.PARALLEL_LAYERS and 0 < Sf <= self.PARALLEL_LAYER_POINTS:
            g["per_layer"] = [dict(added=CR.make_added_layer(FT, arch, (N, N), Sf), al=AddedLayerLin(FT, arch, P, N, Sf),
                                   expk=torch.empty(Sf, dtype=self.dt, device=self.expk.device)) for _ in range(self.fwd.Nz)]
        self._fold[key] = g
        return g

    def _run_folded(self, st):
        global _lane
        pol, qp, dt, S, N, P, pl = self.pol, self.qp, self.dt, self.S, self.N, self.P, self.pl
        moms = self.fwd.moments
        groups = [[m_ for m_ in moms if m_["m"] == 0], [m_ for m_ in moms if m_["m"] > 0]]
        if self.MERGE_M0 and groups[0] and groups[1] and len(moms) <= 24:
            # ONE folded batch: the pairs of m = 0 in front (vsm_elemental_lin_fold tells them apart; doubling! and interaction!
            # do not know the moment) -- half the launches of the layer walk, one chain of interactions instead of two
            groups = [[], groups[0] + groups[1]]
        mu0 = C.c_double(qp.mu0) if dt == torch.float64 else C.c_float(qp.mu0)
        main = torch.cuda.current_stream()
        lanes = len(st)
        for w in st[1:]:
            w["stream"].wait_stream(main)       # (the zeroing of the accumulators and whatever produced the inputs)

        def double_layer(gi, group, iz, lane, added, al, expk):
            """elemental! (lin) + doubling! (lin) of layer iz of a folded group on the current stream (work buffers of `lane`)."""
            global _lane
            _lane = lane
            g = self._fold_group(gi, group)
            fl, ly0 = g["layers"][iz], group[0]["layers"][iz]
            n_m0 = S * sum(1 for m_ in group if m_["m"] == 0) if group[-1]["m"] > 0 else 0
            elemental_lin_(pol, fl["tau_sum"], fl["tsd"], fl["dtau"], fl["dtd"], g["F0"], fl["props"], fl["vd"], None, None, (0, 0),
                           pl, group[-1]["m"], ly0["nd"], self.dq, added, al, n_m0=n_m0)
            _lib.call("vsm_layer_expk", dt, len(group) * S, CR._ptr(fl["dtau"]), mu0, CR._ptr(expk), CR._stream_ptr())
            doubling_allparams_(pol, expk, ly0["nd"], added, al, fl["dtd"], qp.mu0, pl)

        # elemental! + doubling! of a layer depend on nothing but its own optics (rt_kernel_lin.jl:87-160: the added layer is built
        # before interaction! is called).  With so few (moment, point) pairs that one layer's launch fills a sixth of the chip, ALL
        # layers of both groups are doubled side by side on the lane streams, each into an added layer of its own, and only the
        # interactions -- the one sequential part of the adding method -- walk the column in order.  (Plain fork / join of the lane
        # streams from the main stream: what a HIP graph capture takes; layer streams forked from a lane stream crashed
        # hipStreamEndCapture.)
        gstate = [self._fold_group(gi, grp) if grp else None for gi, grp in enumerate(groups)]
        pre = lanes > 1 and all(g_ is None or g_.get("per_layer") is not None for g_ in gstate)
        if pre:
            cnt = 0
            for gi, grp in enumerate(groups):
                if not grp:
                    continue
                for iz in range(self.fwd.Nz):
                    k = 1 + cnt % (lanes - 1)
                    cnt += 1
                    b = gstate[gi]["per_layer"][iz]
                    with torch.cuda.stream(st[k]["stream"]):
                        double_layer(gi, grp, iz, k, b["added"], b["al"], b["expk"])
            _lane = 0
            for w in st[1:]:
                main.wait_stream(w["stream"])
            for w in st[1:]:
                w["stream"].wait_stream(main)

        def tree(gi, group):
            """The interactions of a pre-doubled column as a tree (every tag 11): pairs of neighbouring sub-columns are combined
            level by level -- composite (upper) (+) composite-as-added-layer (lower) -- on the lane streams, ceil(log2 Nz) dependent
            interactions instead of Nz - 1.  Same equations, another association: results agree with the chain to rounding."""
            global _lane
            g = gstate[gi]
            Sf = len(group) * S
            FT, arch = self.FT, self.arch
            pool = g.setdefault("tree_pool", [])
            used = 0
            nodes = [("leaf", iz) for iz in range(self.fwd.Nz)]
            while len(nodes) > 1:
                nxt = []
                for w in st[1:]:
                    w["stream"].wait_stream(main)
                for ip in range(len(nodes) // 2):
                    a, b = nodes[2 * ip], nodes[2 * ip + 1]
                    k = ip % lanes
                    _lane = k
                    ctx = torch.cuda.stream(st[k]["stream"]) if st[k]["stream"] is not None else contextlib.nullcontext()
                    with ctx:
                        if a[0] == "leaf":
                            if used == len(pool):
                                pool.append((CR.make_composite_layer(FT, arch, (N, N), Sf), CompositeLayerLin(FT, arch, P, N, Sf)))
                            comp, cl = pool[used]
                            used += 1
                            la = g["per_layer"][a[1]]
                            CR.copy_added_to_composite_(comp, la["added"])
                            a_, c_ = la["al"].cstruct(), cl.cstruct()
                            _lib.call("vsm_copy_added_to_composite_lin", dt, N, Sf, C.byref(a_), C.byref(c_), CR._stream_ptr())
                        else:
                            comp, cl = a[1], a[2]
                        if b[0] == "leaf":
                            added, al = g["per_layer"][b[1]]["added"], g["per_layer"][b[1]]["al"]
                        else:
                            added, al = _CompositeAsAdded(b[1]), _CompositeLinAsAdded(b[2])
                        interaction_lin_("11", comp, cl, added, al, p_range=self._layer_slots)
                    nxt.append(("comp", comp, cl))
                if len(nodes) % 2:
                    nxt.append(nodes[-1])
                _lane = 0
                for w in st[1:]:
                    main.wait_stream(w["stream"])
                nodes = nxt
            return nodes[0][1], nodes[0][2]

        def chain(gi, group, lane):
            """The layer walk of a folded group on the current stream (work buffers of `lane`)."""
            global _lane
            _lane = lane
            g = gstate[gi]
            Sf = len(group) * S
            comp, cl = g["comp"], g["cl"]
            if (pre and self.TREE_INTERACTIONS and lane == 0 and self.fwd.Nz >= 4
                    and all(group[0]["layers"][iz]["iface"] == "11" for iz in range(1, self.fwd.Nz))):
                return tree(gi, group)
            for iz in range(self.fwd.Nz):
                ly0 = group[0]["layers"][iz]
                if pre:
                    added, al = g["per_layer"][iz]["added"], g["per_layer"][iz]["al"]
                else:
                    added, al = g["added"], g["al"]
                    double_layer(gi, group, iz, lane, added, al, g["expk"])
                if iz == 0:
                    CR.copy_added_to_composite_(comp, added)
                    a_, c_ = al.cstruct(), cl.cstruct()
                    _lib.call("vsm_copy_added_to_composite_lin", dt, N, Sf, C.byref(a_), C.byref(c_), CR._stream_ptr())
                else:
                    interaction_lin_(ly0["iface"], comp, cl, added, al, p_range=self._layer_slots)
            return comp, cl

        def finish(mom, w, lane, comp, cl, im):
            """Moment `im` of a folded composite -> lane workspace `w`, then surface / interaction / post-processing."""
            global _lane
            _lane = lane
            sl = slice(im * S, (im + 1) * S)
            fields = ("R_mp", "R_pm", "T_pp", "T_mm", "J0_p", "J0_m")
            torch._foreach_copy_([getattr(w["comp"], f) for f in fields] + [getattr(w["cl"], f) for f in fields],
                                 [getattr(comp, f)[sl] for f in fields] + [getattr(cl, f)[:, sl] for f in fields])   # (one multi-tensor launch)
            self._finish_moment(mom, w)

        try:
            # the m = 0 chain (and its finish) on lane 1's stream, concurrent with the m >= 1 chain on the main stream
            if groups[0]:
                if lanes > 1:
                    with torch.cuda.stream(st[1]["stream"]):
                        comp0, cl0 = chain(0, groups[0], 1)
                        for im, mom in enumerate(groups[0]):
                            finish(mom, st[1], 1, comp0, cl0, im)
                else:
                    comp0, cl0 = chain(0, groups[0], 0)
                    for im, mom in enumerate(groups[0]):
                        finish(mom, st[0], 0, comp0, cl0, im)
            if groups[1]:
                comp1, cl1 = chain(1, groups[1], 0)
                for w in st[1:]:
                    w["stream"].wait_stream(main)
                for im, mom in enumerate(groups[1]):
                    kk = im % lanes
                    if st[kk]["stream"] is None:
                        finish(mom, st[kk], kk, comp1, cl1, im)
                    else:
                        with torch.cuda.stream(st[kk]["stream"]):
                            finish(mom, st[kk], kk, comp1, cl1, im)
        finally:
            _lane = 0
        for w in st[1:]:
            main.wait_stream(w["stream"])
        for w in st[1:]:
            torch._foreach_add_([self.R, self.T, self.Rd, self.Td], [w["R"], w["T"], w["Rd"], w["Td"]])
        return self.R, self.T, self.Rd, self.Td

    def _run_moment(self, mom, w):
        """One Fourier moment on the workspace `w` (launches on the current stream)."""
        model, pol, qp, FT, dt, fwd = self.model, self.pol, self.qp, self.FT, self.dt, self.fwd
        N, S, P, pl = self.N, self.S, self.P, self.pl
        added, al, comp, cl = w["added"], w["al"], w["comp"], w["cl"]
        mu0 = C.c_double(qp.mu0) if dt == torch.float64 else C.c_float(qp.mu0)
        q_ = self.dq.cstruct()
        m = mom["m"]
        for iz, ly in enumerate(mom["layers"]):
            props = ly["props"]
            dtd, vd, tsd = self.dtau_dot_all[iz], self.varpi_dot[iz], self.tau_sum_dot[iz]
            if self.Zall is not None:
                mixed, k = fwd.zcomp[iz]
                a_, al_ = added.cstruct(), al.cstruct()
                _lib.call("vsm_elemental_lin_mix", dt, C.byref(q_), S, m, ly["nd"], CR._ptr(ly["dtau"]), CR._ptr(props.varpi),
                          CR._ptr(ly["tau_sum"]), CR._ptr(self.F0), self.C_, self.CT, CR._ptr(self.Zall[m][0]),
                          CR._ptr(self.Zall[m][1]), -1 if mixed else k, CR._ptr(self.fz[iz]), pl, CR._ptr(dtd), CR._ptr(vd),
                          CR._ptr(tsd), CR._ptr(self.zdcoef[iz]), C.byref(a_), C.byref(al_), CR._stream_ptr())
            else:
                zpd, zmd, zds = self.host_zdot[m][iz] if self.host_zdot is not None else (None, None, (0, 0))
                elemental_lin_(pol, ly["tau_sum"], tsd, ly["dtau"], dtd, self.F0, props.materialize(), vd, zpd, zmd, zds, pl,
                               m, ly["nd"], self.dq, added, al)
            _lib.call("vsm_layer_expk", dt, S, CR._ptr(ly["dtau"]), mu0, CR._ptr(w["expk"]), CR._stream_ptr())
            doubling_allparams_(pol, w["expk"], ly["nd"], added, al, dtd, qp.mu0, pl)
            if iz == 0:
                CR.copy_added_to_composite_(comp, added)
                a_, c_ = al.cstruct(), cl.cstruct()
                _lib.call("vsm_copy_added_to_composite_lin", dt, N, S, C.byref(a_), C.byref(c_), CR._stream_ptr())
            else:
                interaction_lin_(ly["iface"], comp, cl, added, al, p_range=self._layer_slots)
        self._finish_moment(mom, w)

    def _finish_moment(self, mom, w):
        """Surface layer, its interaction and the post-processing of one Fourier moment on the workspace `w`."""
        model, pol, qp, FT, dt, fwd = self.model, self.pol, self.qp, self.FT, self.dt, self.fwd
        N, S, P, pl = self.N, self.S, self.P, self.pl
        comp, cl = w["comp"], w["cl"]
        isurf = self.layout.surface_index(0)
        q_ = self.dq.cstruct()
        m = mom["m"]
        weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
        a_, al_ = w["added_s"].cstruct(), w["als"].cstruct()
        tau_sum_s, tsd_s = mom["tau_sum_surface"], self.tau_sum_dot[fwd.Nz]
        rho, drho = self.surf[m]
        if isinstance(model.surface, H.CoxMunkSurface):
            _lib.call("vsm_brdf_surface_lin", dt, C.byref(q_), S, m, CR._ptr(rho), CR._ptr(drho), isurf, CR._ptr(tau_sum_s),
                      CR._ptr(tsd_s), pl, CR._ptr(self.F0), C.byref(a_), C.byref(al_), CR._stream_ptr())
        else:
            alb = C.c_double(model.surface.albedo) if dt == torch.float64 else C.c_float(model.surface.albedo)
            _lib.call("vsm_lambertian_surface_lin", dt, C.byref(q_), S, m, alb, isurf, CR._ptr(tau_sum_s), CR._ptr(tsd_s), pl,
                      CR._ptr(self.F0), C.byref(a_), C.byref(al_), CR._stream_ptr())
        interaction_lin_(mom["iface_surface"], comp, cl, w["added_s"], w["als"])
        CR.postprocessing_vza_(pol, comp, model.vza, model.vaz, qp, m, float(weight), w["R"], w["T"])
        row0, wts = _pp_args(pol, qp, model.vza, model.vaz, m, weight, dt)
        _lib.call("vsm_postprocess_vza_lin", dt, N, pol.n, S, len(model.vza), P, row0, wts, CR._ptr(cl.J0_m), CR._ptr(cl.J0_p),
                  CR._ptr(w["Rd"]), CR._ptr(w["Td"]), CR._stream_ptr())

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
        for mom in self.fwd.moments:
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
    _lib.check_device_status("rt_run (linearized)")
    return scene.results_host()
