"""CoreRT host layer for the MI355X backend: the reference's operator / CoreKernel
interface for the rt_run hot path, implemented as calls into libvsmartmom_hip.so.

Mirrors (names follow the reference; Julia's trailing `!` becomes `_`):
  batched_mul, batch_inv_                       ext/gpu_batched_cuda.jl:97-233, tools/cpu_batched.jl
  make_added_layer, make_composite_layer        tools/rt_helper_functions.jl:130-151,259-270
  elemental_, doubling_, interaction_           CoreKernel/elemental.jl, doubling.jl, interaction.jl
  rt_kernel_                                    CoreKernel/rt_kernel.jl:175-250
  create_surface_layer_, postprocessing_vza_    Surfaces/lambertian_surface.jl:41-95, rpv_surface.jl:51-97, tools/postprocessing_vza.jl
  reflectance, apply_ss_correction_             Surfaces/coxmunk_surface.jl:381-460, 481-569
  rt_run                                        rt_run.jl:238-539 (noRS, SFI, Lambertian)

Device arrays are torch tensors (plumbing only: allocation, streams, H2D/D2H).
Memory layout is the reference's column-major [N,N,S]: a matrix batch is a
contiguous tensor of shape (S, N, N) whose slice [s] holds the matrix
*transposed* (element (i,j) at [s, j, i]); vectors [N,1,S] are (S, N).
`to_device_matrix` / `from_device_matrix` convert from/to math order [S,i,j].
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import host_model as H
from .architectures import GPU, CPU, architecture, array_type, devi, synchronize_if_gpu, to_host

IFACE = {"00": 0, "01": 1, "10": 2, "11": 3}
MOMENT_BATCH = 4      # Fourier moments per layer-step call of Scene.run for large batches (each moment needs a CompositeLayer)
MOMENT_BATCH_MAX = 24  # ... for small spectral batches: every moment of the run in one launch (VSM_MM_MAX of the library)
# A/B switches of the host path (module attributes, set by tests and tools; never read from the environment):
MOMENT_BATCHING = True  # False: Scene.run walks the Fourier moments one by one
THERMAL_FUSION = True   # False: the `:thermal` slot at operator level (tools/thermal_timing.py)
NATIVE_RUN = True       # False: the layer loop on the reference-layout CompositeLayer (vsm_layer_forward_multi) instead of vsm_run_*
NATIVE_DROPIN = True    # rt_kernel_ keeps the CompositeLayer it is handed in the kernels' native layout between its calls (a per-
#                         composite registry, exported lazily: the path of the reference's own driver order, rt_run.jl:383-453)
REFERENCE_ORDER = False  # True: Scene.run walks `for m` outside `for iz`, one rt_kernel_ call per (m, iz) on ONE CompositeLayer --
#                         exactly the call sequence of the unpatched rt_run.jl (what julia/vSmartMOMROCmExt.jl is reached through)


def _require_gpu(arch):
    if not isinstance(arch, GPU):
        raise _lib.VSMError("vsmartmom.jl_amd provides the MI355X path only; architecture %r has no compute path here "
                            "(no CPU fallback by design)" % (arch,))
    if not torch.cuda.is_available():
        raise _lib.VSMError("no MI355X visible to HIP (torch.cuda.is_available() is False)")
    # One process drives ONE GPU: kernels are launched on the current device's stream and the library keeps its scratch
    # and kernel attributes per process, so data on any other device would be touched from the wrong GPU.
    if arch.device_index != torch.cuda.current_device():
        raise _lib.VSMError("architecture GPU(%d) is not the current HIP device (%d): one process per GPU -- call "
                            "torch.cuda.set_device(%d) first" % (arch.device_index, torch.cuda.current_device(),
                                                                 arch.device_index))


def _torch_dtype(FT):
    return torch.float64 if np.dtype(FT) == np.float64 else torch.float32


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def to_device_matrix(A: np.ndarray, arch, FT) -> torch.Tensor:
    """math order [S,i,j] (or [i,j]) -> device tensor in the reference's column-major layout."""
    A = np.asarray(A, dtype=FT)
    if A.ndim == 2:
        A = A[None]
    return array_type(arch)(np.ascontiguousarray(A.transpose(0, 2, 1)))


def from_device_matrix(t: torch.Tensor) -> np.ndarray:
    return to_host(t).transpose(0, 2, 1).copy()


# ----------------------------------------------------------------------------
# L1 operator API
# ----------------------------------------------------------------------------
def batched_mul(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """`A ⊠ B` (NNlib.batched_mul; ext/gpu_batched_cuda.jl:208-233): allocating, inputs untouched.
    A: (S, K, M) layout-tensor of an [M,K,S] batch, B: (S, Nc, K) or (S, K) for a vector batch."""
    _require_gpu(architecture(A))
    vec = B.dim() == 2
    S, K, M = A.shape
    Nc = 1 if vec else B.shape[1]
    kb = B.shape[1] if vec else B.shape[2]
    if kb != K or B.shape[0] not in (S, 1):
        raise _lib.VSMError("batched_mul: dimension mismatch %s x %s" % (tuple(A.shape), tuple(B.shape)))
    A, B = A.contiguous(), B.contiguous()
    Cout = torch.empty((S, M) if vec else (S, Nc, M), dtype=A.dtype, device=A.device)
    sb = 0 if B.shape[0] == 1 and S > 1 else K * Nc
    _lib.call("vsm_batched_mul", A.dtype, M, Nc, K, S, _ptr(A), M * K, _ptr(B), sb, _ptr(Cout), _stream_ptr())
    return Cout


def batch_inv_(X: torch.Tensor, A: torch.Tensor, info: Optional[torch.Tensor] = None):
    """`batch_inv!(X, A)` (ext/gpu_batched_cuda.jl:97-182).  X may alias A."""
    _require_gpu(architecture(A))
    S, N, N2 = A.shape
    if N != N2 or X.shape != A.shape:
        raise _lib.VSMError("batch_inv!: square batches of equal shape required")
    _lib.call("vsm_batch_inv", A.dtype, N, S, _ptr(A), _ptr(X), _ptr(info), _stream_ptr())
    return X


def batch_solve_(X: torch.Tensor, A: torch.Tensor, B: torch.Tensor, info: Optional[torch.Tensor] = None):
    """`batch_solve!(X, A, B)` (ext/gpu_batched_cuda.jl:72-94): X[:,:,s] = A[:,:,s] \\ B[:,:,s].  A: (S, N, N) layout tensor,
    B / X: (S, Nrhs, N) (or (S, N) for one right-hand side per slice)."""
    _require_gpu(architecture(A))
    S, N, N2 = A.shape
    nrhs = 1 if B.dim() == 2 else B.shape[1]
    if N != N2 or B.shape[0] != S or B.shape[-1] != N or X.shape != B.shape:
        raise _lib.VSMError("batch_solve!: A [N,N,S] and B, X [N,Nrhs,S] required")
    work = _lib.poison(torch.empty_like(A))
    _lib.call("vsm_batch_solve", A.dtype, N, nrhs, S, _ptr(A.contiguous()), _ptr(B.contiguous()), _ptr(X), _ptr(work), _ptr(info),
              _stream_ptr())
    return X


def batched_pointer_cache(A):
    """ext/gpu_batched_cuda.jl:65-69 -- no pointer arrays are needed by the HIP kernels."""
    return None


# ----------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------
@dataclass
class DeviceQuad:
    """Device copy of QuadPoints + the C struct handed to the kernels."""
    host: H.QuadPoints
    mu: torch.Tensor
    wt: torch.Tensor
    n_stokes: int
    dtype: torch.dtype

    def cstruct(self):
        cls = _lib.vsm_quad_f64 if self.dtype == torch.float64 else _lib.vsm_quad_f32
        return cls(self.mu.data_ptr(), self.wt.data_ptr(), int(self.mu.numel()), self.n_stokes, self.host.imu0,
                   self.host.mu0)


def device_quad(qp: H.QuadPoints, pol: H.PolarizationType, arch, FT) -> DeviceQuad:
    conv = array_type(arch)
    return DeviceQuad(qp, conv(qp.qp_muN.astype(FT)), conv(qp.wt_muN.astype(FT)), pol.n, _torch_dtype(FT))


class AddedLayer:
    """src/CoreRT/types.jl:155-230 AddedLayer (elastic fields).  `shared=True` allocates ONE
    N x N block per matrix, broadcast over the spectral axis (surface layers)."""

    def __init__(self, FT, arch, N, nSpec, shared=False, d_symmetric=0):
        dev, dt = devi(arch), _torch_dtype(FT)
        sm = 1 if shared else nSpec
        z = lambda: torch.zeros((sm, N, N), dtype=dt, device=dev)
        self.r_mp, self.t_pp = z(), z()
        # d_symmetric = nStokes: r+-/t-- are never materialised (derived as D r-+ D, D t++ D by the fused
        # kernels) -- half the HBM traffic of a layer.  0 = the reference's full AddedLayer.
        self.d_symmetric = int(d_symmetric)
        self.r_pm, self.t_mm = (None, None) if self.d_symmetric else (z(), z())
        self.j0_p = torch.zeros((nSpec, N), dtype=dt, device=dev)
        self.j0_m = torch.zeros((nSpec, N), dtype=dt, device=dev)
        self.N, self.nSpec, self.shared, self.dtype = N, nSpec, shared, dt

    def cstruct(self):
        return _lib.vsm_added(self.r_mp.data_ptr(), self.t_pp.data_ptr(),
                              0 if self.r_pm is None else self.r_pm.data_ptr(),
                              0 if self.t_mm is None else self.t_mm.data_ptr(),
                              self.j0_p.data_ptr(), self.j0_m.data_ptr(), 0 if self.shared else self.N * self.N,
                              self.d_symmetric, 0)


def _composite_field(name):
    """A CompositeLayer array as a property.  Read: bring the array up to date first (materialize).  Assign: the caller rebinds the
    array -- CompositeLayer is a mutable struct in the reference -- after the native copy, if any, has been written out."""
    def fget(self):
        return self.materialize()._arr[name]

    def fset(self, value):
        self.materialize()._arr[name] = value
    return property(fget, fset)


class CompositeLayer:
    """src/CoreRT/types.jl CompositeLayer."""

    def __init__(self, FT, arch, N, nSpec):
        dev, dt = devi(arch), _torch_dtype(FT)
        z = lambda: torch.zeros((nSpec, N, N), dtype=dt, device=dev)
        self._native = None       # NativeSlot while rt_kernel_ keeps this composite in the kernels' strip layout
        self._native_ws = None    # its workspace (grow-only, reused from moment to moment)
        # the reference's arrays; read through the properties below, which bring them up to date first
        self._arr = dict(R_mp=z(), R_pm=z(), T_pp=z(), T_mm=z(), J0_p=torch.zeros((nSpec, N), dtype=dt, device=dev),
                         J0_m=torch.zeros((nSpec, N), dtype=dt, device=dev))
        self.N, self.nSpec, self.dtype = N, nSpec, dt

    R_mp, R_pm, T_pp = _composite_field("R_mp"), _composite_field("R_pm"), _composite_field("T_pp")
    T_mm, J0_p, J0_m = _composite_field("T_mm"), _composite_field("J0_p"), _composite_field("J0_m")

    def materialize(self):
        """If rt_kernel_ holds this composite in native layout (a vsm_run of one Fourier moment), write it into the reference's
        arrays (vsm_run_export) and release the run.  Every consumer reaches the arrays through cstruct(), which calls this:
        interaction_, create_surface_layer_'s interaction, postprocessing_vza_, interaction_hdrf_, copy_added_to_composite_."""
        slot = self._native
        if slot is None:
            return self
        self._native = None
        try:
            cc = (_lib.vsm_composite * 1)(self._cstruct_raw())
            _lib.call("vsm_run_export", self.dtype, slot.run, cc, _stream_ptr())
        finally:
            _lib.lib().vsm_run_destroy(slot.run)
        return self

    def drop_native(self):
        """Forget the native copy without exporting it (a TOA layer overwrites the composite)."""
        slot, self._native = self._native, None
        if slot is not None:
            _lib.lib().vsm_run_destroy(slot.run)

    def _cstruct_raw(self):
        a = self._arr
        return _lib.vsm_composite(a["R_mp"].data_ptr(), a["R_pm"].data_ptr(), a["T_pp"].data_ptr(), a["T_mm"].data_ptr(),
                                  a["J0_p"].data_ptr(), a["J0_m"].data_ptr())

    def cstruct(self):
        self.materialize()
        return self._cstruct_raw()

    def __del__(self):
        try:
            self.drop_native()
        except Exception:
            pass


def make_added_layer(FT, arch, dims, nSpec, shared=False, d_symmetric=0) -> AddedLayer:
    _require_gpu(arch)
    return AddedLayer(FT, arch, dims[0], nSpec, shared, d_symmetric)


def make_composite_layer(FT, arch, dims, nSpec) -> CompositeLayer:
    _require_gpu(arch)
    return CompositeLayer(FT, arch, dims[0], nSpec)


@dataclass
class DeviceLayerOptics:
    """CoreScatteringOpticalProperties after `expandOpticalProperties` (device side).
    Z is NOT replicated over the spectral axis when it is shared (z_stride = 0)."""
    tau: torch.Tensor      # [S]
    varpi: torch.Tensor    # [S]
    Zpp: torch.Tensor      # (1|S, N, N) layout tensor
    Zmp: torch.Tensor
    max_tau_varpi: float   # host copy of maximum(τ .* ϖ) (rt_kernel.jl:197) -- avoids a device sync
    tau_h: np.ndarray      # host copies used by get_dtau_ndoubl
    varpi_h: np.ndarray
    # several scatterers: Zpp/Zmp hold the ncomp COMPONENT matrices (ncomp, N, N) and fcomp (S, ncomp) the per-point
    # weights; Z = sum_k fcomp[s,k] Z_k is formed where it is consumed (vsm_layer_forward_mix) or by materialize()
    fcomp: Optional[torch.Tensor] = None
    # Stokes coupling mask of THIS layer's phase matrices at this moment (vsm_stokes_coupling; bit 4a+b); None: not known yet --
    # layer_coupling() computes it on the device (one small D2H) and keeps it here
    coupling: Optional[int] = None

    @property
    def z_stride(self):
        if self.fcomp is not None:
            raise _lib.VSMError("component-mixed layer optics: call materialize() for kernels that take Z[N,N,S]")
        N = self.Zpp.shape[-1]
        return 0 if self.Zpp.shape[0] == 1 else N * N

    def materialize(self) -> "DeviceLayerOptics":
        """Z[N,N,S] = sum_k fcomp[s,k] Z_k on the device (vsm_mix_Z; types.jl:1262-1292)."""
        if self.fcomp is None:
            return self
        S, C = self.fcomp.shape
        N = self.Zpp.shape[-1]
        Zpp = torch.empty((S, N, N), dtype=self.Zpp.dtype, device=self.Zpp.device)
        Zmp = torch.empty_like(Zpp)
        _lib.call("vsm_mix_Z", Zpp.dtype, N, S, C, _ptr(self.Zpp), _ptr(self.Zmp), _ptr(self.fcomp), _ptr(Zpp), _ptr(Zmp),
                  _stream_ptr())
        return DeviceLayerOptics(self.tau, self.varpi, Zpp, Zmp, self.max_tau_varpi, self.tau_h, self.varpi_h)


def expandOpticalProperties(p: H.CoreScatteringOpticalProperties, arch, FT) -> DeviceLayerOptics:
    """compEffectiveLayerProperties.jl:106-117 + H2D."""
    conv = array_type(arch)
    tau = np.atleast_1d(p.tau).astype(FT)
    varpi = np.broadcast_to(np.asarray(p.varpi, dtype=FT), tau.shape).copy()
    return DeviceLayerOptics(conv(tau), conv(varpi), to_device_matrix(p.Zpp, arch, FT), to_device_matrix(p.Zmp, arch, FT),
                             float(np.max(tau * varpi)), tau, varpi)


# ----------------------------------------------------------------------------
# CoreKernel
# ----------------------------------------------------------------------------
def elemental_doubling_(pol: H.PolarizationType, tau_sum: torch.Tensor, dtau: torch.Tensor, F0: torch.Tensor,
                        props: DeviceLayerOptics, m: int, ndoubl: int, dq: DeviceQuad, added: AddedLayer):
    """elemental! followed by doubling! (one fused launch when N fits on-chip)."""
    q, a = dq.cstruct(), added.cstruct()
    _lib.call("vsm_elemental_doubling", added.dtype, C.byref(q), added.nSpec, m, ndoubl, _ptr(dtau), _ptr(props.varpi),
              _ptr(tau_sum), _ptr(F0), _ptr(props.Zpp), _ptr(props.Zmp), props.z_stride, C.byref(a), _stream_ptr())


def layer_forward_(tau_sum, dtau, F0, props: DeviceLayerOptics, m: int, ndoubl: int, dq: DeviceQuad, toa: bool,
                   comp: CompositeLayer, added: AddedLayer):
    """The scattering branch of rt_kernel! (rt_kernel.jl:204-249): elemental! + doubling! + (TOA copy | interaction!(::_11))."""
    q, a, c = dq.cstruct(), added.cstruct(), comp.cstruct()
    if props.fcomp is not None:
        _lib.call("vsm_layer_forward_mix", comp.dtype, C.byref(q), comp.nSpec, m, ndoubl, _ptr(dtau), _ptr(props.varpi),
                  _ptr(tau_sum), _ptr(F0), int(props.fcomp.shape[1]), _ptr(props.Zpp), _ptr(props.Zmp), _ptr(props.fcomp),
                  None, 1 if toa else 0, C.byref(c), C.byref(a), _stream_ptr())
        return
    _lib.call("vsm_layer_forward", comp.dtype, C.byref(q), comp.nSpec, m, ndoubl, _ptr(dtau), _ptr(props.varpi),
              _ptr(tau_sum), _ptr(F0), _ptr(props.Zpp), _ptr(props.Zmp), props.z_stride, 1 if toa else 0, C.byref(c),
              C.byref(a), _stream_ptr())


def elemental_(pol, tau_sum, dtau, F0, props: DeviceLayerOptics, m, ndoubl, dq: DeviceQuad, added: AddedLayer):
    """elemental! alone (elemental.jl:174-230)."""
    q, a = dq.cstruct(), added.cstruct()
    _lib.call("vsm_elemental", added.dtype, C.byref(q), added.nSpec, m, ndoubl, _ptr(dtau), _ptr(props.varpi),
              _ptr(tau_sum), _ptr(F0), _ptr(props.Zpp), _ptr(props.Zmp), props.z_stride, C.byref(a), _stream_ptr())


def doubling_(pol, expk: torch.Tensor, ndoubl: int, added: AddedLayer):
    """doubling! alone (doubling.jl:38-131), operator-for-operator."""
    n = _lib.lib().vsm_doubling_work_elems(added.N, added.nSpec)
    work = _lib.poison(torch.empty(max(int(n), 1), dtype=added.dtype, device=added.r_mp.device))
    a = added.cstruct()
    _lib.call("vsm_doubling", added.dtype, added.N, pol.n, added.nSpec, ndoubl, _ptr(expk), C.byref(a), _ptr(work),
              _stream_ptr())


def zero_added_noscat_(added: AddedLayer, tau: torch.Tensor, dq: DeviceQuad):
    q, a = dq.cstruct(), added.cstruct()
    _lib.call("vsm_noscat_layer", added.dtype, C.byref(q), added.nSpec, _ptr(tau), C.byref(a), _stream_ptr())


def copy_added_to_composite_(comp: CompositeLayer, added: AddedLayer):
    a, c = added.cstruct(), comp.cstruct()
    _lib.call("vsm_copy_added_to_composite", added.dtype, added.N, added.nSpec, C.byref(a), C.byref(c), _stream_ptr())


_work_cache = {}


def layer_forward_multi_(tau_sum, dtau, F0, props_list, ms, ndoubl: int, dq: DeviceQuad, toa: bool, comps, added: AddedLayer):
    """The scattering branch of rt_kernel! (rt_kernel.jl:204-249) for SEVERAL Fourier moments of one layer (same dtau, varpi,
    tau_sum, F0; per moment its Z and its CompositeLayer): one launch where the strip kernel takes the shape
    (vsm_layer_forward_multi), else moment by moment inside the library."""
    nm = len(ms)
    p0 = props_list[0]
    q, a = dq.cstruct(), added.cstruct()
    dtype = comps[0].dtype
    carr = (type(comps[0].cstruct()) * nm)(*[c.cstruct() for c in comps])
    marr = (C.c_int * nm)(*[int(m) for m in ms])
    zpp = (C.c_void_p * nm)(*[p.Zpp.data_ptr() for p in props_list])
    zmp = (C.c_void_p * nm)(*[p.Zmp.data_ptr() for p in props_list])
    ncomp = 0 if p0.fcomp is None else int(p0.fcomp.shape[1])
    _lib.call("vsm_layer_forward_multi", dtype, C.byref(q), comps[0].nSpec, nm, marr, ndoubl, _ptr(dtau), _ptr(p0.varpi),
              _ptr(tau_sum), _ptr(F0), ncomp, zpp, zmp, 0 if ncomp else p0.z_stride, _ptr(p0.fcomp), None, 1 if toa else 0,
              carr, C.byref(a), _stream_ptr())


def run_layer_native_(run, nm: int, ndoubl: int, dtau, varpi, tau_sum, F0, ncomp: int, zpp, zmp, z_stride: int, fcomp, toa: bool,
                      layer_coupling=None, dtype=torch.float64):
    """rt_kernel!(::noRS) of one scattering layer for the nm Fourier moments of a native-layout run (vsm_run_layer: per class of
    sub-problems the elemental pre-pass and ONE layer launch); zpp / zmp: ctypes arrays of the moments' Z device pointers;
    layer_coupling: ctypes int array, per moment the Stokes coupling mask of THIS layer's phase matrices (None: the run's)."""
    _lib.call("vsm_run_layer", dtype, run, ndoubl, _ptr(dtau), _ptr(varpi), _ptr(tau_sum), _ptr(F0), ncomp, zpp, zmp, z_stride,
              _ptr(fcomp), 1 if toa else 0, layer_coupling, _stream_ptr())


def interaction_(scattering_interface: str, comp: CompositeLayer, added: AddedLayer, oplevel: bool = False,
                 work: Optional[torch.Tensor] = None):
    """interaction! (interaction.jl:268-285).  `oplevel=True` forces the operator-for-operator path
    (batched products + batch_inv!, like the reference executes it) instead of the fused kernel."""
    a, c = added.cstruct(), comp.cstruct()
    N, S = comp.N, comp.nSpec
    fused = (scattering_interface == "11" and not oplevel
             and N <= _lib.lib().vsm_fused_max_n(8 if comp.dtype == torch.float64 else 4))
    if not fused and work is None:
        key = (N, S, comp.dtype, str(comp.R_mp.device))
        work = _work_cache.get(key)
        if work is None:
            _work_cache.clear()
            work = _lib.poison(torch.empty(int(_lib.lib().vsm_interaction_work_elems(N, S)), dtype=comp.dtype, device=comp.R_mp.device))
            _work_cache[key] = work
    name = "vsm_interaction_oplevel" if oplevel else "vsm_interaction"
    _lib.call(name, comp.dtype, IFACE[scattering_interface], N, S, C.byref(c), C.byref(a), _ptr(work), _stream_ptr())


_phi_cache = {}


def brdf_azimuth_quadrature(arch, FT):
    """The azimuth nodes / weights of reflectance(): 100-point Gauss-Legendre on [0, pi] (coxmunk_surface.jl:394-395,
    CanopyOptics.gauleg), as device vectors."""
    key = (str(devi(arch)), np.dtype(FT).str)
    if key not in _phi_cache:
        phi, w = H.gauleg(H.NQUAD_PHI_BRDF, 0.0, math.pi)
        conv = array_type(arch)
        _phi_cache[key] = (conv(phi.astype(FT)), conv(w.astype(FT)))
    return _phi_cache[key]


def _coxmunk_cstruct(surf: H.CoxMunkSurface, dtype):
    nw = H.get_n_water(surf)
    cls = _lib.vsm_coxmunk_f64 if dtype == torch.float64 else _lib.vsm_coxmunk_f32
    return cls(float(surf.wind_speed), nw.real, nw.imag, float(surf.whitecap_albedo), int(bool(surf.include_whitecaps)),
               int(bool(surf.shadowing)))


BRDF_SURFACES = (H.CoxMunkSurface, H.rpvSurfaceScalar, H.RossLiSurfaceScalar)


def reflectance(surf, dq: DeviceQuad, m: int, arch, FT, deriv: bool = False):
    """reflectance(surf, pol_type, qp_mu, m) / reflectance_and_deriv (coxmunk_surface.jl:381-460) on the device; for the
    kernel-driven land BRDFs (rpvSurfaceScalar, RossLiSurfaceScalar: rpv_surface.jl:160-190) the block is a per-moment scene
    constant evaluated on the host and uploaded.  Returns layout tensors (N, N) [column-major rho[N,N]]; drho/dU is None unless deriv."""
    if isinstance(surf, (H.rpvSurfaceScalar, H.RossLiSurfaceScalar)):
        if deriv:
            raise _lib.VSMError("%s has no linearization (types.jl:472-512)" % type(surf).__name__)
        return to_device_matrix(H.brdf_reflectance(surf, dq.n_stokes, dq.host.qp_mu.astype(np.float64), m), arch, FT)[0].contiguous(), None
    N = int(dq.mu.numel())
    rho = torch.empty((N, N), dtype=dq.dtype, device=dq.mu.device)
    drho = torch.empty_like(rho) if deriv else None
    phi, w = brdf_azimuth_quadrature(arch, FT)
    cs, q = _coxmunk_cstruct(surf, dq.dtype), dq.cstruct()
    _lib.call("vsm_coxmunk_reflectance", dq.dtype, C.byref(cs), C.byref(q), m, int(phi.numel()), _ptr(phi), _ptr(w), _ptr(rho),
              _ptr(drho), _stream_ptr())
    return rho, drho


def create_surface_layer_(surface, added_surface: AddedLayer, m: int, dq: DeviceQuad, tau_sum: torch.Tensor, rho=None,
                          arch=None, FT=None):
    """create_surface_layer!: LambertianSurfaceScalar (lambertian_surface.jl:41-95; `surface` may be the bare albedo) or a
    BRDF surface through its Fourier reflectance block (rpv_surface.jl:51-97; CoxMunkSurface: coxmunk_surface.jl:381-460).
    `rho` = a precomputed reflectance(surface, pol, qp_mu, m) block (Scene caches one per moment)."""
    q, a = dq.cstruct(), added_surface.cstruct()
    if isinstance(surface, (H.LambertianSurfaceLegendre, H.LambertianSurfaceSpline)):
        # lambertian_surface.jl:97-213: `rho` = the per-point albedo [nSpec] (device), one r-+ block per spectral point
        if added_surface.shared or rho is None:
            raise _lib.VSMError("spectral Lambertian surfaces need a per-point surface AddedLayer and their albedo spectrum")
        _lib.call("vsm_lambertian_surface_spectral", added_surface.dtype, C.byref(q), added_surface.nSpec, m, _ptr(rho),
                  _ptr(tau_sum), C.byref(a), _stream_ptr())
        return
    if not added_surface.shared:
        raise _lib.VSMError("surface AddedLayer must be allocated with shared=True")
    if isinstance(surface, H.LambertianSurfaceScalar):
        surface = surface.albedo
    if isinstance(surface, (int, float)):
        alb = C.c_double(surface) if added_surface.dtype == torch.float64 else C.c_float(surface)
        _lib.call("vsm_lambertian_surface", added_surface.dtype, C.byref(q), added_surface.nSpec, m, alb, _ptr(tau_sum),
                  C.byref(a), _stream_ptr())
        return
    if isinstance(surface, BRDF_SURFACES):
        if rho is None:
            rho, _ = reflectance(surface, dq, m, arch, FT)
        _lib.call("vsm_brdf_surface", added_surface.dtype, C.byref(q), added_surface.nSpec, m, _ptr(rho), _ptr(tau_sum),
                  C.byref(a), _stream_ptr())
        return
    raise _lib.VSMError("surface %r is not built in this backend (LambertianSurfaceScalar / Legendre / Spline, CoxMunkSurface, "
                        "rpvSurfaceScalar, RossLiSurfaceScalar)"
                        % (surface,))


def apply_ss_correction_(R_SFI: torch.Tensor, surf: H.CoxMunkSurface, pol, vza, vaz, mu0, tau_total: torch.Tensor, m_max: int,
                         arch, FT):
    """apply_ss_correction! (coxmunk_surface.jl:481-545; rt_run.jl:520-524): R_SFI (S, n, nV) layout tensor, in place.
    Returns the per-geometry coefficients [nV, n] (device (n, nV) layout tensor) for inspection."""
    nV, S = len(vza), int(R_SFI.shape[0])
    ctype = C.c_double if R_SFI.dtype == torch.float64 else C.c_float
    mu_v = (ctype * nV)(*[float(FT(H.cosd(v))) for v in vza])
    dph = (ctype * nV)(*[float(FT(math.radians(float(a)))) for a in vaz])
    coef = torch.empty((pol.n, nV), dtype=R_SFI.dtype, device=R_SFI.device)
    phi, w = brdf_azimuth_quadrature(arch, FT)
    cs = _coxmunk_cstruct(surf, R_SFI.dtype)
    _lib.call("vsm_coxmunk_ss_correction", R_SFI.dtype, C.byref(cs), pol.n, S, nV, mu_v, dph, ctype(float(mu0)), int(m_max),
              int(phi.numel()), _ptr(phi), _ptr(w), _ptr(tau_total), _ptr(coef), _ptr(R_SFI), _stream_ptr())
    return coef


def postprocessing_vza_(pol: H.PolarizationType, comp: CompositeLayer, vza, vaz, qp: H.QuadPoints, m: int, weight: float,
                        R_SFI: torch.Tensor, T_SFI: torch.Tensor):
    """postprocessing_vza! noRS / SFI (postprocessing_vza.jl:23-94).  R_SFI/T_SFI: (S, nStokes, nVZA) layout tensors
    of the reference's [nVZA, nStokes, nSpec] arrays."""
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
    _lib.call("vsm_postprocess_vza", comp.dtype, comp.N, n, comp.nSpec, nV, row0, w, _ptr(comp.J0_m), _ptr(comp.J0_p),
              _ptr(R_SFI), _ptr(T_SFI), _stream_ptr())


def _vza_rows_weights(pol, vza, vaz, qp, m, weight, dtype):
    """_precompute_vza_weights (postprocessing_vza.jl): first row of each viewing stream and weight * {cos, cos, sin, sin}(m vaz)."""
    n, nV = pol.n, len(vza)
    row0 = (C.c_int * nV)()
    ctype = C.c_double if dtype == torch.float64 else C.c_float
    w = (ctype * (nV * n))()
    for v in range(nV):
        imu = int(np.argmin(np.abs(qp.qp_mu - qp.qp_mu.dtype.type(H.cosd(vza[v])))))
        row0[v] = imu * n
        c, s = H.cosd(m * vaz[v]), H.sind(m * vaz[v])
        for k in range(n):
            w[v + nV * k] = weight * [c, c, s, s][k]
    return row0, w


def interaction_hdrf_(pol, comp: CompositeLayer, added_surface: AddedLayer, m: int, dq: DeviceQuad, vza, vaz, qp, weight: float,
                      hdr_J: torch.Tensor, hdr: torch.Tensor, bhr_uw: torch.Tensor, bhr_dw: torch.Tensor):
    """interaction_hdrf! + postprocessing_vza_hdrf! (CoreKernel/interaction_hdrf.jl:4-42, tools/postprocessing_vza.jl:103-115):
    hdr (S, n, nV) += w * (r-+_surf J0+ + j0-_surf)[rows(vza)]; bhr_uw / bhr_dw (S, n) written for m == 0."""
    q, c, a = dq.cstruct(), comp.cstruct(), added_surface.cstruct()
    _lib.call("vsm_interaction_hdrf", comp.dtype, C.byref(q), comp.nSpec, m, C.byref(c), C.byref(a), _ptr(hdr_J), _ptr(bhr_uw),
              _ptr(bhr_dw), _stream_ptr())
    row0, w = _vza_rows_weights(pol, vza, vaz, qp, m, weight, comp.dtype)
    _lib.call("vsm_postprocess_vza_hdrf", comp.dtype, comp.N, pol.n, comp.nSpec, len(vza), row0, w, _ptr(hdr_J), _ptr(hdr),
              _stream_ptr())


def init_layer(props: DeviceLayerOptics, qp: H.QuadPoints, FT, numerics: H.RTNumericalParameters, arch):
    """rt_kernel.jl:339-349: (dτ, ndoubl) -- expk = exp(-dτ/μ₀) is formed inside the fused kernel."""
    dtau_h, nd = H.get_dtau_ndoubl(props.tau_h, props.varpi_h, qp, FT, numerics)
    return array_type(arch)(dtau_h), nd


# ---- the per-CompositeLayer registry behind rt_kernel_ (the drop-in form of the native-layout run) -------------------------------
# The reference's driver (rt_run.jl:383-453) loops `for m` outside `for iz` and hands rt_kernel! ONE CompositeLayer; nothing else
# reads that composite until the surface interaction (rt_run.jl:455-470).  rt_kernel_ therefore may keep it in the layer kernels'
# strip layout from the TOA call (iz == 1: a one-moment vsm_run is created for it) to the first consumer that needs the reference's
# [N,N,S] arrays (CompositeLayer.cstruct -> materialize -> vsm_run_export).  julia/vSmartMOMROCmExt.jl does the same with an IdDict.
@dataclass
class NativeSlot:
    run: C.c_void_p
    m: int
    coupling: int          # the run's Stokes coupling mask (vsm_run_create)
    grp_of: List[int]      # block of each Stokes component under that mask


def stokes_groups(n_stokes: int, coupling: int) -> List[int]:
    """Block (connected component of the symmetrised mask) of each Stokes component -- the rule of vsm_run_create."""
    adj = [[a == b or coupling < 0 or bool((coupling >> (4 * a + b)) & 1) or bool((coupling >> (4 * b + a)) & 1)
            for b in range(n_stokes)] for a in range(n_stokes)]
    grp = [-1] * n_stokes
    ng = 0
    for a in range(n_stokes):
        if grp[a] >= 0:
            continue
        grp[a], stack = ng, [a]
        while stack:
            x = stack.pop()
            for b in range(n_stokes):
                if adj[x][b] and grp[b] < 0:
                    grp[b] = ng
                    stack.append(b)
        ng += 1
    return grp


def layer_coupling(props: DeviceLayerOptics, N: int, n_stokes: int) -> int:
    """The Stokes coupling mask of a layer's phase matrices (OR over its scatterers / spectral points), from the data
    (vsm_stokes_coupling).  Scene fills props.coupling from its per-scatterer masks; a bare caller pays one small D2H here."""
    if props.coupling is None:
        Zpp, Zmp = props.Zpp.contiguous(), props.Zmp.contiguous()
        nb = int(Zpp.shape[0])
        mask = torch.zeros(nb, dtype=torch.int32, device=Zpp.device)
        for b0 in range(0, nb, 65535):
            b1 = min(nb, b0 + 65535)
            _lib.call("vsm_stokes_coupling", Zpp.dtype, N, n_stokes, b1 - b0, _ptr(Zpp[b0:b1]), _ptr(Zmp[b0:b1]),
                      C.c_void_p(mask[b0:b1].data_ptr()), _stream_ptr())
        props.coupling = int(np.bitwise_or.reduce(mask.cpu().numpy().astype(np.int64)))
    return props.coupling


def run_supported(dtype, N: int, ns: int, mask: int) -> bool:
    """vsm_run_supported of the model's float type: blocks of <= 96 rows (FP64), <= 128 rows (Float32: the FP32 native kernels)."""
    L = _lib.lib()
    return (L.vsm_run_supported_f32 if dtype == torch.float32 else L.vsm_run_supported)(int(N), int(ns), int(mask)) != 0


def run_workspace_bytes(dtype, N: int, ns: int, S: int, nm: int, carr) -> int:
    L = _lib.lib()
    return int((L.vsm_run_workspace_bytes_f32 if dtype == torch.float32 else L.vsm_run_workspace_bytes)(N, ns, S, nm, carr))


def _native_open(comp: CompositeLayer, dq: DeviceQuad, m: int, mask: int, import_arrays: bool) -> bool:
    """make_composite_layer in native layout for ONE Fourier moment (vsm_run_create on the composite's own workspace);
    import_arrays: continue from the composite's current [N,N,S] arrays (vsm_run_import).  False: the blocks do not fit."""
    L = _lib.lib()
    N, ns, S = comp.N, dq.n_stokes, comp.nSpec
    if not run_supported(comp.dtype, N, ns, mask):
        return False
    marr, carr = (C.c_int * 1)(int(m)), (C.c_int * 1)(int(mask))
    nbytes = run_workspace_bytes(comp.dtype, N, ns, S, 1, carr)
    ws = comp._native_ws
    if ws is None or ws.numel() * 8 < nbytes:
        ws = comp._native_ws = _lib.poison(torch.empty(max(nbytes // 8, 2), dtype=torch.float64, device=comp._arr["R_mp"].device))
    q = dq.cstruct()
    comp._native_q = (q, dq)                     # (the run keeps q's mu / wt pointers: they must outlive it)
    run = C.c_void_p()
    _lib.call("vsm_run_create", comp.dtype, C.byref(q), S, 1, marr, carr, _ptr(ws), nbytes, C.byref(run))
    if import_arrays:
        try:
            cc = (_lib.vsm_composite * 1)(comp._cstruct_raw())
            _lib.call("vsm_run_import", comp.dtype, run, cc, _stream_ptr())
        except Exception:
            L.vsm_run_destroy(run)
            raise
    comp._native = NativeSlot(run, int(m), int(mask), stokes_groups(ns, mask))
    return True


def _rt_kernel_native(comp: CompositeLayer, props: DeviceLayerOptics, tau_sum, m: int, dq: DeviceQuad, iz: int, F0, dtau,
                      ndoubl: int) -> bool:
    """The scattering "11" / TOA branch of rt_kernel! on the native copy of `comp` (vsm_run_layer).  The blocks of the run are the
    connected Stokes components of the phase matrices seen SO FAR (data, vsm_stokes_coupling: at m = 0 no phase matrix couples (I,Q)
    with (U,V), compute_Z_matrices.jl:26-110; Rayleigh leaves U alone there): a layer whose matrices couple two of the run's blocks
    (the first aerosol layer of a column, say) re-opens the run under the wider mask (export -> create -> import: once per
    widening).  False: not taken (shape outside the native kernels) -- the caller continues on the reference's arrays, which
    cstruct() brings up to date."""
    N, ns, S = comp.N, dq.n_stokes, comp.nSpec
    ncomp = 0 if props.fcomp is None else int(props.fcomp.shape[1])
    if ncomp > 4 or S == 0:
        return False
    lc = layer_coupling(props, N, ns)
    if iz == 1:
        comp.drop_native()                       # copy_added_to_composite! overwrites whatever the composite held
        if not _native_open(comp, dq, m, lc, False):
            return False
    slot = comp._native
    if slot is None or slot.m != m:
        return False
    g = slot.grp_of
    if any((lc >> (4 * a + b)) & 1 and g[a] != g[b] for a in range(ns) for b in range(ns)):
        wider = slot.coupling | lc
        comp.materialize()
        if not _native_open(comp, dq, m, wider, True):
            return False
        slot = comp._native
    zpp, zmp = (C.c_void_p * 1)(props.Zpp.data_ptr()), (C.c_void_p * 1)(props.Zmp.data_ptr())
    lcarr = (C.c_int * 1)(int(lc))
    run_layer_native_(slot.run, 1, int(ndoubl), dtau, props.varpi, tau_sum, F0, ncomp, zpp, zmp,
                      0 if ncomp else props.z_stride, props.fcomp, iz == 1, lcarr, comp.dtype)
    return True


def rt_kernel_(pol, added: AddedLayer, comp: CompositeLayer, props: DeviceLayerOptics, scattering_interface: str,
               tau_sum: torch.Tensor, m: int, dq: DeviceQuad, arch, iz: int, F0: torch.Tensor, FT,
               numerics: H.RTNumericalParameters, dtau: Optional[torch.Tensor] = None, ndoubl: Optional[int] = None,
               trace: Optional[list] = None, work: Optional[torch.Tensor] = None, keep_added: bool = False):
    """rt_kernel!(::noRS, ...) (rt_kernel.jl:175-250).  iz is 1-based.  dtau/ndoubl may be
    passed pre-computed (they only depend on the layer optics).  keep_added: leave the doubled layer in `added` (no fused
    layer step) -- a later non-scattering layer reads its j0+ (zero_added_noscat! never writes it, rt_helpers.jl:174-180)."""
    scatter = props.max_tau_varpi > 2 * np.finfo(FT).eps
    nd = 0
    if scatter:
        if dtau is None:
            dtau, ndoubl = init_layer(props, dq.host, FT, numerics, arch)
        nd = ndoubl
        if (iz == 1 or scattering_interface == "11") and not keep_added:
            # the whole layer step in one call (one launch when the strip kernels take the shape)
            if trace is not None:
                trace.append(dict(iz=iz, m=m, scatter=True, ndoubl=nd, iface=scattering_interface))
            elif NATIVE_DROPIN and NATIVE_RUN and _rt_kernel_native(comp, props, tau_sum, m, dq, iz, F0, dtau, nd):
                return                      # the layer step ran on the composite's native copy (exported lazily)
            layer_forward_(tau_sum, dtau, F0, props, m, nd, dq, iz == 1, comp, added)
            return
        elemental_doubling_(pol, tau_sum, dtau, F0, props.materialize(), m, nd, dq, added)
    else:
        zero_added_noscat_(added, props.tau, dq)
    if trace is not None:
        trace.append(dict(iz=iz, m=m, scatter=bool(scatter), ndoubl=nd, iface=scattering_interface))
    if iz == 1:
        copy_added_to_composite_(comp, added)
    else:
        interaction_(scattering_interface, comp, added, work=work)


# ----------------------------------------------------------------------------
# rt_run
# ----------------------------------------------------------------------------
class Scene:
    """Everything `rt_run` needs, resident in HBM.  The raw optical depths (tau_rayl, tau_abs [nSpec, Nz], the aerosol
    tables) are uploaded ONCE (`upload()`); `prepare()` then builds the layer-kernel inputs on the device: tau, varpi,
    tau_sum per (point, layer), the per-point weights of the component phase matrices, max(tau*varpi) per layer
    (vsm_layer_optics), the Fourier moments Z(m) of every scatterer (vsm_compute_Z_moments), and -- after ONE small D2H of
    the per-layer maxima, from which the host derives ndoubl and the interface tags exactly like the reference's host
    reductions (rt_kernel.jl:197,282-283; compEffectiveLayerProperties.jl:87) -- dtau = tau / 2^ndoubl (vsm_layer_dtau).
    `run()` is device only.

    `spec_slice` selects the spectral shard this rank owns.  The optics pass always covers the FULL spectral axis (it is
    O(nSpec Nz) elementwise work, microseconds) so that ndoubl and the tags are batch-global on every rank without a
    collective; only the shard's columns feed the layer kernels.  `host_optics=True` builds the same inputs with the
    host mirror (host_model.constructLayerOpticsComponents) instead -- kept as a cross-check."""

    def __init__(self, model: H.RTModel, spec_slice: Optional[slice] = None, host_optics: bool = False,
                 full_added_layer: bool = False):
        arch, FT = model.architecture, model.float_type
        _require_gpu(arch)
        self.model, self.arch, self.FT = model, arch, FT
        pol, qp = model.polarization_type, model.quad_points
        self.pol, self.qp = pol, qp
        self.ss_correction = True   # Cox-Munk TMS term (tests switch it off to look at the Fourier-summed field)
        if isinstance(model.surface, H.CoxMunkSurface) and len(model.vza) > 64:
            # fail before the Fourier loop, not after it: vsm_coxmunk_ss_correction holds 64 viewing geometries per call
            raise _lib.VSMError("Cox-Munk scenes take at most 64 viewing geometries (TMS single-scattering correction); got %d"
                                % len(model.vza))
        self.host_optics = bool(host_optics)
        self.full_added_layer = bool(full_added_layer)   # the linearized run's kernels take the reference's full AddedLayer
        S_full, self.Nz = model.tau_rayl.shape
        self.S_full = S_full
        self.sl = spec_slice if spec_slice is not None else slice(0, S_full)
        self.lo, self.hi, _ = self.sl.indices(S_full)
        self.hi = max(self.hi, self.lo)
        self.S = self.hi - self.lo
        self.N = qp.Nquad * pol.n
        conv = array_type(arch)
        dt, dev = _torch_dtype(FT), devi(arch)
        self.dt, self.dev = dt, dev
        self.dq = device_quad(qp, pol, arch, FT)
        F0 = model.F0
        if F0 is None:
            F0 = np.zeros((pol.n, S_full))
            F0[0, :] = 1.0
        # sources (rt_run(model; sources = ...)): the solar slot lives inside the layer kernels (F0), every other source
        # has a slot of its own (here: :thermal)
        srcs = tuple(model.sources) if model.sources is not None else (H.SolarBeam(),)
        unknown = [s_ for s_ in srcs if not isinstance(s_, (H.SolarBeam, H.ThermalEmission))]
        if unknown:
            raise _lib.VSMError("rt_run: unsupported source %r (SolarBeam, ThermalEmission)" % (unknown[0],))
        if not any(isinstance(s_, H.SolarBeam) for s_ in srcs):
            F0 = np.zeros((pol.n, S_full))
        th = [s_ for s_ in srcs if isinstance(s_, H.ThermalEmission) and s_.B_layer is not None and np.any(s_.B_layer != 0)]
        self.thermal_B = None
        self.thermal_reset_noscat = bool(th and th[0].reset_slot_in_nonscattering_layers)
        self._thermal_carry = False
        if th:
            B = th[0].B_layer
            if B.shape[1] != S_full:
                raise _lib.VSMError("ThermalEmission: B_layer has %d spectral points, the model %d" % (B.shape[1], S_full))
            self.thermal_B = conv(np.ascontiguousarray(B[:, self.sl].astype(FT)))          # (rows, S)
        self.F0 = conv(np.ascontiguousarray(np.asarray(F0, dtype=FT)[:, self.sl].T))  # [n,S] col-major == (S,n)
        N, S, L, C_ = self.N, self.S, self.Nz, 1 + len(model.aerosol_optics)
        # device state of the optics pass (full spectral axis; layout [nSpec, Nz] column-major == tensors (Nz, nSpec))
        z = lambda *sh: torch.empty(sh, dtype=dt, device=dev)
        self.tau, self.varpi, self.dtau, self.tau_sum = z(L, S_full), z(L, S_full), z(L, S_full), z(L + 1, S_full)
        self.fcomp = z(L, S_full, C_) if C_ > 1 else None
        self.max_tw = z(L)
        self.nd_dev = torch.zeros(L, dtype=torch.int32, device=dev)
        self.Zc = [(z(C_, N, N), z(C_, N, N)) for _ in range(model.m_max + 1)]
        self.greek_dev = None
        self.moments = []
        # every layer scatters and N fits on chip -> all steps run in the fused kernels, which can derive
        # r+-/t-- by D-symmetry instead of moving them through HBM (decided in prepare(), allocated lazily)
        self.added = None
        self.spectral_surface = isinstance(model.surface, (H.LambertianSurfaceLegendre, H.LambertianSurfaceSpline))
        self.added_surface = make_added_layer(FT, arch, (N, N), S, shared=not self.spectral_surface)
        self.albedo_d = (conv(np.ascontiguousarray(H.surface_albedo_spectrum(model.surface, S_full, FT)[self.sl]))
                         if self.spectral_surface else None)
        self.composite = make_composite_layer(FT, arch, (N, N), S)
        self._composites = [self.composite]     # one per Fourier moment of a batch (run() allocates the others on first use)
        # the interaction work buffer belongs to the scene (a captured graph must not point into a shared cache)
        self.work = _lib.poison(torch.empty(max(int(_lib.lib().vsm_interaction_work_elems(N, max(S, 1))), 1), dtype=dt, device=dev))
        nV = len(model.vza)
        self.R_SFI = torch.zeros((S, pol.n, nV), dtype=dt, device=dev)
        self.T_SFI = torch.zeros((S, pol.n, nV), dtype=dt, device=dev)
        # HDRF / BHR diagnostics of the reference's 7-tuple (rt_run.jl:300-303,467-494,535)
        self.compute_hdrf = True
        self.hdr = torch.zeros((S, pol.n, nV), dtype=dt, device=dev)
        self.hdr_J = torch.zeros((S, N), dtype=dt, device=dev)
        self.bhr_uw = torch.zeros((S, pol.n), dtype=dt, device=dev)
        self.bhr_dw = torch.zeros((S, pol.n), dtype=dt, device=dev)
        self.upload()
        self.prepare()

    # -- inputs -----------------------------------------------------------------------------------------------------------
    def upload(self):
        """H2D of the scene's raw inputs: tau_rayl, tau_abs [nSpec, Nz] (FP64, the reference's model arrays) and the small
        aerosol / Greek tables.  Everything else is derived on the device by `prepare()`."""
        model = self.model
        conv = array_type(self.arch)
        L, nA = self.Nz, len(model.aerosol_optics)
        # numpy holds [S, L] row-major; the C ABI wants the reference's column-major [nSpec, Nz]: transpose on the device
        self.tau_rayl_d = conv(np.asarray(model.tau_rayl, dtype=np.float64)).t().contiguous()
        self.tau_abs_d = conv(np.asarray(model.tau_abs, dtype=np.float64)).t().contiguous()
        if nA:
            self.tau_aer_d = conv(np.ascontiguousarray(np.asarray(model.tau_aer, dtype=np.float64).T))   # (L, nAer)
            self.ssa_d = conv(np.array([ao.ssa for ao in model.aerosol_optics], dtype=np.float64))
            self.ftr_d = conv(np.array([ao.f_trunc for ao in model.aerosol_optics], dtype=np.float64))
            modes, self.zcomp = H.layer_mix_modes(model)
            self.mode_d = conv(np.ascontiguousarray(modes.T.astype(np.int32)))                          # (L, nAer)
        else:
            self.tau_aer_d = self.ssa_d = self.ftr_d = self.mode_d = None
            self.zcomp = [(False, 0)] * L
        if self.greek_dev is None:
            self.greek_dev = []
            for g in [model.greek_rayleigh] + [ao.greek_coefs for ao in model.aerosol_optics]:
                tab = np.stack([np.asarray(getattr(g, k), dtype=np.float64) for k in
                                ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")])
                self.greek_dev.append((conv(np.ascontiguousarray(tab)), tab.shape[1]))

    # -- device optics ----------------------------------------------------------------------------------------------------
    def prepare(self):
        model, FT, dt = self.model, self.FT, self.dt
        qp, L, S_full, N = self.qp, self.Nz, self.S_full, self.N
        nA = len(model.aerosol_optics)
        if self.host_optics:
            return self._prepare_host()
        _lib.call("vsm_layer_optics", dt, S_full, L, nA, _ptr(self.tau_rayl_d), _ptr(self.tau_abs_d),
                  C.c_double(float(model.varpi_Cabannes)), _ptr(self.tau_aer_d), _ptr(self.ssa_d), _ptr(self.ftr_d),
                  _ptr(self.mode_d), _ptr(self.tau), _ptr(self.varpi), _ptr(self.tau_sum), _ptr(self.fcomp), _ptr(self.max_tw),
                  _stream_ptr())
        if getattr(self, "_zc_of", None) is not self.greek_dev:
            # Z(m) of every scatterer depends on its Greek table and the quadrature only: scene constants like the tables (a step
            # re-uploads optical depths; `greek_dev = None` before upload() re-uploads the tables and recomputes these)
            q = self.dq.cstruct()
            for m in range(model.m_max + 1):
                Zpp, Zmp = self.Zc[m]
                for k, (gd, lmax) in enumerate(self.greek_dev):
                    _lib.call("vsm_compute_Z_moments", dt, C.byref(q), m, lmax, _ptr(gd), _ptr(Zpp[k]), _ptr(Zmp[k]), _stream_ptr())
            self._zc_of = self.greek_dev
            self.coupling_comp = None
        self._compute_coupling()
        mx = self.max_tw.cpu().numpy()          # the ONE device -> host hand-off of the optics pass (Nz scalars)
        nds, tags, tag = [], [], "00"
        for iz in range(L):
            tw = FT(mx[iz])
            scatter = bool(tw > 2 * np.finfo(FT).eps)
            nds.append(H.ndoubl_from_max(tw, qp, FT, model.numerics) if scatter else 0)
            tag = H.get_scattering_interface(tag, scatter, iz + 1)
            tags.append(tag)
        self.nd_dev.copy_(torch.as_tensor(np.asarray(nds, dtype=np.int32)))
        _lib.call("vsm_layer_dtau", dt, S_full, L, _ptr(self.nd_dev), _ptr(self.tau), _ptr(self.dtau), _stream_ptr())
        self._assemble(nds, tags, [float(v) for v in mx])

    def _assemble(self, nds, tags, maxima):
        """Per-moment / per-layer views into the optics tensors, as `run()` walks them."""
        model, FT, L, lo, hi = self.model, self.FT, self.Nz, self.lo, self.hi
        # The structure below is views into tensors the scene owns: a step that leaves ndoubl, the interface tags, the scatterers
        # per layer, the coupling masks and the buffers themselves what they were only refreshes the per-layer maxima (a small
        # batch is host bound: 726 property records of the ocean scene were 6 of the 24 ms of its linearized step).
        ptrs = tuple(t.data_ptr() for pair in self.Zc for t in pair) + tuple(
            0 if t is None else t.data_ptr() for t in (self.tau, self.varpi, self.dtau, self.tau_sum, getattr(self, "fcomp", None),
                                                       self.albedo_d if torch.is_tensor(self.albedo_d) else None))
        cc = getattr(self, "coupling_comp", None)
        def freeze(x):
            if isinstance(x, np.ndarray):
                return (x.shape, x.tobytes())
            return tuple(freeze(v) for v in x) if isinstance(x, (tuple, list)) else x
        surf_key = ((type(model.surface).__name__, freeze(dataclasses.astuple(model.surface)))
                    if dataclasses.is_dataclass(model.surface) else None)
        key = (tuple(nds), tuple(tags), tuple(self.zcomp), ptrs, lo, hi, None if cc is None else cc.tobytes(), surf_key,
               self.full_added_layer)
        if surf_key is not None and getattr(self, "_assemble_key", None) == key and getattr(self, "moments", None):
            for mom in self.moments:
                for iz, ly in enumerate(mom["layers"]):
                    ly["props"].max_tau_varpi = maxima[iz]
            return
        self._assemble_key = key
        self.moments = []
        for m in range(model.m_max + 1):
            Zpp, Zmp = self.Zc[m]
            layers = []
            for iz in range(L):
                mixed, k = self.zcomp[iz]
                if mixed:
                    Zp, Zm, fc = Zpp, Zmp, self.fcomp[iz, lo:hi]
                else:
                    Zp, Zm, fc = Zpp[k:k + 1], Zmp[k:k + 1], None
                props = DeviceLayerOptics(self.tau[iz, lo:hi], self.varpi[iz, lo:hi], Zp, Zm, maxima[iz], None, None, fc,
                                          coupling=self._layer_coupling(m, iz))
                layers.append(dict(props=props, iface=tags[iz], nd=nds[iz], dtau=self.dtau[iz, lo:hi],
                                   tau_sum=self.tau_sum[iz, lo:hi]))
            rho = self.albedo_d
            if isinstance(model.surface, BRDF_SURFACES):   # scene constant like Z(m): one N x N block per moment
                # (kept across prepare() calls for as long as the surface's parameters stay what they were: a step that only
                # changes the optical depths does not re-evaluate 22 Fourier blocks -- 5 of the 6 ms of the ocean scene's prepare)
                key = (type(model.surface).__name__, dataclasses.astuple(model.surface), m)
                cache = self.__dict__.setdefault("_rho_cache", {})
                if key not in cache:
                    if len(cache) > 4 * (model.m_max + 1):
                        cache.clear()
                    cache[key] = reflectance(model.surface, self.dq, m, self.arch, FT)[0]
                rho = cache[key]
            self.moments.append(dict(m=m, layers=layers, iface_surface=tags[-1], rho=rho, tau_sum_surface=self.tau_sum[L, lo:hi]))
        all11 = all(t == "11" for t in tags)
        fused_ok = self.N <= _lib.lib().vsm_fused_max_n(8 if np.dtype(FT) == np.float64 else 4)
        dsym = self.pol.n if (all11 and fused_ok and not self.full_added_layer) else 0
        if self.added is None or self.added.d_symmetric != dsym:
            self.added = make_added_layer(FT, self.arch, (self.N, self.N), self.S, d_symmetric=dsym)

    def _prepare_host(self):
        """The same inputs from the host mirror (numpy): the reference's own structure, per Fourier moment."""
        model, FT, L, S_full = self.model, self.FT, self.Nz, self.S_full
        conv = array_type(self.arch)
        nds = tags = maxima = None
        for m in range(model.m_max + 1):
            Zc_pp, Zc_mp, lods = H.constructLayerOpticsComponents(model, m)
            self.Zc[m][0].copy_(to_device_matrix(Zc_pp, self.arch, FT))
            self.Zc[m][1].copy_(to_device_matrix(Zc_mp, self.arch, FT))
            if m > 0:
                continue
            tags, tau_sum_all = H.extractEffectiveProps(lods, FT)
            nds, maxima = [], []
            zcomp = []
            for iz, lo in enumerate(lods):
                tau_full = np.atleast_1d(lo.tau).astype(FT)
                varpi_full = np.broadcast_to(np.asarray(lo.varpi, dtype=FT), tau_full.shape)
                tw = float(np.max(tau_full * varpi_full))
                scatter = tw > 2 * np.finfo(FT).eps
                dtau_full, nd = (H.get_dtau_ndoubl(tau_full, varpi_full, self.qp, FT, model.numerics) if scatter else (tau_full, 0))
                nds.append(nd)
                maxima.append(tw)
                self.tau[iz].copy_(conv(tau_full))
                self.varpi[iz].copy_(conv(np.ascontiguousarray(varpi_full)))
                self.dtau[iz].copy_(conv(np.ascontiguousarray(dtau_full)))
                self.tau_sum[iz].copy_(conv(tau_sum_all[:, iz].astype(FT)))
                if lo.coef.ndim == 1:
                    zcomp.append((False, int(np.argmax(lo.coef))))
                else:
                    zcomp.append((True, 0))
                    self.fcomp[iz].copy_(conv(np.ascontiguousarray(lo.coef.astype(FT))))
            self.tau_sum[L].copy_(conv(tau_sum_all[:, -1].astype(FT)))
            self.zcomp = zcomp
        self._compute_coupling()
        self._assemble(nds, tags, maxima)

    def _compute_coupling(self):
        """Which Stokes components the phase matrix of each scatterer couples, per Fourier moment (vsm_stokes_coupling: one mask
        per block of the Z stacks): components that no scatterer of the run couples walk the layers as independent sub-problems
        (vsm_run_*; for m = 0 the (I,Q) x (U,V) blocks of every phase matrix are exactly zero, compute_Z_matrices.jl:26-110), and
        a block that the scatterers of ONE layer leave exactly zero takes that layer as a diagonal step."""
        C_ = int(self.Zc[0][0].shape[0])
        if not self.host_optics and getattr(self, "coupling_comp", None) is not None:
            return      # device optics: Z(m) depends on the Greek tables and the quadrature only -- scene constants, like the masks
        if getattr(self, "_coupling_d", None) is None:
            self._coupling_d = torch.zeros((len(self.Zc), C_), dtype=torch.int32, device=self.dev)
        for m, (Zpp, Zmp) in enumerate(self.Zc):
            _lib.call("vsm_stokes_coupling", self.dt, self.N, self.pol.n, C_, _ptr(Zpp), _ptr(Zmp),
                      C.c_void_p(self._coupling_d[m].data_ptr()), _stream_ptr())
        self.coupling_comp = self._coupling_d.cpu().numpy().astype(np.int64)             # [moment, scatterer]
        self.coupling = [int(np.bitwise_or.reduce(row)) for row in self.coupling_comp]    # per moment: every scatterer of the run

    def _layer_coupling(self, m, iz):
        """The coupling mask of the phase matrices of layer iz at moment m: the scatterers present in the layer."""
        mixed, k = self.zcomp[iz]
        row = self.coupling_comp[m]
        return int(np.bitwise_or.reduce(row)) if mixed else int(row[k])

    def run(self, trace: Optional[list] = None, streams: Optional[list] = None):
        """The device-resident part of rt_run (rt_run.jl:383-517): Fourier loop -> layer loop ->
        surface -> interaction -> postprocessing.  Asynchronous; returns the device tensors.  `streams`: a list that receives,
        per Fourier moment, what the reference hands its `streams_callback` (rt_run.jl:496-516): clones of the composite layer's
        R-+, T++ and of J0-+ (per-source slots added) after the surface interaction, with m and the Fourier weight."""
        model, pol, FT = self.model, self.pol, self.FT
        self.R_SFI.zero_()
        self.T_SFI.zero_()
        self.hdr.zero_()
        if self.S == 0:          # a rank that owns no spectral point (world > nSpec): nothing to launch
            return self.R_SFI, self.T_SFI
        self.added.j0_p.zero_()  # a fresh make_added_layer: zero_added_noscat! never writes j0+ (rt_helpers.jl:174-180)
        # Fourier moments are independent until post-processing (rt_run.jl:383): groups of up to MOMENT_BATCH moments walk the
        # layers together, a scattering "11" / TOA layer step being ONE call for the group (vsm_layer_forward_multi: one launch
        # with three times the workgroups where the strip kernel takes the shape), each moment on a CompositeLayer of its own
        # (only when every layer scatters: a non-scattering layer reads the added layer's j0+ as the previous step left it --
        # zero_added_noscat! never writes it, rt_helpers.jl:174-180 -- which ties the moments to their sequential order)
        eps2 = 2 * np.finfo(FT).eps
        has_noscat = any(ly["props"].max_tau_varpi <= eps2 for ly in self.moments[0]["layers"])
        sequential = trace is not None or not MOMENT_BATCHING or has_noscat or REFERENCE_ORDER
        # small batches are launch-latency bound (C3: 2 points, 22 moments x 33 layers): as many moments per launch as fill the chip
        want = max(MOMENT_BATCH, min(MOMENT_BATCH_MAX, 4096 // max(self.S, 1)))
        nb = 1 if sequential else min(want, len(self.moments))
        while len(self._composites) < nb:
            self._composites.append(make_composite_layer(FT, self.arch, (self.N, self.N), self.S))
        # (REFERENCE_ORDER: every layer step is an rt_kernel_ call on the ONE CompositeLayer, like the reference's driver; the native
        # layout is then reached through rt_kernel_'s per-composite registry, NATIVE_DROPIN, instead of _run_layers_native)
        native = self._native_moments() if trace is None and not has_noscat and not REFERENCE_ORDER else set()
        for g0 in range(0, len(self.moments), nb):
            group = self.moments[g0:g0 + nb]
            comps = self._composites[:len(group)]
            nat = [k for k in range(len(group)) if (g0 + k) in native]
            if nat:   # the layer loop on the native-layout composite (vsm_run_*), then the reference's arrays for the surface step
                self._run_layers_native([group[k] for k in nat], [comps[k] for k in nat])
            rest = [k for k in range(len(group)) if (g0 + k) not in native]
            group_l, comps_l = [group[k] for k in rest], [comps[k] for k in rest]
            for iz in range(self.Nz if rest else 0):
                ly0 = group_l[0]["layers"][iz]
                if len(group_l) > 1 and ly0["props"].max_tau_varpi > eps2 and (iz == 0 or ly0["iface"] == "11"):
                    layer_forward_multi_(ly0["tau_sum"], ly0["dtau"], self.F0, [mom["layers"][iz]["props"] for mom in group_l],
                                         [mom["m"] for mom in group_l], ly0["nd"], self.dq, iz == 0, comps_l, self.added)
                    continue
                for mom, comp in zip(group_l, comps_l):
                    ly = mom["layers"][iz]
                    rt_kernel_(pol, self.added, comp, ly["props"], ly["iface"], ly["tau_sum"], mom["m"], self.dq,
                               self.arch, iz + 1, self.F0, FT, model.numerics, dtau=ly["dtau"], ndoubl=ly["nd"], trace=trace,
                               work=self.work, keep_added=has_noscat)
            for mom, comp in zip(group, comps):
                m = mom["m"]
                weight = FT(0.5 / math.pi) if m == 0 else FT(1.0 / math.pi)
                create_surface_layer_(model.surface, self.added_surface, m, self.dq, mom["tau_sum_surface"], rho=mom["rho"])
                interaction_(mom["iface_surface"], comp, self.added_surface, work=self.work)
                if self.compute_hdrf:
                    interaction_hdrf_(pol, comp, self.added_surface, m, self.dq, model.vza, model.vaz, self.qp, float(weight),
                                      self.hdr_J, self.hdr, self.bhr_uw, self.bhr_dw)
                postprocessing_vza_(pol, comp, model.vza, model.vaz, self.qp, m, float(weight), self.R_SFI, self.T_SFI)
                thermal_pass = self.thermal_B is not None and (m == 0 or (m == 1 and self._thermal_carry))
                if thermal_pass:
                    self._thermal_slot(mom, float(weight))
                if streams is not None:
                    Jm, Jp = comp.J0_m.clone(), comp.J0_p.clone()
                    if thermal_pass:
                        Jm += self._th_comp.J0_m
                        Jp += self._th_comp.J0_p
                    streams.append(dict(m=m, weight=float(weight), R_mp=comp.R_mp.clone(), T_pp=comp.T_pp.clone(), J0_m=Jm, J0_p=Jp))
        if isinstance(model.surface, H.CoxMunkSurface) and self.ss_correction:   # rt_run.jl:520-524 (SFI is always on here)
            apply_ss_correction_(self.R_SFI, model.surface, pol, model.vza, model.vaz, self.qp.mu0,
                                 self.moments[-1]["tau_sum_surface"], model.m_max, self.arch, FT)
        return self.R_SFI, self.T_SFI

    def _native_moments(self):
        """Indices of the Fourier moments whose layer loop runs on the native-layout composite (vsm_run_*): every layer scattering
        with the 11 interface (the only steps the run object takes), at most four scatterers per layer, and every block of coupled
        Stokes components within the native kernels' size (FP64: 96 rows; Float32 models: 128 rows, FP32 records and arithmetic)."""
        if not NATIVE_RUN or getattr(self, "coupling", None) is None or not self.moments:
            return set()
        eps2 = 2 * np.finfo(self.FT).eps
        for iz, ly in enumerate(self.moments[0]["layers"]):
            p = ly["props"]
            if p.max_tau_varpi <= eps2 or (iz > 0 and ly["iface"] != "11") or (p.fcomp is not None and p.fcomp.shape[1] > 4):
                return set()
        return {i for i, mom in enumerate(self.moments) if run_supported(self.dt, self.N, self.pol.n, int(self.coupling[mom["m"]]))}

    def _run_layers_native(self, group, comps):
        """rt_run's layer loop (rt_run.jl:383-453) for the moments of `group` with the CompositeLayer in kernel-native layout:
        vsm_run_create -> Nz x vsm_run_layer -> vsm_run_export into the reference-layout CompositeLayers `comps`."""
        L = _lib.lib()
        nm, N, S, ns = len(group), self.N, self.S, self.pol.n
        marr = (C.c_int * nm)(*[int(mom["m"]) for mom in group])
        carr = (C.c_int * nm)(*[int(self.coupling[mom["m"]]) for mom in group])
        nbytes = run_workspace_bytes(self.dt, N, ns, S, nm, carr)
        ws = getattr(self, "_native_ws", None)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = self._native_ws = _lib.poison(torch.empty(max(nbytes // 8, 2), dtype=torch.float64, device=self.dev))
        q = self.dq.cstruct()
        run = C.c_void_p()
        _lib.call("vsm_run_create", self.dt, C.byref(q), S, nm, marr, carr, _ptr(ws), nbytes, C.byref(run))
        try:
            for iz in range(self.Nz):
                ly0 = group[0]["layers"][iz]
                props = [mom["layers"][iz]["props"] for mom in group]
                p0 = props[0]
                zpp = (C.c_void_p * nm)(*[p.Zpp.data_ptr() for p in props])
                zmp = (C.c_void_p * nm)(*[p.Zmp.data_ptr() for p in props])
                ncomp = 0 if p0.fcomp is None else int(p0.fcomp.shape[1])
                lc = (C.c_int * nm)(*[self._layer_coupling(mom["m"], iz) for mom in group])
                run_layer_native_(run, nm, int(ly0["nd"]), ly0["dtau"], p0.varpi, ly0["tau_sum"], self.F0, ncomp, zpp, zmp,
                                  0 if ncomp else p0.z_stride, p0.fcomp, iz == 0, lc, self.dt)
            cc = (type(comps[0].cstruct()) * nm)(*[c.cstruct() for c in comps])
            _lib.call("vsm_run_export", self.dt, run, cc, _stream_ptr())
        finally:
            L.vsm_run_destroy(run)

    def _thermal_slot(self, mom, weight):
        """The `:thermal` per-source slot (rt_kernel.jl:204-232, doubling.jl:62-81, interaction.jl per-source recurrences,
        postprocessing_vza.jl:68-82): the solar slot's linear source recurrences driven by the thermal source
        (vsm_thermal_source between elemental! and doubling!, the slot's own expk = 1); its J0 is added to R_SFI / T_SFI.
        On an AddedLayer / CompositeLayer of its own (r, t, R, T evolve exactly like the solar pass's).

        State, as the reference carries it (default; ThermalEmission(reset_slot_in_nonscattering_layers=True) = the corrected
        variant): the slot's j0+- belong to the AddedLayer that is allocated once per run (rt_run.jl:326-335); only the scatter
        branch resets them (rt_kernel.jl:217-221), so a non-scattering layer interacts with the doubled slot of the last
        scattering layer before it, and a column that begins with non-scattering layers enters moment m = 1 with the slot the
        last scattering layer of moment m = 0 left behind -- that is the one case in which this pass runs for m = 1 (contribute!
        itself is m = 0 only; with a zero slot on entry every later moment stays zero)."""
        model, pol, FT = self.model, self.pol, self.FT
        N, S = self.N, self.S
        m = mom["m"]
        if getattr(self, "_th_added", None) is None:
            self._th_added = make_added_layer(FT, self.arch, (N, N), S, d_symmetric=0)
            self._th_surf = make_added_layer(FT, self.arch, (N, N), S, shared=not self.spectral_surface)
            self._th_comp = make_composite_layer(FT, self.arch, (N, N), S)
            self._th_F0 = torch.zeros_like(self.F0)
            self._th_ones = torch.ones(max(S, 1), dtype=self.dt, device=self.dev)
        added, comp = self._th_added, self._th_comp
        q = self.dq.cstruct()
        eps2 = 2 * np.finfo(FT).eps
        scat = [ly["props"].max_tau_varpi > eps2 for ly in mom["layers"]]
        keep_state = not self.thermal_reset_noscat          # the reference as written
        if m == 0:
            added.j0_p.zero_()                               # a fresh AddedLayer (rt_run.jl:326-335)
            added.j0_m.zero_()
            self._thermal_carry = keep_state and any(scat) and not scat[0] and model.m_max >= 1
        # FP64, 32 < N <= 60 / FP32, 64 < N <= 96: a scattering layer's slot in ONE fused launch (vsm_layer_forward_thermal: the strip layer kernel
        # with the thermal source and expk = 1; m = 0); everything else operator level, layer by layer on the same composite.  With
        # the reference's slot state and a non-scattering layer in the column the doubled slot must stay in the AddedLayer for the
        # layers below: no fused step then (as in the solar pass, rt_kernel_'s keep_added)
        fused = (THERMAL_FUSION and m == 0 and (all(scat) or not keep_state)
                 and _lib.lib().vsm_layer_thermal_fused(N, 1 if FT == np.float64 else 0) != 0)
        for iz, ly in enumerate(mom["layers"]):
            props = ly["props"]
            scatter = scat[iz]
            if (fused and scatter and (iz == 0 or ly["iface"] == "11") and iz < self.thermal_B.shape[0]
                    and (props.fcomp is None or props.fcomp.shape[1] <= 4)):
                c = comp.cstruct()
                ncomp = 0 if props.fcomp is None else int(props.fcomp.shape[1])
                _lib.call("vsm_layer_forward_thermal", self.dt, C.byref(q), S, ly["nd"], _ptr(ly["dtau"]), _ptr(props.varpi),
                          _ptr(self.thermal_B[iz]), ncomp, _ptr(props.Zpp), _ptr(props.Zmp),
                          0 if ncomp else props.z_stride, _ptr(props.fcomp), 1 if iz == 0 else 0, C.byref(c), _stream_ptr())
                continue
            if scatter:
                # elemental! with F0 = 0 leaves j0+- = 0: the slot reset of rt_kernel.jl:217-221
                elemental_(pol, ly["tau_sum"], ly["dtau"], self._th_F0, props.materialize(), m, ly["nd"], self.dq, added)
                if m == 0 and iz < self.thermal_B.shape[0]:           # contribute! (isotropic: m = 0 only)
                    a = added.cstruct()
                    _lib.call("vsm_thermal_source", self.dt, C.byref(q), S, _ptr(ly["dtau"]), _ptr(props.varpi),
                              _ptr(self.thermal_B[iz]), C.byref(a), _stream_ptr())
                self._th_ones.fill_(1.0)             # doubling! squares the slot's expk in place (1 stays 1)
                doubling_(pol, self._th_ones, ly["nd"], added)
            elif keep_state:
                jm = added.j0_m.clone()              # zero_added_noscat! zeroes the SOLAR j0-, never a per-source slot
                zero_added_noscat_(added, props.tau, self.dq)
                added.j0_m.copy_(jm)
            else:
                zero_added_noscat_(added, props.tau, self.dq)
                added.j0_p.zero_()
                added.j0_m.zero_()
            if iz == 0:
                copy_added_to_composite_(comp, added)
            else:
                interaction_(ly["iface"], comp, added, oplevel=True, work=self.work)
        create_surface_layer_(model.surface, self._th_surf, m, self.dq, mom["tau_sum_surface"], rho=mom["rho"])
        self._th_surf.j0_p.zero_()      # no solar beam in this slot (surface emission is a separate source type)
        self._th_surf.j0_m.zero_()
        interaction_(mom["iface_surface"], comp, self._th_surf, oplevel=True, work=self.work)
        postprocessing_vza_(pol, comp, model.vza, model.vaz, self.qp, m, weight, self.R_SFI, self.T_SFI)

    def run_graph(self):
        """`run()` replayed from a HIP graph: the launch sequence of a scene (per moment: one launch per layer, surface,
        interaction, post-processing) is captured once and replayed -- for launch-bound scenes (few spectral points,
        small N) the replay removes the per-launch host cost; the results are the same device tensors as `run()`."""
        if getattr(self, "_graph", None) is None:
            # every buffer a captured launch points at must belong to the scene: the interaction work tensor does
            # (self.work); the library's grow-only scratch (Z materialisation of component-mixed layers outside the strip
            # kernels, the non-"11" operator chains) does not, so such scenes are refused
            if self.added.d_symmetric == 0 or any(ly["props"].fcomp is not None for ly in self.moments[0]["layers"]):
                raise _lib.VSMError("run_graph(): only scenes that run entirely in the fused layer kernels (every layer "
                                    "scattering, N within the on-chip limit, one scatterer per layer) can be captured")
            # warm-up outside the capture, on the very stream the capture uses: the library's lazy initialisation, and its
            # scratch, which is keyed by (device, stream) and may not grow under a capture
            gs = torch.cuda.Stream()
            gs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(gs):
                self.run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=gs):
                self.run()
            self._graph, self._graph_stream = g, gs
        self._graph.replay()
        return self.R_SFI, self.T_SFI

    def results_host(self):
        """(R_SFI, T_SFI) as the reference returns them: [nVZA, nStokes, nSpec] numpy arrays."""
        return to_host(self.R_SFI).transpose(2, 1, 0).copy(), to_host(self.T_SFI).transpose(2, 1, 0).copy()

    def results_host_full(self):
        """The reference's SFI return tuple (rt_run.jl:535): (R_SFI, T_SFI, ieR_SFI, ieT_SFI, hdr, bhr_uw[1,:], bhr_dw[1,:]);
        the inelastic slots are zero for noRS, like the reference's freshly allocated arrays."""
        R, T = self.results_host()
        return (R, T, np.zeros_like(R), np.zeros_like(T), to_host(self.hdr).transpose(2, 1, 0).copy(),
                to_host(self.bhr_uw)[:, 0].copy(), to_host(self.bhr_dw)[:, 0].copy())

    def flops_per_point(self) -> float:
        """ALGORITHMIC flops per spectral point: per moment
        sum_l nd_l (12N^3+8N^2) + [interactions incl. surface] (24N^3+8N^2)."""
        N = float(self.N)
        tot = 0.0
        for mom in self.moments:
            for iz, ly in enumerate(mom["layers"]):
                tot += ly["nd"] * (12 * N ** 3 + 8 * N ** 2)
                if iz > 0:
                    tot += 24 * N ** 3 + 8 * N ** 2
            tot += 24 * N ** 3 + 8 * N ** 2  # surface interaction
        return tot


def prepare_scene(model: H.RTModel, spec_slice: Optional[slice] = None) -> Scene:
    return Scene(model, spec_slice)


@dataclass
class StreamRTResult:
    """StreamRTResult of the reference (rt_run.jl:107-123): per-Fourier-moment operators and SFI vectors at ALL quadrature
    streams instead of post-processed (vza, vaz) outputs; arrays in the reference's layout."""
    qp_mu: np.ndarray          # quadrature nodes (one per stream, not repeated per Stokes component)
    i_mu0: int                 # 1-based index of the SZA stream, like the reference's
    mu0: float
    pol_n: int
    weight: List[float]        # Fourier weight per moment (0.5/pi for m = 0, 1/pi else)
    R_mp_per_m: List[np.ndarray]   # [N, N, nSpec] composite R-+ at TOA
    T_pp_per_m: List[np.ndarray]   # [N, N, nSpec] composite T++ at BOA
    J_m_per_m: List[np.ndarray]    # [N, 1, nSpec] upwelling SFI vector, per-source slots added
    J_p_per_m: List[np.ndarray]    # [N, 1, nSpec] downwelling
    tau_total: np.ndarray          # [nSpec, nLayer] = tau_rayl + tau_abs (the reference leaves the aerosols out)
    tau_rayl: np.ndarray
    tau_abs: np.ndarray


def rt_run_streams(model: H.RTModel) -> StreamRTResult:
    """rt_run_streams(model) (rt_run.jl:125-192): one run, everything a caller needs to do the Fourier sum and the stream
    lookup of postprocessing_vza! offline for any number of viewing geometries (test/test_CoreRT.jl:45-108)."""
    scene = prepare_scene(model)
    scene.compute_hdrf = False
    st: list = []
    scene.run(streams=st)
    synchronize_if_gpu()
    st.sort(key=lambda d: d["m"])
    mat = lambda t: from_device_matrix(t).transpose(1, 2, 0).copy()          # (S, j, i) layout tensor -> [i, j, S]
    vec = lambda t: to_host(t).T[:, None, :].copy()                          # (S, N) -> [N, 1, S]
    qp = model.quad_points
    FT = model.float_type
    return StreamRTResult(np.asarray(qp.qp_mu, dtype=FT).copy(), int(qp.imu0) + 1, float(qp.mu0), model.polarization_type.n,
                          [d["weight"] for d in st], [mat(d["R_mp"]) for d in st], [mat(d["T_pp"]) for d in st],
                          [vec(d["J0_m"]) for d in st], [vec(d["J0_p"]) for d in st],
                          (np.asarray(model.tau_rayl) + np.asarray(model.tau_abs)).astype(FT), np.asarray(model.tau_rayl, dtype=FT),
                          np.asarray(model.tau_abs, dtype=FT))


def rt_run(model: H.RTModel, trace: Optional[list] = None, full_output: bool = False):
    """rt_run(model) (rt_run.jl:53-58 -> :238-539): returns (R_SFI, T_SFI) as host arrays [nVZA, nStokes, nSpec] -- the
    reference's first two return values -- or, with `full_output`, its whole SFI tuple
    (R_SFI, T_SFI, ieR_SFI, ieT_SFI, hdr, bhr_uw[1,:], bhr_dw[1,:]) (rt_run.jl:535)."""
    scene = prepare_scene(model)
    scene.compute_hdrf = bool(full_output)
    scene.run(trace)
    synchronize_if_gpu()
    _lib.check_device_status("rt_run")
    return scene.results_host_full() if full_output else scene.results_host()
