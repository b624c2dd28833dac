"""ctypes binding of libvsmartmom_hip.so (C ABI: include/vsmartmom_hip.h).

No compute happens in Python: every function here forwards device pointers to
the HIP library.  If the library is missing or a call fails this module raises
-- there is NO CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSM_LIB_PATH") or os.path.join(_HERE, "lib", "libvsmartmom_hip.so")   # (override: A/B builds)


class VSMError(RuntimeError):
    pass


class vsm_quad_f64(C.Structure):
    _fields_ = [("mu", C.c_void_p), ("wt", C.c_void_p), ("N", C.c_int), ("n_stokes", C.c_int),
                ("i_mu0", C.c_int), ("mu0", C.c_double)]


class vsm_quad_f32(C.Structure):
    _fields_ = [("mu", C.c_void_p), ("wt", C.c_void_p), ("N", C.c_int), ("n_stokes", C.c_int),
                ("i_mu0", C.c_int), ("mu0", C.c_float)]


class vsm_added(C.Structure):  # same layout for f32/f64 (pointers + stride)
    _fields_ = [("r_mp", C.c_void_p), ("t_pp", C.c_void_p), ("r_pm", C.c_void_p), ("t_mm", C.c_void_p),
                ("j0_p", C.c_void_p), ("j0_m", C.c_void_p), ("mat_stride", C.c_longlong),
                ("d_symmetric", C.c_int), ("reserved", C.c_int)]


class vsm_composite(C.Structure):
    _fields_ = [("R_mp", C.c_void_p), ("R_pm", C.c_void_p), ("T_pp", C.c_void_p), ("T_mm", C.c_void_p),
                ("J0_p", C.c_void_p), ("J0_m", C.c_void_p)]


class vsm_added_lin(C.Structure):
    _fields_ = [("ap_r_mp", C.c_void_p), ("ap_t_pp", C.c_void_p), ("ap_r_pm", C.c_void_p), ("ap_t_mm", C.c_void_p),
                ("ap_J0_p", C.c_void_p), ("ap_J0_m", C.c_void_p), ("P", C.c_int), ("reserved", C.c_int),
                ("mat_stride", C.c_longlong)]


class vsm_composite_lin(C.Structure):
    _fields_ = [("R_mp", C.c_void_p), ("R_pm", C.c_void_p), ("T_pp", C.c_void_p), ("T_mm", C.c_void_p),
                ("J0_p", C.c_void_p), ("J0_m", C.c_void_p), ("P", C.c_int), ("reserved", C.c_int)]


class vsm_added_rs(C.Structure):
    _fields_ = [("ier_mp", C.c_void_p), ("iet_pp", C.c_void_p), ("ier_pm", C.c_void_p), ("iet_mm", C.c_void_p),
                ("ieJ0_p", C.c_void_p), ("ieJ0_m", C.c_void_p), ("K", C.c_int), ("reserved", C.c_int)]


class vsm_composite_rs(C.Structure):
    _fields_ = [("ieR_mp", C.c_void_p), ("ieR_pm", C.c_void_p), ("ieT_pp", C.c_void_p), ("ieT_mm", C.c_void_p),
                ("ieJ0_p", C.c_void_p), ("ieJ0_m", C.c_void_p), ("K", C.c_int), ("reserved", C.c_int)]


class vsm_coxmunk_f64(C.Structure):
    _fields_ = [("wind_speed", C.c_double), ("n_water_re", C.c_double), ("n_water_im", C.c_double),
                ("whitecap_albedo", C.c_double), ("include_whitecaps", C.c_int), ("shadowing", C.c_int)]


class vsm_coxmunk_f32(C.Structure):
    _fields_ = [("wind_speed", C.c_float), ("n_water_re", C.c_float), ("n_water_im", C.c_float),
                ("whitecap_albedo", C.c_float), ("include_whitecaps", C.c_int), ("shadowing", C.c_int)]


class vsm_rrs(C.Structure):
    _fields_ = [("shift", C.c_void_p), ("varpi_ie", C.c_void_p), ("fscatt", C.c_void_p), ("Zpp", C.c_void_p),
                ("Zmp", C.c_void_p)]


_P, _I, _LL, _SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_size_t

# name -> (restype, argtypes); {T} expands to f64/f32, {R} to c_double/c_float
_SIGS = {
    "vsm_version": (_I, []),
    "vsm_last_error": (C.c_char_p, []),
    "vsm_build_id": (C.c_char_p, []),
    "vsm_device_count": (_I, [C.POINTER(_I)]),
    "vsm_device_name": (_I, [_I, C.c_char_p, _SZ]),
    "vsm_sync": (_I, [_P]),
    "vsm_release_scratch": (_I, []),
    "vsm_device_status": (_I, [C.POINTER(_I), _I, _P]),
    "vsm_fused_max_n": (_I, [_I]),
    "vsm_layer_thermal_fused": (_I, [_I, _I]),
    "vsm_doubling_work_elems": (_SZ, [_I, _I]),
    "vsm_interaction_work_elems": (_SZ, [_I, _I]),
    "vsm_batched_mul_{T}": (_I, [_I, _I, _I, _I, _P, _LL, _P, _LL, _P, _P]),
    "vsm_batch_inv_{T}": (_I, [_I, _I, _P, _P, _P, _P]),
    "vsm_batch_solve_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_elemental_doubling_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _LL, _P, _P]),
    "vsm_layer_forward_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _LL, _I, _P, _P, _P]),
    "vsm_layer_forward_multi_{T}": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _LL, _P, _P, _I, _P, _P, _P]),
    "vsm_layer_forward_thermal_{T}": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _LL, _P, _I, _P, _P]),
    "vsm_layer_forward_mix_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "vsm_mix_Z_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_mix_Z_moments_{T}": (_I, [_I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "vsm_elemental_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _LL, _P, _P]),
    "vsm_doubling_{T}": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "vsm_noscat_layer_{T}": (_I, [_P, _I, _P, _P, _P]),
    "vsm_thermal_source_{T}": (_I, [_P, _I, _P, _P, _P, _P, _P]),
    "vsm_copy_added_to_composite_{T}": (_I, [_I, _I, _P, _P, _P]),
    "vsm_interaction_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "vsm_interaction_oplevel_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "vsm_lambertian_surface_{T}": (_I, [_P, _I, _I, "{R}", _P, _P, _P]),
    "vsm_postprocess_vza_{T}": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_compute_Z_moments_{T}": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "vsm_layer_optics_{T}": (_I, [_I, _I, _I, _P, _P, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_layer_dtau_{T}": (_I, [_I, _I, _P, _P, _P, _P]),
    "vsm_layer_optics_lin_{T}": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, _P, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                      _P, _P, _P]),
    "vsm_layer_expk_{T}": (_I, [_I, _P, "{R}", _P, _P]),
    "vsm_coxmunk_reflectance_{T}": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "vsm_lambertian_surface_spectral_{T}": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "vsm_brdf_surface_{T}": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "vsm_brdf_surface_lin_{T}": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P]),
    "vsm_interaction_hdrf_{T}": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_postprocess_vza_hdrf_{T}": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "vsm_coxmunk_ss_correction_{T}": (_I, [_P, _I, _I, _I, _P, _P, "{R}", _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_doubling_lin_work_elems": (_SZ, [_I, _I, _I]),
    "vsm_interaction_lin_work_elems": (_SZ, [_I, _I, _I]),
    "vsm_elemental_lin_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _LL, _I, _P, _P, _P, _P, _P, _LL, _LL, _P, _P, _P]),
    "vsm_elemental_lin_fold_{T}": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _LL, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_elemental_lin_mix_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_doubling_lin_{T}": (_I, [_I, _I, _I, _I, _P, _P, "{R}", _I, _P, _P, _P, _P]),
    "vsm_interaction_lin_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_interaction_lin_range_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vsm_copy_added_to_composite_lin_{T}": (_I, [_I, _I, _P, _P, _P]),
    "vsm_lambertian_surface_lin_{T}": (_I, [_P, _I, _I, "{R}", _I, _P, _P, _I, _P, _P, _P, _P]),
    "vsm_postprocess_vza_lin_{T}": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_doubling_inelastic_work_elems": (_SZ, [_I, _I, _I]),
    "vsm_interaction_inelastic_work_elems": (_SZ, [_I, _I, _I]),
    "vsm_elemental_inelastic_rrs_{T}": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_doubling_inelastic_rrs_{T}": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsm_interaction_inelastic_rrs_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_copy_added_to_composite_ie_{T}": (_I, [_I, _I, _P, _P, _P]),
    "vsm_postprocess_vza_ie_{T}": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsm_run_supported": (_I, [_I, _I, _I]),
    "vsm_run_workspace_bytes": (_SZ, [_I, _I, _I, _I, _P]),
    "vsm_run_supported_f32": (_I, [_I, _I, _I]),
    "vsm_run_workspace_bytes_f32": (_SZ, [_I, _I, _I, _I, _P]),
    "vsm_run_create_{T}": (_I, [_P, _I, _I, _P, _P, _P, _SZ, _P]),
    "vsm_run_layer_{T}": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _P, _LL, _P, _I, _P, _P]),
    "vsm_run_export_{T}": (_I, [_P, _P, _P]),
    "vsm_run_import_{T}": (_I, [_P, _P, _P]),
    "vsm_run_destroy": (_I, [_P]),
    "vsm_stokes_coupling_{T}": (_I, [_I, _I, _I, _P, _P, _P, _P]),
    "vsm_test_lds_mm_{T}": (_I, [_I, _I, _P, _P, _P, _P]),
    "vsm_test_lds_inv_{T}": (_I, [_I, _I, _P, _P, _I, _P, _P]),
    "vsm_test_poison_lds": (_I, [_P]),
}


def exported_symbols():
    """Every symbol include/vsmartmom_hip.h declares (used by the CPU-side ABI test)."""
    out = []
    for name in _SIGS:
        if "{T}" in name:
            out += [name.format(T="f64"), name.format(T="f32")]
        else:
            out.append(name)
    return out


_lib = None


def lib():
    """Load the shared library (raises VSMError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VSMError("libvsmartmom_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        for T, R in (("f64", C.c_double), ("f32", C.c_float)):
            n = name.format(T=T)
            try:
                fn = getattr(L, n)
            except AttributeError:
                if os.environ.get("VSM_LIB_PATH"):     # an A/B build of an earlier round may lack the newest entry points
                    if "{T}" not in name:
                        break
                    continue
                raise
            fn.restype = res
            fn.argtypes = [R if a == "{R}" else a for a in args]
            if "{T}" not in name:
                break
    _lib = L
    return L


def build_info() -> dict:
    """Identity of the loaded library: `source_hash` = vsm_build_id() (SHA-256 prefix over the library's sources, compiled in),
    plus what csrc/Makefile wrote next to it (the commit checked out at build time)."""
    import json
    info = {"source_hash": lib().vsm_build_id().decode() if hasattr(lib(), "vsm_build_id") else None}
    try:
        with open(os.path.join(os.path.dirname(LIB_PATH), "BUILD_INFO.json")) as f:
            side = json.load(f)
        if side.get("source_hash") == info["source_hash"]:
            info.update(side)
    except (OSError, ValueError):
        pass
    return info


def poison(t):
    """Work buffers are handed to the library uninitialised; with VSM_POISON_WORK=1 (a test-run switch) they are filled
    with NaN first, so that a kernel reading scratch it has not written shows up as NaN instead of stale data."""
    if os.environ.get("VSM_POISON_WORK"):
        t.fill_(float("nan"))
    return t


def check(rc):
    if rc != 0:
        raise VSMError("libvsmartmom_hip: status %d: %s" % (rc, lib().vsm_last_error().decode()))


last_device_status = [0, 0, 0, 0]


def check_device_status(what="rt_run"):
    """The in-kernel inverses cannot return `info` from an asynchronous launch; they raise device flags instead
    (vsm_device_status).  Called after the synchronisation that ends a run: a singular (I - R r) raises here the way the
    reference's LU raises SingularException on the host (cpu_batched.jl:32-47)."""
    global last_device_status
    flags = (C.c_int * 4)()
    check(lib().vsm_device_status(flags, 1, None))
    last_device_status = list(flags)     # (tests read the counters: which kernel family a run landed on)
    if flags[0] & 1:
        raise VSMError("%s: singular matrix in an in-kernel inverse ((I - R r) or (I - r r) has an exactly zero pivot)" % what)
    if flags[0] & 2:
        raise VSMError("%s: NaN / Inf operand in an in-kernel inverse" % what)
    if flags[0] & 4:
        raise VSMError("%s: a phase matrix handed to vsm_run_layer has a non-zero element outside the declared Stokes coupling "
                       "(VSM_DEVSTAT_MASK): the native run dropped that coupling" % what)


def suffix(dtype) -> str:
    import torch
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise VSMError("unsupported float type %r (Float64 / Float32 only)" % (dtype,))


def call(name, dtype, *args):
    fn = getattr(lib(), "%s_%s" % (name, suffix(dtype)))
    check(fn(*args))
