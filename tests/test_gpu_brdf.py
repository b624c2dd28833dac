"""Kernel-driven land BRDF surfaces (rpvSurfaceScalar, RossLiSurfaceScalar) on the device: the shapes of the reference's
config/vegetation_rpv.yaml and config/vegetation_rossli.yaml through parameters_from_yaml -> model_from_parameters -> rt_run,
against the oracle (oracle/vsm_oracle_brdf.py, which is tied to the golden-pinned Lambertian path by exact limits)."""
import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_brdf as OB

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

VEG_YAML = """
# the shape of config/vegetation_rpv.yaml / vegetation_rossli.yaml: Stokes_I, 11 streams, 9 viewing angles in the principal plane,
# two spectral points, pure Rayleigh over a BRDF surface (a 6-layer profile instead of the 33-level one)
radiative_transfer:
  spec_bands: ["[19417.0 19418.0]"]
  surface:
    - %s
  polarization_type: Stokes_I()
  nstreams: 11
  truncation: NoTruncation()
  depol: -1
  float_type: Float64
  architecture: default_architecture
geometry: {sza: 40, vza: [60, 45, 30, 15, 0, 15, 30, 45, 60], vaz: [180, 180, 180, 180, 0, 0, 0, 0, 0], obs_alt: 1000.0}
atmospheric_profile: {T: [220.0, 230.0, 250.0, 265.0, 280.0, 287.0], p: [1.0, 50.0, 200.0, 400.0, 650.0, 850.0, 1000.0], profile_reduction: -1}
"""


@pytest.fixture(scope="module")
def vsm():
    import vsmartmom_jl_amd as v
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    v._lib.lib()
    return v


@pytest.fixture(scope="module")
def arch(vsm):
    return vsm.Architectures.GPU()


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("ctor,osurf", [("rpvSurfaceScalar(0.12, 0.08, 0.75, -0.25)", OB.RPVSurface(0.12, 0.08, 0.75, -0.25)),
                                        ("RossLiSurfaceScalar(0.05, 0.03, 0.10)", OB.RossLiSurface(0.05, 0.03, 0.10))])
def test_rt_run_vegetation_configs(vsm, arch, ctor, osurf):
    io = vsm.io_yaml
    params = io.parameters_from_yaml(VEG_YAML % ctor)
    model = io.model_from_parameters(params, arch)
    assert type(model.surface).__name__ == ctor.split("(")[0]
    assert model.m_max == min(2 * params.nstreams - 1, params.max_m - 1, params.l_trunc)     # component_m_max.jl:73-74: no cap of its own
    R, T = vsm.CoreRT.rt_run(model)
    om = O.build_model(params.polarization_type, params.l_trunc, params.sza, params.vza, params.vaz, model.tau_rayl, depol=0.0,
                       albedo=0.0, m_max=model.m_max)
    om.greek_rayleigh = O.GreekCoefs(**{k: np.asarray(getattr(model.greek_rayleigh, k)) for k in
                                        ("alpha", "beta", "gamma", "delta", "epsilon", "zeta")})
    Ro, To = OB.rt_run(om, osurf)
    assert R.shape == Ro.shape and np.max(np.abs(Ro)) > 0
    assert _rel(R, Ro) < 1e-9 and _rel(T, To) < 1e-9


def test_rpv_lambertian_limit_equals_lambertian_surface(vsm, arch):
    """rpvSurfaceScalar(rho0, 1, 1, 0) == LambertianSurfaceScalar(rho0) through the BRDF surface path of the device."""
    io = vsm.io_yaml
    Rb, Tb = vsm.CoreRT.rt_run(io.model_from_parameters(io.parameters_from_yaml(VEG_YAML % "rpvSurfaceScalar(0.2, 1.0, 1.0, 0.0)"), arch))
    Rl, Tl = vsm.CoreRT.rt_run(io.model_from_parameters(io.parameters_from_yaml(VEG_YAML % "LambertianSurfaceScalar(0.2)"), arch))
    assert _rel(Rb, Rl) < 1e-11 and _rel(Tb, Tl) < 1e-11
