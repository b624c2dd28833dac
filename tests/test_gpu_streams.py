"""Stream contract of the C ABI (include/vsmartmom_hip.h, Conventions): calls on different HIP streams are independent.
Library-owned scratch is keyed by (device, stream), so two scenes that run concurrently on two streams -- or the Fourier-moment
lanes of the linearized run -- never share the parked strips of k_dbl128 / k_ia128 (64 < N <= 128) or the pre-pass images of the
layer kernels.  Each test compares the concurrent run bit for bit with the sequential one and with the oracle (1e-8)."""
import numpy as np
import pytest

from oracle import vsm_oracle as O
from oracle import vsm_oracle_lin as OL

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


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
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _atmosphere(seed, S, L):
    rng = np.random.default_rng(seed)
    tau_rayl = np.tile(0.05 * np.ones(L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, 0.3, (S, L))
    return dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, m_max=2)


# N = 72 (IQU, 24 streams), 112 (IQUV, 28 streams): the k_dbl128 / k_ia128 family with its parked strips; N = 57: pre-pass images
@pytest.mark.parametrize("pol,l_trunc,N", [("IQU", 41, 72), ("IQUV", 49, 112), ("IQU", 31, 57)])
def test_two_forward_scenes_on_two_streams(vsm, arch, pol, l_trunc, N):
    H = vsm.host_model
    S, L = 40, 3
    geo = (pol, l_trunc, 40.0, [30.0, 5.0], [0.0, 60.0])
    kws = [_atmosphere(11, S, L), _atmosphere(12, S, L)]
    albedos = [0.1, 0.4]
    models = [H.model_from_arrays(arch, *geo, albedo=a, **kw) for a, kw in zip(albedos, kws)]
    scenes = [vsm.CoreRT.Scene(mo) for mo in models]
    assert scenes[0].N == N
    # sequential reference on the current stream
    seq = []
    for sc in scenes:
        R, T = sc.run()
        torch.cuda.synchronize()
        seq.append((R.clone(), T.clone()))
    # grow the scratch of one stream between runs of the other (a grow must not free under running work)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        for sc, st in zip(scenes, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                sc.run()
        if rep == 0:   # a third, larger scene on stream 0 while stream 1 is still busy: its scratch grows
            big = vsm.CoreRT.Scene(H.model_from_arrays(arch, *geo, albedo=0.2, **_atmosphere(13, 3 * S, L)))
            with torch.cuda.stream(streams[0]):
                big.run()
                scenes[0].run()
        torch.cuda.synchronize()
        for sc, (R0, T0) in zip(scenes, seq):
            assert torch.equal(sc.R_SFI, R0) and torch.equal(sc.T_SFI, T0), rep
    for sc, a, kw in zip(scenes, albedos, kws):
        Ro, To = O.rt_run(O.build_model(*geo, albedo=a, **kw))
        R, T = sc.results_host()
        assert _rel(R, Ro) < 1e-8 and _rel(T, To) < 1e-8
    assert vsm._lib.lib().vsm_release_scratch() == 0
    R, T = scenes[0].run()     # the scratch comes back after a release
    torch.cuda.synchronize()
    assert torch.equal(R, seq[0][0]) and torch.equal(T, seq[0][1])


@pytest.mark.parametrize("pol,l_trunc,N", [("IQU", 41, 72), ("IQUV", 49, 112)])
def test_lin_moment_lanes_strip128_shapes(vsm, arch, pol, l_trunc, N):
    """SceneLin on concurrent moment lanes (no-fold and fold) at the shapes whose forward kernels park strips in library scratch:
    bit-equal between repeated runs, equal to the sequential walk up to the reordering of the sum over moments, 1e-8 from the
    oracle."""
    rng = np.random.default_rng(5)
    S, L = 3, 3
    H = vsm.host_model
    ga = 10.0 ** rng.uniform(-2.5, -0.5, (S, L))
    kw = dict(tau_rayl=np.tile(0.03 * np.ones(L), (S, 1)), tau_abs=ga, depol=0.0279, m_max=3)
    geo = (pol, l_trunc, 40.0, [30.0, 5.0], [0.0, 60.0])
    pm = H.model_from_arrays(arch, *geo, albedo=0.2, **kw)
    scene = vsm.CoreRTLin.SceneLin(pm, H.LinModel([ga]), 0, 1, 1)
    assert scene.N == N
    seq = [t.clone() for t in scene.run(lanes=1)]
    torch.cuda.synchronize()
    runs = {}
    for name, kwargs in (("lanes", dict(lanes=4, fold=False)), ("fold", dict(lanes=4, fold=True))):
        a = [t.clone() for t in scene.run(**kwargs)]
        torch.cuda.synchronize()
        b = [t.clone() for t in scene.run(**kwargs)]
        torch.cuda.synchronize()
        for x, y, z in zip(a, b, seq):
            assert torch.equal(x, y), name
            assert float((x - z).abs().max()) <= 1e-12 * float(z.abs().max()), name
        runs[name] = a
    Ro, To, Rdo, Tdo = OL.rt_run_lin(O.build_model(*geo, albedo=0.2, **kw), OL.LinModel([ga]))
    R, T, Rd, Td = scene.results_host()
    assert _rel(R, Ro) < 1e-8 and _rel(T, To) < 1e-8 and _rel(Rd, Rdo) < 1e-8 and _rel(Td, Tdo) < 1e-8


def test_forward_scene_and_lin_lanes_concurrently(vsm, arch):
    """A forward scene (N = 112, parked strips) on one stream while a linearized scene of the same shape runs its lanes."""
    rng = np.random.default_rng(6)
    H = vsm.host_model
    geo = ("IQUV", 49, 40.0, [30.0, 5.0], [0.0, 60.0])
    kwf = _atmosphere(21, 24, 3)
    fwd = vsm.CoreRT.Scene(H.model_from_arrays(arch, *geo, albedo=0.3, **kwf))
    ga = 10.0 ** rng.uniform(-2.5, -0.5, (3, 3))
    kwl = dict(tau_rayl=np.tile(0.03 * np.ones(3), (3, 1)), tau_abs=ga, depol=0.0279, m_max=3)
    lin = vsm.CoreRTLin.SceneLin(H.model_from_arrays(arch, *geo, albedo=0.2, **kwl), H.LinModel([ga]), 0, 1, 1)
    R0, T0 = [t.clone() for t in fwd.run()]
    torch.cuda.synchronize()
    l0 = [t.clone() for t in lin.run(lanes=4, fold=False)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(2):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd.run()
            fwd.run()
        l1 = lin.run(lanes=4, fold=False)
        torch.cuda.synchronize()
        assert torch.equal(fwd.R_SFI, R0) and torch.equal(fwd.T_SFI, T0)
        for x, y in zip(l0, l1):
            assert torch.equal(x, y)
