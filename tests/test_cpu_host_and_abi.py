"""CPU-side tests (`-m "not gpu"`): the C-ABI library loads and exports every symbol the header
declares, the host-side model code agrees with the oracle, and the spectral sharding + gather logic
works across 2 processes (gloo).  No compute kernel is called here (there is no GPU)."""
import os
import re
import socket
import sys

import numpy as np
import pytest

from oracle import vsm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vsm():
    import vsmartmom_jl_amd as v
    return v


def test_library_exports_every_declared_symbol(vsm):
    header = open(os.path.join(ROOT, "include", "vsmartmom_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vsm_[A-Za-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    bound = set(vsm._lib.exported_symbols())
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    lib = vsm._lib.lib()  # raises if libvsmartmom_hip.so was not built
    for s in sorted(declared):
        assert hasattr(lib, s), s
    assert lib.vsm_version() >= 100
    assert lib.vsm_fused_max_n(8) == 64 and lib.vsm_fused_max_n(4) == 96
    assert lib.vsm_doubling_work_elems(60, 10) == 3 * 3600 * 10 + 4 * 60 * 10
    assert isinstance(lib.vsm_last_error(), bytes)


def test_no_cpu_fallback(vsm):
    """The product path must fail loudly without a GPU -- never route through a CPU implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    H = vsm.host_model
    m = H.model_from_arrays(vsm.Architectures.GPU(), "I", 5, 30.0, [10.0], [0.0], tau_rayl=[[0.1]])
    with pytest.raises(vsm.VSMError):
        vsm.CoreRT.rt_run(m)
    m_cpu = H.model_from_arrays(vsm.Architectures.CPU(), "I", 5, 30.0, [10.0], [0.0], tau_rayl=[[0.1]])
    with pytest.raises(vsm.VSMError):
        vsm.CoreRT.rt_run(m_cpu)
    assert "oracle" not in " ".join(sys.modules[k].__name__ for k in list(sys.modules) if k.startswith("vsmartmom_jl_amd"))


def test_architectures_surface(vsm):
    A = vsm.Architectures
    assert repr(A.CPU()) == "Architectures.CPU()" and isinstance(A.default_architecture(), (A.CPU, A.GPU))
    conv = A.array_type(A.CPU())
    t = conv(np.arange(6.0).reshape(2, 3))
    assert isinstance(A.architecture(t), A.CPU) and np.array_equal(A.to_host(t), np.arange(6.0).reshape(2, 3))
    assert A.ARCH_MAP["Architectures.GPU()"] is A.GPU


@pytest.mark.parametrize("pol", ["I", "IQ", "IQU", "IQUV"])
def test_host_streams_and_Z_moments_match_oracle(vsm, pol, golden_dir):
    import json
    H = vsm.host_model
    fx = json.load(open(os.path.join(golden_dir, "solar_tester_vector.json")))
    g, go = H.GreekCoefs.from_dict(fx["greek"]), O.greek_from_dict(fx["greek"])
    for FT in (np.float64, np.float32):
        qp = H.rt_set_streams(15, 35.0, [10.0, 20.0, 40.0, 35.0], H.polarization_type(pol), FT)
        qo = O.rt_set_streams_gausslegquad(15, 35.0, [10.0, 20.0, 40.0, 35.0], O.polarization(pol), FT)
        assert np.array_equal(qp.qp_mu, qo.qp_mu) and np.array_equal(qp.wt_muN, qo.wt_muN)
        assert (qp.imu0, qp.Nquad, qp.Nstreams) == (qo.imu0, qo.Nquad, qo.Nstreams) and qp.Nquad == 8 + 3 + 1
    for m in (0, 1, 2, 5, 15):
        a = H.compute_Z_moments(H.polarization_type(pol), qp.qp_mu, g, m)
        b = O.compute_Z_moments(O.polarization(pol), qo.qp_mu, go, m)
        assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-12) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-12)
    r = H.get_greek_rayleigh(0.0279)
    ro = O.get_greek_rayleigh(0.0279)
    assert np.array_equal(r.beta, ro.beta) and np.array_equal(r.delta, ro.delta)


def test_host_layer_optics_and_ndoubl_match_oracle(vsm):
    H = vsm.host_model
    rng = np.random.default_rng(4)
    S, L = 7, 5
    tau_rayl = 0.01 + 0.05 * rng.random((S, L))
    tau_abs = 10.0 ** rng.uniform(-4, 1, (S, L))
    tau_aer = np.array([[0.0, 0.0, 0.02, 0.05, 0.1]])
    g = H.henyey_greenstein_greek(0.7, 9)
    pm = H.model_from_arrays(vsm.Architectures.CPU(), "IQU", 9, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                             tau_aer=tau_aer, aerosol_optics=[H.AerosolOptics(g, 0.95, 0.1)], depol=0.0279, m_max=3)
    om = O.build_model("IQU", 9, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, tau_aer=tau_aer,
                       aerosols=[O.AerosolOptics(O.hg_greek(0.7, 9), 0.95, 0.1)], depol=0.0279, m_max=3)
    for m in (0, 2):
        lp, lo = H.constructCoreOpticalProperties(pm, m), O.construct_core_optical_properties(om, m)
        tp, sp = H.extractEffectiveProps(lp, np.float64)
        to, so = O.extract_effective_props(lo, np.float64)
        assert tp == to and np.allclose(sp, so, rtol=1e-15)
        for a, b in zip(lp, lo):
            assert np.allclose(a.tau, b.tau, rtol=1e-15) and np.allclose(a.varpi, b.varpi, rtol=1e-15)
            assert np.allclose(a.Zpp, b.Zpp, rtol=1e-12, atol=1e-13) and a.Zpp.shape == b.Zpp.shape
            for FT in (np.float64, np.float32):
                da, na = H.get_dtau_ndoubl(np.atleast_1d(a.tau), np.broadcast_to(a.varpi, np.atleast_1d(a.tau).shape),
                                           pm.quad_points, FT, pm.numerics)
                db, nb = O.get_dtau_ndoubl(np.atleast_1d(b.tau), np.broadcast_to(b.varpi, np.atleast_1d(b.tau).shape),
                                           om.quad_points, FT)
                assert na == nb and np.array_equal(da, db)
    assert [H.get_scattering_interface(p, s, i) for p, s, i in
            (("00", True, 1), ("00", False, 1), ("00", True, 2), ("00", False, 2), ("01", True, 3), ("11", False, 3))] == \
        ["11", "00", "01", "00", "11", "10"]


def test_shard_bounds_cover_axis(vsm):
    P = vsm.parallel
    for S in (1, 7, 10, 10000, 100000):
        for W in (1, 2, 3, 8):
            b = [P.shard_bounds(S, r, W) for r in range(W)]
            assert b[0][0] == 0 and b[-1][1] == S
            assert all(b[i][1] == b[i + 1][0] for i in range(W - 1)) and all(lo <= hi for lo, hi in b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    """One rank of the 2-process gloo test.  The HIP engine is replaced by the oracle (allowed in tests/
    only) so that the sharding + global-ndoubl + gather logic is exercised end to end on CPU."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import vsmartmom_jl_amd as v
    from oracle import vsm_oracle as Oo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, L = 9, 3
    rng = np.random.default_rng(0)
    tau_rayl = np.tile(0.02 * np.ones(L), (S, 1))
    tau_rayl[S - 1] *= 40.0        # one point (owned by the LAST rank) dominates max(tau*varpi) -> global ndoubl
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.2, m_max=2)
    model = v.host_model.model_from_arrays(v.Architectures.CPU(), "IQU", 9, 40.0, [30.0, 10.0], [0.0, 90.0], **kw)

    def oracle_executor(mdl, sl):
        # same contract as Scene: ndoubl / interface tags from the FULL axis, compute on the shard
        om = Oo.build_model("IQU", 9, 40.0, [30.0, 10.0], [0.0, 90.0], **kw)
        full_R, full_T = Oo.rt_run(om)          # oracle computes with batch-global ndoubl by construction
        R = torch.from_numpy(np.ascontiguousarray(full_R[:, :, sl].transpose(2, 1, 0)))
        T = torch.from_numpy(np.ascontiguousarray(full_T[:, :, sl].transpose(2, 1, 0)))
        return R, T

    R, T = v.parallel.rt_run_sharded(model, executor=oracle_executor, rank=rank, world=world)
    if rank == 0:
        om = Oo.build_model("IQU", 9, 40.0, [30.0, 10.0], [0.0, 90.0], **kw)
        Ro, To = Oo.rt_run(om)
        q.put((bool(np.array_equal(R, Ro)), bool(np.array_equal(T, To)), R.shape))
    else:
        assert R is None and T is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == (True, True, (2, 3, 9))


def _worker_lin(rank, world, port, q):
    """One rank of the gloo test of the sharded LINEARIZED run: rt_run_lin_sharded gathers R, T, Rdot, Tdot in one collective.
    The HIP engine is replaced by the oracle (Cox-Munk ocean, gas + wind-speed slots) evaluated on the full axis and cut to
    the rank's block -- the contract SceneLin implements (ndoubl / tags from the full spectral axis)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import vsmartmom_jl_amd as v
    from oracle import vsm_oracle as Oo
    from oracle import vsm_oracle_lin as OLl
    from oracle import vsm_oracle_coxmunk as CMm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, L = 5, 2                      # 5 points over 2 ranks: blocks of 3 and 2 (padding in the gather)
    rng = np.random.default_rng(1)
    tau_rayl = np.tile(0.03 * np.ones(L), (S, 1))
    tau_abs = 10.0 ** rng.uniform(-3, -1, (S, L))
    g = tau_abs * rng.uniform(0.5, 1.5, (S, L))
    geo = ("IQU", 5, 30.0, [20.0, 50.0], [0.0, 180.0])
    surf = v.host_model.CoxMunkSurface(wind_speed=6.0)
    model = v.host_model.model_from_arrays(v.Architectures.CPU(), *geo, tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03,
                                           m_max=3, surface=surf)
    om = Oo.build_model(*geo, tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.03, m_max=3)
    full = CMm.rt_run_lin(om, OLl.LinModel([g]), CMm.CoxMunkSurface(6.0))

    def oracle_executor(mdl, lin, sl):
        R, T, Rd, Td = full
        t3 = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, :, sl].transpose(2, 1, 0)))
        t4 = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, :, sl, :].transpose(3, 2, 1, 0)))
        return t3(R), t3(T), t4(Rd), t4(Td)

    out = v.parallel.rt_run_lin_sharded(model, v.host_model.LinModel([g]), 0, 1, 1, executor=oracle_executor, rank=rank, world=world)
    if rank == 0:
        q.put(tuple(bool(np.array_equal(a, b)) for a, b in zip(out, full)) + (out[2].shape,))
    else:
        assert all(x is None for x in out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_linearized_shard_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_lin, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == (True, True, True, True, (2, 3, 5, 2))


def _worker_bench_step(rank, world, port, q, S=7):
    """One rank of the gloo test that drives bench.py's OWN timed step (bench.make_step: upload -> prepare -> run -> pack R|T ->
    ONE gather -> D2H) with a CPU stand-in for the HIP Scene: the oracle evaluated on the full axis and cut to the rank's block
    (oracle use is confined to tests/).  What the driver's multi-GPU bench executes around the kernels is exercised here."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import vsmartmom_jl_amd as v
    import bench
    from oracle import vsm_oracle as Oo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = 3                            # S = 7 over 2 ranks: blocks of 4 and 3; over 3 ranks: 3, 3, 1; S = 4 over 3 ranks: 2, 2, 0
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
    geo = ("IQU", 9, 40.0, [30.0], [0.0])
    om = Oo.build_model(*geo, tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=2)
    Ro, To = Oo.rt_run(om)           # [nVZA, nStokes, S]
    sl = v.parallel.shard_slice(S, rank, world)

    class OracleScene:               # the Scene surface make_step uses
        calls = []

        def upload(self):
            self.calls.append("upload")

        def prepare(self):
            self.calls.append("prepare")

        def run(self):
            self.calls.append("run")
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, :, sl].transpose(2, 1, 0)))
            return t(Ro), t(To)

    phase = {"h2d": 0.0, "optics": 0.0, "run": 0.0, "gather_d2h": 0.0}
    scene = OracleScene()
    step = bench.make_step(scene, v.parallel, S, rank, world, phase)
    out = step()
    out2 = step(split=True)
    if rank == 0:
        n, nV = Ro.shape[1], Ro.shape[0]
        g = out.numpy()
        Rg = g[:, :n * nV].reshape(S, n, nV).transpose(2, 1, 0)
        Tg = g[:, n * nV:].reshape(S, n, nV).transpose(2, 1, 0)
        q.put((bool(np.array_equal(Rg, Ro)), bool(np.array_equal(Tg, To)), bool(torch.equal(out, out2)), tuple(out.shape),
               scene.calls[:3], all(vv >= 0.0 for vv in phase.values())))
    else:
        assert out is None and out2 is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bench_step():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == (True, True, True, (7, 6), ["upload", "prepare", "run"], True)


@pytest.mark.parametrize("S", [7, 4])
def test_three_rank_gloo_bench_step_ragged_blocks(S):
    """bench.make_step at world size 3 with a spectral axis the ranks do not divide: blocks of 3, 3, 1 (S = 7) and 2, 2, 0
    (S = 4: the last rank owns nothing and takes no part in the gather) -- the gather moves exactly each rank's block."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bench_step, args=(r, 3, port, q, S)) for r in range(3)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == (True, True, True, (S, 6), ["upload", "prepare", "run"], True)


def test_scene_uses_global_ndoubl_for_shards(vsm):
    """`ndoubl` of a shard must come from the full spectral axis (rt_kernel.jl:197,282-283 are batch-global)."""
    H = vsm.host_model
    S, L = 8, 2
    tau_rayl = np.tile(0.01 * np.ones(L), (S, 1))
    tau_rayl[-1] *= 64.0
    m = H.model_from_arrays(vsm.Architectures.CPU(), "I", 9, 40.0, [30.0], [0.0], tau_rayl=tau_rayl)
    lods = H.constructCoreOpticalProperties(m, 0)
    full = H.get_dtau_ndoubl(lods[0].tau, np.broadcast_to(lods[0].varpi, lods[0].tau.shape), m.quad_points, np.float64, m.numerics)[1]
    first_half_alone = H.get_dtau_ndoubl(lods[0].tau[:4], np.broadcast_to(np.asarray(lods[0].varpi)[...,None] if np.ndim(lods[0].varpi)==0 else lods[0].varpi[:4], (4,)), m.quad_points, np.float64, m.numerics)[1]
    assert full == first_half_alone + 6
    src = open(os.path.join(ROOT, "vsmartmom.jl_amd", "core_rt.py")).read()
    # Scene runs the optics pass (and the per-layer maxima that decide ndoubl) over the FULL spectral axis, then slices
    assert '_lib.call("vsm_layer_optics", dt, S_full, L' in src and "self.tau[iz, lo:hi]" in src


def test_raman_halo_slices(vsm):
    """Raman sharding host logic: the halo-extended slice holds every donor of every owned recipient."""
    hs = vsm.parallel.raman_halo_slices
    assert hs(100, slice(0, 50), [-3, 2, 7]) == (slice(0, 57), slice(0, 50))
    assert hs(100, slice(50, 100), [-3, 2, 7]) == (slice(43, 100), slice(7, 57))
    assert hs(100, slice(40, 60), [500, -1]) == (slice(39, 61), slice(1, 21))     # |shift| >= S never couples
    assert hs(10, slice(10, 10), [1]) == (slice(10, 10), slice(0, 0))             # empty shard
    S, shifts = 37, [-5, 4, 11]
    for world in (2, 3, 8):
        for rank in range(world):
            own = vsm.parallel.shard_slice(S, rank, world)
            ext, crop = hs(S, own, shifts)
            assert (ext.start + crop.start, ext.start + crop.stop) == (own.start, own.stop) or own.stop <= own.start
            for n1 in range(own.start, own.stop):
                for s in shifts:
                    if 0 <= n1 + s < S:
                        assert ext.start <= n1 + s < ext.stop


def _worker_raman(rank, world, port, q):
    """2-rank gloo run of the Raman sharding: the HIP engine is replaced by the oracle on the halo-extended slice."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import vsmartmom_jl_amd as v
    from oracle import vsm_oracle as Oo
    from oracle import vsm_oracle_raman as ORr
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, L = 11, 2
    rng = np.random.default_rng(2)
    tau_rayl = np.tile(0.03 * np.ones(L), (S, 1))         # tau*varpi is spectrally flat -> ndoubl is shard-independent
    tau_abs = 10.0 ** rng.uniform(-3, 0, (S, L))
    shifts, w_ie = np.array([-2, 1, 3]), np.array([0.01, 0.02, 0.005])
    geo = ("IQU", 7, 40.0, [30.0], [0.0])
    kw = dict(tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.01, albedo=0.2, m_max=1)
    model = v.host_model.model_from_arrays(v.Architectures.CPU(), *geo, **kw)
    greek = Oo.get_greek_rayleigh(0.2)
    prs = v.CoreRTRaman.RRS(shifts, w_ie, None)

    def run_oracle(sl):
        kws = dict(kw, tau_rayl=tau_rayl[sl], tau_abs=tau_abs[sl])
        return ORr.rt_run_rrs(Oo.build_model(*geo, **kws), ORr.RRS(i_shift=shifts, varpi_ie=w_ie, greek_raman=greek))

    def oracle_executor(rs, mdl, sl):
        ext, crop = v.parallel.raman_halo_slices(S, sl, rs.i_lambda1lambda0)
        return tuple(torch.from_numpy(np.ascontiguousarray(a[:, :, crop].transpose(2, 1, 0))) for a in run_oracle(ext))

    res = v.CoreRTRaman.rt_run_sharded(prs, model, rank=rank, world=world, executor=oracle_executor)
    if rank == 0:
        full = run_oracle(slice(0, S))
        q.put(tuple(float(np.max(np.abs(a - b)) / np.max(np.abs(b))) for a, b in zip(res, full)))
    else:
        assert all(r is None for r in res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_raman_halo_shard_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_raman, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert max(res) <= 1e-13, res


def test_yaml_front_end_parses_the_rayleigh_lambertian_schema():
    """io_yaml: the `nstreams` schema of src/IO/Parameters.jl:1102-1175 and the refusal of blocks outside this backend."""
    import vsmartmom_jl_amd as V
    io = V.io_yaml
    assert np.allclose(io.parse_spec_band("[19417.0 19418.0]"), [19417.0, 19418.0])
    b = io.parse_spec_band("(1e7/765):0.5:(1e7/762)")
    assert len(b) == 103 and abs(b[0] - 1e7 / 765) < 1e-9 and abs(b[1] - b[0] - 0.5) < 1e-12
    text = """
radiative_transfer:
  spec_bands: ["[12987.0]"]
  surface: [LambertianSurfaceScalar(0.15)]
  nstreams: 3
  polarization_type: Stokes_I()
  depol: -1
  float_type: Float64
geometry: {sza: 60.0, vza: [60.0], vaz: [180.0], obs_alt: 1000.0}
atmospheric_profile: {T: [250.0, 275.0], p: [100.0, 500.0, 1000.0]}
"""
    p = io.parameters_from_yaml(text)
    assert (p.l_trunc, p.max_m, p.albedo, p.q) == (5, 6, [0.15], [0.0, 0.0])
    m = io.model_from_parameters(p, None)
    assert m.tau_rayl.shape == (1, 2) and abs(m.tau_rayl[0, 1] / m.tau_rayl[0, 0] - 500.0 / 400.0) < 1e-12
    assert m.quad_points.Nquad == 3 and m.quad_points.Nstreams == 3      # cos 60 deg IS the middle GL-3 node
    assert 0.0244 < m.tau_rayl.sum() < 0.0245   # Bodhaine 1999 at 770 nm x 1000/1013.25 hPa
    pc = io.parameters_from_yaml(text.replace("LambertianSurfaceScalar(0.15)", '"CoxMunkSurface(wind_speed=5.0, shadowing=false)"'))
    mc = io.model_from_parameters(pc, None)
    assert mc.surface == V.host_model.CoxMunkSurface(5.0, None, 0.22, True, False) and mc.m_max == 5   # user_l_cap = 2*3 - 1
    assert io.parse_surface("CoxMunkSurface(3.5)").wind_speed == 3.5
    assert io.parse_surface("CoxMunkSurface(wind_speed=4, n_water=1.34+0.01im)").n_water == complex(1.34, 0.01)
    prpv = io.parameters_from_yaml(text.replace("LambertianSurfaceScalar(0.15)", '"rpvSurfaceScalar(0.1, 0.2, 0.3, 0.4)"'))
    assert isinstance(io.model_from_parameters(prpv, None).surface, V.host_model.rpvSurfaceScalar)
    with pytest.raises(NotImplementedError):   # (the canopy surface is outside SURVEY 8)
        io.parameters_from_yaml(text.replace("LambertianSurfaceScalar(0.15)", '"CanopySurface(LAI=3.0)"'))
    with pytest.raises(ValueError):
        io.parse_surface("CoxMunkSurface(n_water=1.33)")
    with pytest.raises(NotImplementedError):
        io.parameters_from_yaml(text + "absorption: {molecules: [[O2]]}\n")
    with pytest.raises(ValueError):
        io.parameters_from_yaml(text.replace("nstreams: 3", "nstreams: 2"))


def test_host_aerosol_jacobian_optics_equal_the_oracle_chain():
    """host_model.constructCoreOpticalPropertiesLin with aerosol slots (closed-form derivatives of the mixed layer) against the
    oracle's restatement of the reference's pairwise `+` chain (types_lin.jl:196-380, compEffectiveLayerProperties_lin.jl:330-395)."""
    import vsmartmom_jl_amd as vsm
    from oracle import vsm_oracle as O, vsm_oracle_lin as OL
    from tests.test_oracle_lin import _aerosol_scene
    H = vsm.host_model
    om, ol = _aerosol_scene("IQU")
    ao, lao = om.aerosol_optics[0], ol.lin_aerosol_optics[0]
    pm = H.model_from_arrays(vsm.Architectures.CPU(), "IQU", 9, 35.0, [20.0, 50.0], [0.0, 120.0], tau_rayl=om.tau_rayl,
                             tau_abs=om.tau_abs, tau_aer=om.tau_aer,
                             aerosol_optics=[H.AerosolOptics(H.GreekCoefs(**vars(ao.greek)), ao.ssa, ao.f_trunc)], depol=0.03,
                             albedo=om.albedo, m_max=om.m_max)
    pl = H.LinModel(ol.tau_abs_dot, tau_aer_dot=ol.tau_aer_dot,
                    lin_aerosol_optics=[H.LinAerosolOptics([H.GreekCoefs(**vars(g)) for g in lao.greek_dot], lao.ssa_dot, lao.f_trunc_dot)])
    for m in (0, 1, 3):
        a = H.constructCoreOpticalPropertiesLin(pm, pl, H.constructCoreOpticalProperties(pm, m), m)
        b = OL.layer_optics_lin(om, ol, O.construct_core_optical_properties(om, m), m)
        for x, y in zip(a, b):
            for u, v in ((x.tau_dot, y.tau_dot), (x.varpi_dot, y.varpi_dot), (x.Zpp_dot, y.Zpp_dot), (x.Zmp_dot, y.Zmp_dot)):
                assert np.max(np.abs(u - v)) <= 1e-12 * max(np.max(np.abs(v)), 1e-30), m


def test_land_brdf_surfaces_host_equals_oracle_and_yaml_parses():
    """host_model.brdf_reflectance (rpvSurfaceScalar / RossLiSurfaceScalar Fourier blocks) against the oracle's restatement, and
    the surface constructors of config/vegetation_rpv.yaml / vegetation_rossli.yaml through parse_surface."""
    import vsmartmom_jl_amd as vsm
    from oracle import vsm_oracle_brdf as OB
    H, io = vsm.host_model, vsm.io_yaml
    s1, s2 = io.parse_surface("rpvSurfaceScalar(0.12, 0.08, 0.75, -0.25)"), io.parse_surface("RossLiSurfaceScalar(0.05, 0.03, 0.10)")
    assert (s1.rho0, s1.rho_c, s1.k, s1.Theta) == (0.12, 0.08, 0.75, -0.25) and (s2.fvol, s2.fgeo, s2.fiso) == (0.05, 0.03, 0.10)
    mu = np.array([0.05, 0.3, 0.5, 0.7660444431189781, 0.95, 1.0])
    for hs, osf in ((s1, OB.RPVSurface(0.12, 0.08, 0.75, -0.25)), (s2, OB.RossLiSurface(0.05, 0.03, 0.10))):
        for m in (0, 1, 4, 9):
            for ns in (1, 3):
                a, b = H.brdf_reflectance(hs, ns, mu, m), OB.reflectance(osf, ns, mu, m)
                assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b)), (type(hs).__name__, m, ns)
    with pytest.raises(NotImplementedError):
        io.parse_surface("CanopySurface(1.0)")
