"""Secondary workloads of bench.py (`config.secondary` of the JSON line): one timed step each of the other BASELINE.json
configurations that fit one GPU, so that their figures are driver-timed and not only builder-run (tools/*.py):

  C2-dropin / C2-dropin-native   the headline's C2 step in the call order of the reference's unpatched driver (m outside, one
          rt_kernel_ call per (m, iz) on ONE CompositeLayer): on the reference-layout kernels / with rt_kernel_'s per-composite
          registry keeping the composite in native layout (the path julia/vSmartMOMROCmExt.jl's rt_kernel!(::noRS) takes)
  C4      forward, N = 96 FP32, 60 layers, 12 500 points (one GPU's share of configs[3])
  C2-aer  the SURVEY 8(d) aerosol variant of C2 (HG aerosol in the lowest 6 layers, per-point Z, m = 0..35), 10 000 points
  C2-lin  rt_run(model, lin_model, 0, 1, 1) on the C2 shape (1 gas column + albedo), 2 048 points; C2-lin-10k: 10 000 points
  C3-lin  the ocean / Cox-Munk scene of config/ocean_coxmunk.yaml (IQUV, N = 60, 33 layers, 2 points, m = 0..21), linearized
  C5      rotational Raman, N = 21, 12 layers, K = 40 lines, 4 000 points (configs[4] at a fifth of its spectral axis);
          C5-10k: 10 000 points (one GPU's share of the 2-GPU configs[4])
  C1      quickstart-shaped (config/quickstart.yaml: Stokes_I, nstreams = 3, Rayleigh, Lambertian 0.15; BASELINE configs[0]:
          ~10 layers, ~100 spectral points): the WHOLE rt_run(model) call -- host model / optics, scene allocation, H2D, device
          pass, D2H (north_star: >= 10^4 spectral-points/s on quickstart-shaped atmospheres)
  IA      the interaction kernel in isolation: interaction!(::ScatteringInterface_11), N = 60 FP64, 10 240 points
          (k_ia_native<4, 15>; north_star: >= 40 % MFMA utilisation in the interaction kernel -- inside the headline it is part
          of the fused layer kernel); IA-long: the same at 4 x the reflectances (series orders >= 15: the out-of-line inverse)
  N112    forward run at the reference's VLIDORT case-A size (test/vlidort_baseline/cases/case_A_siewert2000.jl:29-50:
          IQUV, N = 112), 2 000 points, 10 layers: the k_dbl128 / k_ia128 family

Each entry: points/s of the step, its wall time, the device time of the pass by HIP events, the algorithmic TFLOP/s and the
fraction of the MFMA peak of its dtype BY THE AS-WRITTEN (dense N x N) OPERATION COUNT of SURVEY 8(d), and the dominant kernel
(per-kernel durations: profiles/r05/).  Since round 5 the runs do not form products with exact zeros (Stokes blocks the phase matrices
do not couple: DESIGN 4.0), so an as-written fraction can exceed what the MFMA pipe could deliver on the dense problem; entries whose
executed work differs say so (`frac_of_mfma_peak_executed_products`, `note`).  A step is what the
headline times: H2D of the raw inputs (`upload()`), device layer optics (`prepare()`), the device pass (`run()`) and the D2H of the
result, on a scene whose buffers are allocated beforehand, with one untimed warm-up; the C5 and C1 steps are whole `rt_run(...)`
calls (host optics, allocation, H2D, device pass, D2H).  Only bench.py imports this module, on rank 0 of a 1-GPU run.
"""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK = {"f64": 78.6, "f32": 157.3}


def _timed(torch, f):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    out = f()
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, e0.elapsed_time(e1), out


def _entry(name, workload, S, wall, dev_ms, flops_pt, dtype, kernel):
    tf = flops_pt * S / wall / 1e12 if flops_pt else None
    return {"name": name, "workload": workload, "points": S, "value": S / wall, "unit": "spectral-points/s", "ms_per_step": 1e3 * wall,
            "device_pass_ms": dev_ms, "dtype": dtype, "algorithmic_gflop_per_point": flops_pt / 1e9 if flops_pt else None,
            "achieved_tflops": tf, "frac_of_mfma_peak": tf / PEAK[dtype] if tf else None, "dominant_kernel": kernel}


def _executed(e, scene, what):
    """frac_of_mfma_peak by the products that ARE formed (blocks of Stokes components that do not couple run as independent
    sub-problems: the as-written count of SURVEY 8d stays beside it as frac_of_mfma_peak_as_written)."""
    import bench
    ratio = bench.executed_over_algorithmic(scene)
    e["frac_of_mfma_peak_as_written"] = e["frac_of_mfma_peak"]
    e["executed_over_algorithmic_flops"] = ratio
    e["frac_of_mfma_peak"] = e["frac_of_mfma_peak_as_written"] * ratio
    e["note"] = ("frac_of_mfma_peak_as_written counts every layer step as a dense N x N problem, as the reference writes it (SURVEY 8d); the run "
                 "forms the products of the Stokes blocks that couple only (%s): frac_of_mfma_peak is by those, whole step" % what)
    return e


def _lin_step_inputs(scene):
    """The input half of a linearized step, like the headline's: H2D of the raw optical depths and their derivatives, layer
    optics and their derivatives on the device (forward scene + SceneLin)."""
    scene.fwd.upload()
    scene.fwd.prepare()
    scene.upload()
    scene.prepare()


def c4(vsm, torch, arch, o2a, points=12500):
    cfg = dict(pol="IQU", l_trunc=59, L=60)
    tau_rayl, tau_abs = o2a(points, cfg["L"])
    model = vsm.host_model.model_from_arrays(arch, cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                                             depol=0.0279, albedo=0.15, m_max=2, float_type=np.float32)
    scene = vsm.CoreRT.prepare_scene(model)

    def step():
        scene.upload()
        scene.prepare()
        R, T = scene.run()
        return R.cpu(), T.cpu()
    wall, dev, _ = _timed(torch, step)
    e = _entry("C4", "N=96 FP32, 60 layers, m=0..2, Rayleigh + O2, Lambertian (one GPU's share of BASELINE configs[3])", points, wall, dev,
               scene.flops_per_point(), "f32", "k_layer_native32<6, 24> (m = 1, 2: two points per workgroup) + k_layer_native32<4, 16> (m = 0: a 64-row "
               "(I,Q) block, U as a diagonal step): FP32 records and arithmetic")
    _executed(e, scene, "m = 0: a 64-row (I,Q) block and a diagonal step for U; m = 1, 2 dense")
    del scene
    return e


def c2_dropin(vsm, torch, arch, o2a, native, points=10000):
    """The headline's C2 step issued in the call order of the reference's UNPATCHED driver (rt_run.jl:383-470: `for m` outside
    `for iz`, one rt_kernel! call per (m, iz) on ONE CompositeLayer, then create_surface_layer! / interaction! /
    postprocessing_vza!) -- what julia/vSmartMOMROCmExt.jl is reached through; CoreRT.REFERENCE_ORDER makes Scene.run issue
    exactly that sequence.  native = False: every layer step on the reference-layout composite (vsm_layer_forward: one Fourier
    moment per launch, the round-4 kernel); native = True: rt_kernel_ keeps the composite in the kernels' strip layout between its
    calls (the per-composite registry: a one-moment vsm_run created at the TOA call, exported lazily by the surface interaction)."""
    L = 40
    tau_rayl, tau_abs = o2a(points, L)
    model = vsm.host_model.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                             albedo=0.15, m_max=2)
    CR = vsm.CoreRT
    saved = (CR.REFERENCE_ORDER, CR.NATIVE_RUN, CR.NATIVE_DROPIN)
    CR.REFERENCE_ORDER, CR.NATIVE_RUN, CR.NATIVE_DROPIN = True, bool(native), bool(native)
    try:
        scene = CR.prepare_scene(model)
        calls = []
        orig = CR._rt_kernel_native
        CR._rt_kernel_native = lambda *a: (calls.append(orig(*a)), calls[-1])[1]

        def step():
            scene.upload()
            scene.prepare()
            R, T = scene.run()
            return R.cpu(), T.cpu()
        try:
            wall, dev, _ = _timed(torch, step)
        finally:
            CR._rt_kernel_native = orig
        vsm._lib.check_device_status("C2-dropin")
        e = _entry("C2-dropin-native" if native else "C2-dropin",
                   "the headline's C2 step (N=60 FP64, 40 layers, m=0..2, %d points; H2D + device optics + run + D2H) in the call order of "
                   "the reference's unpatched driver: for m: for iz: rt_kernel!(..., composite_layer, ...) -- %s" %
                   (points, "the composite kept in native layout by rt_kernel_'s per-composite registry (vsm_run_create at the TOA "
                    "call, one vsm_run_layer per call, lazy vsm_run_export by the surface interaction)" if native else
                    "every layer step on the reference-layout composite (vsm_layer_forward, one moment per launch)"),
                   points, wall, dev, scene.flops_per_point(), "f64",
                   "k_layer_native<4, 15> (m = 1, 2), k_layer_native<3, 10> (m = 0)" if native else "k_layer_strip_mm<15>")
        e["rt_kernel_calls_on_the_native_copy"] = int(sum(calls) // 2)     # (warm-up + timed step)
        e["rt_kernel_calls"] = 3 * L
        del scene
        return e
    finally:
        CR.REFERENCE_ORDER, CR.NATIVE_RUN, CR.NATIVE_DROPIN = saved


def c2_lin(vsm, torch, arch, o2a, points=2048, name="C2-lin"):   # (2048: 24 full rounds of 256 single-workgroup CUs per layer launch)
    L = 40
    tau_rayl, tau_abs = o2a(points, L)
    H = vsm.host_model
    model = H.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=2)
    lin = H.LinModel([tau_abs])
    scene = vsm.CoreRTLin.SceneLin(model, lin, 0, 1, 1)

    def step():
        _lin_step_inputs(scene)
        scene.run()
        return scene.results_host()
    wall, dev, _ = _timed(torch, step)
    e = _entry(name, "linearized rt_run (1 gas column + albedo), N=60 FP64, 40 layers, m=0..2 (C2 shape)", points, wall, dev,
               scene.flops_per_point(), "f64", "k_dbl_lin_multi<15,1> (+ k_ia128_lin<4>) for m = 1, 2; m = 0 as a Stokes_IQ scene (N = 40): "
               "k_dbl128_lin<3> + k_ia128_lin<3>")
    # the as-written count runs every parameter slot through every interaction (interaction_lin.jl:242,291); the surface slot of a
    # layer interaction is exact zeros and is not computed (vsm_interaction_lin_range): flops of the products that are formed
    N = float(scene.N)
    skipped = sum(len(mom["layers"]) - 1 for mom in scene.fwd.moments) * (scene.P - scene.pl) * (48 * N ** 3 + 16 * N ** 2)
    e["frac_of_mfma_peak_executed_products"] = (scene.flops_per_point() - skipped) * points / wall / 1e12 / PEAK["f64"]
    del scene
    return e


def c2_aer(vsm, torch, arch, o2a, points=10000):
    """SURVEY 8(d) aerosol variant of C2: HG aerosol (g = 0.7, ssa = 0.95, tau = 0.2) in the lowest 6 layers, Z mixed per spectral
    point, all 2 nstreams = 36 Fourier moments (bench.py --variant aerosol); the full step of the headline."""
    L, l_trunc = 40, 35
    tau_rayl, tau_abs = o2a(points, L)
    Hm = vsm.host_model
    nstreams = (l_trunc + 2) // 2
    tau_aer = np.zeros((1, L))
    tau_aer[0, -6:] = 0.2 / 6.0
    model = Hm.model_from_arrays(arch, "IQU", l_trunc, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15,
                                 m_max=2 * nstreams - 1, tau_aer=tau_aer,
                                 aerosol_optics=[Hm.AerosolOptics(Hm.henyey_greenstein_greek(0.7, 2 * nstreams - 1), 0.95, 0.0)])
    scene = vsm.CoreRT.prepare_scene(model)

    def step():
        scene.upload()
        scene.prepare()
        R, T = scene.run()
        return R.cpu(), T.cpu()
    wall, dev, _ = _timed(torch, step)
    nds = [ly["nd"] for ly in scene.moments[0]["layers"]]
    e = _entry("C2-aer", "C2 with the HG aerosol of SURVEY 8(d) in the lowest 6 layers: Z mixed per point, m=0..%d, ndoubl %d..%d; N=60 "
               "FP64, 40 layers; full step (H2D + device optics + run + D2H)" % (model.m_max, min(nds), max(nds)), points, wall, dev,
               scene.flops_per_point(), "f64", "k_layer_native<4, 15> (m = 1, 2), k_layer_native<3, 10> / <2, 5> (blocks of m = 0 and of "
               "the aerosol layers at m >= 3), k_native_diag_layer (Rayleigh-only layers at m >= 3: the Rayleigh phase matrix vanishes)")
    import bench
    ratio = bench.executed_over_algorithmic(scene)
    e["frac_of_mfma_peak_as_written"] = e["frac_of_mfma_peak"]
    e["executed_over_algorithmic_flops"] = ratio
    e["frac_of_mfma_peak"] = e["frac_of_mfma_peak_as_written"] * ratio
    e["note"] = ("frac_of_mfma_peak_as_written counts the dense 60 x 60 products of all 36 moments x 40 layers as the reference writes "
                 "them (SURVEY 8d) -- above 1 because the run forms only the products whose Stokes block the layer's phase matrices do "
                 "not leave exactly zero (the Rayleigh phase matrix vanishes for m >= 3: 34 of the 40 layers take a diagonal step there; "
                 "HG aerosol couples I alone: a 20-row block); frac_of_mfma_peak is the flops of the products that ARE formed, whole step")
    del scene
    return e


def c3_lin(vsm, torch, arch):
    import yaml
    with open(os.path.join(ROOT, "tests", "golden", "ocean_coxmunk_scene.json")) as f:
        d = json.load(f)
    d.pop("source")
    io, H = vsm.io_yaml, vsm.host_model
    model = io.model_from_parameters(io.parameters_from_yaml(yaml.safe_dump(d)), arch)
    S, L = model.tau_rayl.shape
    prof = np.linspace(0.2, 1.8, L)[None, :] * np.array([[0.004], [0.0015]])
    model.tau_abs = prof * 1.0
    lin = H.LinModel([prof * 1.0])
    scene = vsm.CoreRTLin.SceneLin(model, lin, 0, 1, 1)

    def step():
        _lin_step_inputs(scene)
        scene.run()
        return scene.results_host()
    wall_eager, _, ref = _timed(torch, step)
    # a scene that is stepped again and again (a retrieval loop; this bench) replays its pass from a HIP graph: the ~ 2500 launches and
    # library calls of a pass are host bound launch by launch (one thread issues them: the two layer chains never overlap on the
    # device), the graph is captured once per layer structure (ndoubl, interface tags) and reads the optics where prepare() puts them
    scene.graph_replay = True
    step()
    wall, dev, out = _timed(torch, step)
    same = all(np.array_equal(a, b) for a, b in zip(ref, out))
    fl = scene.flops_per_point() if hasattr(scene, "flops_per_point") else None
    e = _entry("C3-lin", "ocean / Cox-Munk scene (config/ocean_coxmunk.yaml): IQUV, N=60 FP64, 33 layers, m=0..21, linearized "
               "(gas column + wind speed) -- a latency-bound two-point batch; step = H2D + device optics + the pass REPLAYED FROM A HIP "
               "GRAPH + D2H", S, wall, dev, fl, "f64",
               "latency: ONE folded batch of (moment, point) pairs (m = 0 in front); the 33 layers doubled side by side on the lane "
               "streams (k_dbl128_lin<4>), then their interactions as a tree of composites (6 dependent levels, 2 k_ia128_lin<4> each); graph replay")
    e["ms_per_step_launch_by_launch"] = 1e3 * wall_eager
    e["graph_replay_equals_launch_by_launch"] = bool(same)
    del scene
    return e


def c5(vsm, torch, arch, points=4000, lines=40, layers=12, name="C5"):
    rng = np.random.default_rng(20260929)
    S, K, L = points, lines, layers
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S, 1))
    tau_abs = (10.0 ** rng.uniform(-4, 0, (S, 1))) * dp[None, :]
    H = vsm.host_model
    model = H.model_from_arrays(arch, "IQU", 9, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075, albedo=0.05, m_max=2)
    model.varpi_Cabannes = 0.96
    N = model.quad_points.Nquad * 3
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 7)
    rs = vsm.CoreRTRaman.RRS(shifts, np.full(len(shifts), 0.04 / len(shifts)), H.get_greek_rayleigh(0.75))
    wall, dev, _ = _timed(torch, lambda: vsm.CoreRTRaman.rt_run(rs, model, 1))
    n3, n2 = float(N) ** 3, float(N) ** 2
    lods = H.constructCoreOpticalProperties(model, 0)
    nds = [H.get_dtau_ndoubl(np.atleast_1d(lo.tau), np.broadcast_to(np.asarray(lo.varpi), np.atleast_1d(lo.tau).shape),
                             model.quad_points, np.float64, model.numerics)[1] for lo in lods]
    n1 = np.arange(S)
    kin = sum(((n1 + int(sh) >= 0) & (n1 + int(sh) < S)).astype(float) for sh in shifts).mean()
    # as-written operation count of the reference (doubling_inelastic.jl:13-164: 16 products + 10 mat-vecs per in-band line and
    # step; interaction_inelastic.jl:319-521: 18 products + 8 mat-vecs); the kernels execute 10 and 9 products per line
    per_m = sum(nd * (12 * n3 + 8 * n2 + kin * (32 * n3 + 20 * n2)) for nd in nds) + L * (24 * n3 + 8 * n2 + kin * (36 * n3 + 16 * n2))
    exe_m = sum(nd * (12 * n3 + 8 * n2 + kin * (20 * n3 + 16 * n2)) for nd in nds) + L * (24 * n3 + 8 * n2 + kin * (18 * n3 + 12 * n2))
    e = _entry(name, "rotational Raman (RRS), nStokes=3, N=%d FP64, %d layers, %d Raman lines, m=0..2 (BASELINE configs[4] at %d of its "
               "20000 points); step = whole rt_run(RRS) incl. host optics, H2D, D2H" % (N, L, len(shifts), S), S, wall, dev, 3 * per_m, "f64",
               "k_raman_doubling_chain<21> (+ k_raman_interaction_quad<21>) for m = 1, 2; m = 0 as a Stokes_IQ scene (N = 14): "
               "k_raman_doubling_chain<14> + k_raman_interaction_quad<14>")
    e["frac_of_mfma_peak_executed_products"] = 3 * exe_m * S / wall / 1e12 / PEAK["f64"]
    e["peak_device_memory_gb"] = torch.cuda.max_memory_allocated() / 1e9
    e["hbm_bytes_per_step"] = None   # whole-step HBM bytes from the newest committed PMC passes of the same workload
    import bench
    per_point, src = bench.c5_traffic_per_point(True)     # (newest committed PMC passes; None while their tag sources have changed)
    if per_point is not None:
        e["hbm_bytes_per_step"] = per_point * S
    e["hbm_bytes_source"] = src
    return e


def c1(vsm, torch, arch, points=100, layers=10, reps=20):
    H = vsm.host_model
    tau_rayl = np.tile(0.1 * np.diff(np.linspace(0.0, 1.0, layers + 1)), (points, 1)) * np.linspace(0.9, 1.1, points)[:, None]

    def call():   # everything a user's rt_run(model) costs, from the arrays of optical depths
        model = H.model_from_arrays(arch, "I", 5, 60.0, [60.0], [180.0], tau_rayl=tau_rayl, tau_abs=np.zeros_like(tau_rayl),
                                    depol=0.0279, albedo=0.15, m_max=2)
        return model, vsm.CoreRT.rt_run(model)

    def step():
        for _ in range(reps):
            out = call()
        return out
    wall, dev, (model, _) = _timed(torch, step)
    N = model.quad_points.Nquad * model.polarization_type.n
    sc = vsm.CoreRT.prepare_scene(model)
    e = _entry("C1", "quickstart-shaped (config/quickstart.yaml geometry: Stokes_I, nstreams=3, sza=vza=60 -> N=%d; Rayleigh, Lambertian "
               "0.15, %d layers, %d points, FP64, m=0..2); step = whole rt_run(model) incl. host model/optics, scene allocation, H2D, "
               "D2H; mean of %d calls" % (N, layers, points, reps), points, wall / reps, dev / reps, sc.flops_per_point(), "f64",
               "k_layer_native<1, KS> per layer, k_ia_native for the surface (launch-latency bound)")
    e["north_star_target_points_per_s"] = 1e4
    del sc
    return e


def ia_kernel(vsm, torch, arch, points=10240, N=60, reps=10, scale=1.0, name="IA"):
    FT = np.float64
    rng = np.random.default_rng(1)
    CR = vsm.CoreRT
    S = points
    pc = CR.make_composite_layer(FT, arch, (N, N), S)
    # the added layer as doubling! leaves it for a scattering layer of a Stokes_IQU run: r+- = D r-+ D, t-- = D t++ D are derived
    # inside the kernel (d_symmetric = nStokes), as in every unfused rt_kernel! step of a run whose layers all scatter
    pa = CR.make_added_layer(FT, arch, (N, N), S, d_symmetric=3)
    dev = pc.R_mp.device

    def refl(scale):   # physically shaped operators: small reflections, near-diagonal transmissions (tools/ia_timing.py)
        return torch.rand((S, N, N), dtype=torch.float64, device=dev) * (scale / N)

    def trans():
        d = 0.3 + 0.65 * torch.rand((S, N), dtype=torch.float64, device=dev)
        return torch.diag_embed(d) + torch.rand((S, N, N), dtype=torch.float64, device=dev) * (0.05 / N)
    # clear-sky scale of the C2 run: ||R+- r-+||_F ~ 2e-3, i.e. the series inverse at order 7 (the order the layer interactions of
    # the headline workload take); tools/ia_timing.py --refl sweeps the scale (orders 15 / 31 and the Gauss-Jordan path run out of
    # line at about half this rate)
    init = dict(R_mp=refl(0.1 * scale), R_pm=refl(0.1 * scale), T_pp=trans(), T_mm=trans(), J0_p=torch.rand((S, N), dtype=torch.float64, device=dev),
                J0_m=torch.rand((S, N), dtype=torch.float64, device=dev))
    pa.r_mp.copy_(refl(0.075 * scale))
    pa.t_pp.copy_(trans())
    pa.j0_p.copy_(torch.rand((S, N), dtype=torch.float64, device=dev))
    pa.j0_m.copy_(torch.rand((S, N), dtype=torch.float64, device=dev))
    warm = 10                          # untimed launches first: the clocks have dropped during the latency-bound entries before
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for it in range(warm + reps):
        for k, v in init.items():      # (the composite is updated in place: every repetition starts from the same operators)
            getattr(pc, k).copy_(v)
        if it >= warm:
            evs[it - warm][0].record()
        CR.interaction_("11", pc, pa)
        if it >= warm:
            evs[it - warm][1].record()
    torch.cuda.synchronize()
    tot = sum(a.elapsed_time(b) for a, b in evs)
    ms = tot / reps
    flop_pt = 24.0 * N ** 3 + 8.0 * N ** 2
    e = _entry(name, "interaction!(::ScatteringInterface_11) alone (interaction.jl:207-266), N=%d FP64, %d points per launch, physically "
               "shaped random layers at %s, added layer D-symmetric as doubling! leaves it; HIP events around each of %d launches"
               % (N, S, "the clear-sky scale of the C2 run (||R r|| ~ 2e-3: series inverse of order 7)" if scale == 1.0 else
                  "%g x the reflectances of the IA entry (||R r|| ~ %.0e: the long series orders 15 / 16 / 31, out of line)"
                  % (scale, 2e-3 * scale * scale), reps), S, ms * 1e-3, ms, flop_pt, "f64", "k_ia_native<4, 15, true>")
    if scale != 1.0:
        del pc, pa, init
        return e
    e["north_star_target_mfma_utilisation"] = 0.40
    # what the MFMA pipe executes: 10 products of the one-inverse interaction + 4 of the order-7 inverse in its factored form
    # (I + E)(I + E^2)(I + E^4) (ninvert7; Horner's rule took 6), each 4 waves x 60 v_mfma_f64_16x16x4 (2048 flop): 14 x 491 520
    # flop per point (the padded 64 x 64 x 60 tiles and the series are not in the algorithmic count)
    exe_pt = 14 * 4 * 60 * 2048.0
    e["mfma_utilisation_executed"] = exe_pt * S / (ms * 1e-3) / 1e12 / PEAK["f64"]
    e["note"] = ("frac_of_mfma_peak is ALGORITHMIC flops (24N^3+8N^2 per point) / launch time / 78.6; mfma_utilisation_executed is the "
                 "flops of the MFMA instructions the kernel issues (14 products of 240 v_mfma_f64_16x16x4 per point) / launch time / "
                 "78.6 -- the pipe's busy fraction, which the PMC pass measures directly (SQ_VALU_MFMA_BUSY_CYCLES, "
                 "profiles/r05/ia/summary.json)")
    del pc, pa, init
    return e


def n112(vsm, torch, arch, o2a, points=2000, layers=10):
    tau_rayl, tau_abs = o2a(points, layers)
    model = vsm.host_model.model_from_arrays(arch, "IQUV", 51, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                             albedo=0.15, m_max=2)
    N = model.quad_points.Nquad * 4
    scene = vsm.CoreRT.prepare_scene(model)

    def step():
        scene.upload()
        scene.prepare()
        R, T = scene.run()
        return R.cpu(), T.cpu()
    wall, dev, _ = _timed(torch, step)
    e = _entry("N112", "forward, IQUV N=%d FP64 (the reference's VLIDORT case-A size), %d layers, %d points, m=0..2, Rayleigh + O2, "
               "Lambertian" % (N, layers, points), points, wall, dev, scene.flops_per_point(), "f64",
               "k_layer_native<6, 21> (m = 1, 2: the 84-row (I,Q,U) block; V, which the Rayleigh matrix couples with nothing, as a 28-row "
               "block k_layer_native<2, 7>); m = 0 as its uncoupled blocks (k_layer_native<4, 14> for (I,Q)); the surface interaction on k_ia128<7>")
    _executed(e, scene, "Rayleigh: V couples with nothing, m = 0 splits into (I,Q) and (U,V)")
    del scene
    return e


def run_all(vsm, torch, arch, o2a):
    out = []
    for f in (lambda: c2_dropin(vsm, torch, arch, o2a, False), lambda: c2_dropin(vsm, torch, arch, o2a, True),
              lambda: c4(vsm, torch, arch, o2a), lambda: c2_aer(vsm, torch, arch, o2a), lambda: c2_lin(vsm, torch, arch, o2a),
              lambda: c2_lin(vsm, torch, arch, o2a, 10000, "C2-lin-10k"), lambda: c3_lin(vsm, torch, arch),
              lambda: c5(vsm, torch, arch), lambda: c5(vsm, torch, arch, 10000, name="C5-10k"), lambda: c1(vsm, torch, arch),
              lambda: ia_kernel(vsm, torch, arch), lambda: ia_kernel(vsm, torch, arch, scale=4.0, name="IA-long"),
              lambda: n112(vsm, torch, arch, o2a)):
        try:
            out.append(f())
        except Exception as ex:   # a secondary workload must not take the headline line down
            out.append({"error": "%s: %s" % (type(ex).__name__, ex)})
        torch.cuda.empty_cache()
    return out
