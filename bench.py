#!/usr/bin/env python3
"""bench.py -- spectral-points/s of an rt_run-equivalent on synthetic O2-A-band atmospheres.

Workload (BASELINE.json configs[1], "C2"): O2-A band 759-770 nm, nStokes = 3, Nquad = 20
(18 Gauss-Legendre + SZA 40 deg + VZA 30 deg => N = 60), 40 layers, 10 000 spectral points per GPU,
FP64, Rayleigh (depol 0.0279) + synthetic O2-like absorption (40 pseudo-lines, column tau 1e-4..50),
Lambertian 0.15, Fourier moments m = 0..2.   A "step" = one full rt_run-equivalent over the batch:
H2D of the raw optical depths, layer optics on the device, all m, all layers (elemental -> doubling ->
interaction), surface, VZA post-processing, ONE gather of R/T over the ranks and the D2H of the result.
N GPUs => N x 10 000 points (weak scaling; `--total-points` = strong scaling).  `--config C4`: configs[3] (N = 96, FP32; 10^5 points
in total over the GPUs), `--config C5`: configs[4] (rotational Raman, 2 10^4 points in total, halo-extended blocks per rank).  `--gpus N` without a launcher
starts its own N ranks (torch.distributed.run, one process per GPU over RCCL) or fails if N GPUs are not visible.

Prints ONE JSON line on rank 0 (contract in the task statement), including `roofline` for the dominant
kernel (the native-layout layer step k_layer_native<4, 15> with its elemental pre-pass k_elemental_native<4, ...>; algorithmic flops /
HIP-event time of the pair) and `cpu_baseline` (the oracle port timed on a bounded sample of the same workload, with `anchor`: the
same port on the reference's published CPU shape).  `config.secondary` holds the other configurations and the two DROP-IN entries:
the same C2 step issued in the call order of the reference's unpatched driver (rt_run.jl:383-453).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # MI355X dense MFMA peaks


def o2a_atmosphere(S_total, L, seed=20260929):
    """Synthetic inputs of SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    p = np.linspace(0.0, 1.0, L + 1)
    dp = np.diff(p)                                  # pressure-proportional layer split
    tau_rayl = np.tile(0.025 * dp, (S_total, 1))
    nu = np.linspace(12987.0, 13175.0, S_total)
    nu_k = rng.uniform(12990.0, 13170.0, 40)
    A_k = 10.0 ** rng.uniform(-3, math.log10(30.0), 40)
    gam = 0.08
    col = np.sum(A_k[None, :] * gam ** 2 / ((nu[:, None] - nu_k[None, :]) ** 2 + gam ** 2), axis=1) + 1e-4
    tau_abs = col[:, None] * dp[None, :]
    return tau_rayl, tau_abs


CONFIGS = {
    # name: (polarization, l_trunc, FT, layers, points per GPU)
    "C2": dict(pol="IQU", l_trunc=35, FT="f64", L=40, S=10000, N=60),
    "C4": dict(pol="IQU", l_trunc=59, FT="f32", L=60, S=12500, N=96),
    # BASELINE.json configs[4]: rotational Raman, 20 000 spectral points IN TOTAL (strong scaling; the reference quotes it on 2 GPUs)
    "C5": dict(pol="IQU", l_trunc=9, FT="f64", L=12, S=20000, N=21, K=40),
}


def native_rt(n):
    """Row tiles of the native layer kernel of a block of n rows (vsm_native.hip rt_of)."""
    ks = (n + 3) // 4
    return ks // 4 if (ks % 4 == 0 and ks >= 4) else (4 * ks + 2 + 15) // 16


def stokes_groups(ns, coupling):
    """Blocks of Stokes components that the phase matrices couple (bit masks), from the mask of vsm_stokes_coupling: the
    connected components the library's run object forms (vsm_native.hip stokes_groups)."""
    if coupling is None or coupling < 0:
        return [(1 << ns) - 1]
    seen, out = 0, []
    for a in range(ns):
        if seen >> a & 1:
            continue
        grp, stack = 1 << a, [a]
        while stack:
            x = stack.pop()
            for b in range(ns):
                if not grp >> b & 1 and ((coupling >> (4 * x + b)) & 1 or (coupling >> (4 * b + x)) & 1):
                    grp |= 1 << b
                    stack.append(b)
        seen |= grp
        out.append(grp)
    return out


def executed_over_algorithmic(scene, native=None):
    """Executed / as-written flops of the layer steps of a scene: a block of n rows costs (n / N)^3 of the dense count, a block
    that the phase matrices of a layer leave exactly zero costs none (an elementwise scaling of the composite); moments outside
    `native` (default: the scene's native moments) run dense."""
    N = scene.N
    n3, n2 = float(N) ** 3, float(N) ** 2
    if native is None:
        native = set(scene._native_moments())
    ex_num = ex_den = 0.0
    for i, mom in enumerate(scene.moments):
        for iz, ly in enumerate(mom["layers"]):
            wl = ly["nd"] * (12 * n3 + 8 * n2) + (24 * n3 + 8 * n2 if iz else 0.0)
            ex_den += wl
            if i not in native:
                ex_num += wl
                continue
            lc = scene._layer_coupling(mom["m"], iz)
            for g in stokes_groups(scene.pol.n, scene.coupling[mom["m"]]):
                comps = [a for a in range(scene.pol.n) if g >> a & 1]
                if any(lc >> (4 * a + b) & 1 for a in comps for b in comps):
                    ex_num += wl * (scene.N // scene.pol.n * len(comps) / float(N)) ** 3
    return ex_num / ex_den if ex_den else 1.0


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def make_step(scene, parallel, S_total, rank, world, phase=None, sync=None):
    """The timed step of the bench as a closure over `scene` (anything with upload() / prepare() / run() -> (R, T) torch tensors
    whose first axis is the rank's spectral block: the HIP Scene here, an oracle-backed stand-in in the 2-rank gloo test that
    drives this very function):  H2D of the raw optical depths, layer optics, all Fourier moments x layers x (elemental ->
    doubling -> interaction), surface, post-processing, ONE gather of R/T over the ranks (packed into one buffer per rank),
    D2H of the result on rank 0.  `split` adds synchronisations to attribute the time to phases (untimed pass)."""
    import torch

    def step(split=False):
        t = [time.perf_counter()]

        def mark():
            if split:
                if sync is not None:
                    sync()
                t.append(time.perf_counter())

        scene.upload()
        mark()
        scene.prepare()
        mark()
        R, T = scene.run()
        mark()
        g = parallel.gather_spectral(parallel.pack_RT(R, T), S_total, rank, world)   # the one collective step (RCCL) of the data path
        out = g.cpu() if g is not None else None                     # D2H of R/T [nSpec, 2 nStokes nVZA] on rank 0
        mark()
        if split and phase is not None:
            for k, a, b in zip(phase, t[:-1], t[1:]):
                phase[k] = b - a
        return out
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--points", type=int, default=None, help="spectral points per GPU (default: config)")
    ap.add_argument("--total-points", type=int, default=None,
                    help="strong scaling: this many spectral points in total, split over the GPUs (C4: 100000)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--variant", default="rayleigh", choices=["rayleigh", "aerosol"],
                    help="aerosol: SURVEY 8(d) variant -- HG aerosol (g=0.7, ssa=0.95, tau=0.2) in the lowest 6 layers, "
                         "2*nstreams-1 moments: Z differs per point and all 2*nstreams Fourier moments run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip config.secondary (one timed step each of C4, the linearized C2 shape, the linearized C3 scene, C5 at 4000 points)")
    ap.add_argument("--cpu-sample", type=int, default=96, help="upper bound of spectral points PER HOST CORE of the CPU-baseline sample (sized for ~15 s)")
    args = ap.parse_args()

    import torch
    # --gpus N without a launcher: start the N ranks ourselves (one process per GPU over RCCL), or fail loudly
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible -- refusing to report a smaller run" % (args.gpus, have))
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
                  + sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))

    import vsmartmom_jl_amd as vsm
    from vsmartmom_jl_amd import parallel

    rank, world, local = parallel.init_process_group_from_env()
    if args.config == "C5":
        return bench_c5(args, vsm, parallel, torch, rank, world, local)
    cfg = dict(CONFIGS[args.config], name=args.config)
    if args.points:
        cfg["S"] = args.points
    if args.layers:
        cfg["L"] = args.layers
    scaling = "weak"
    if args.config == "C4" and args.gpus > 1 and not args.total_points and not args.points:
        args.total_points = 100000   # BASELINE.json configs[3] is a STRONG-scaling case: 10^5 points over the GPUs of the node
    if args.total_points:
        scaling = "strong"
        if args.total_points % world:
            sys.exit("bench.py: --total-points must be divisible by --gpus")
        cfg["S"] = args.total_points // world
    FT = np.float64 if cfg["FT"] == "f64" else np.float32
    S_local, L = cfg["S"], cfg["L"]
    S_total = S_local * world
    arch = vsm.Architectures.GPU(local)
    vsm._lib.lib()  # fail loudly if the HIP library is absent

    tau_rayl, tau_abs = o2a_atmosphere(S_total, L)
    extra = {}
    m_max = 2
    if args.variant == "aerosol":
        Hm = vsm.host_model
        nstreams = (cfg["l_trunc"] + 2) // 2
        tau_aer = np.zeros((1, L))
        tau_aer[0, -6:] = 0.2 / 6.0
        extra = dict(tau_aer=tau_aer, aerosol_optics=[Hm.AerosolOptics(Hm.henyey_greenstein_greek(0.7, 2 * nstreams - 1), 0.95, 0.0)])
        m_max = 2 * nstreams - 1
        args.no_cpu_baseline = True
    model = vsm.host_model.model_from_arrays(arch, cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tau_rayl,
                                             tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=m_max, float_type=FT, **extra)
    N = model.quad_points.Nquad * model.polarization_type.n
    assert N == cfg["N"], (N, cfg["N"])
    sl = parallel.shard_slice(S_total, rank, world)
    scene = vsm.CoreRT.prepare_scene(model, sl)     # allocations (the reference's make_added_layer / make_composite_layer)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    phase = {"h2d": 0.0, "optics": 0.0, "run": 0.0, "gather_d2h": 0.0}

    step = make_step(scene, parallel, S_total, rank, world, phase, torch.cuda.synchronize)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    timed_region_s = dt
    mg = {"rccl_ranks": 1, "per_rank_step_ms": [1e3 * dt / args.steps], "per_rank_device": [torch.cuda.get_device_name(local)]}
    if world > 1:
        dist = torch.distributed
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)      # (a real RCCL collective: the rank count below is observed after it)
        mg = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
        try:   # (diagnostics only: nothing here may take the bench line down)
            every = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(every, torch.tensor([dt], dtype=torch.float64, device="cuda"))
            names = [None] * world
            dist.all_gather_object(names, "%s (cuda:%d)" % (torch.cuda.get_device_name(local), local))
            mg.update(per_rank_step_ms=[1e3 * float(x.item()) / args.steps for x in every], per_rank_device=names)
            mg["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as ex:
            mg["note"] = "per-rank diagnostics unavailable: %s" % ex
        dt = float(tmax.item())
        timed_region_s = dt
    step(split=True)   # phase attribution, outside the timed region

    # ---- roofline of the dominant kernel: timed live with events on the launch stream ----------
    # one extra pass with an event pair around every layer-step call; the same pass gives the device-only time of run()
    ev = []
    orig, orig_multi, orig_native = vsm.CoreRT.layer_forward_, vsm.CoreRT.layer_forward_multi_, vsm.CoreRT.run_layer_native_

    def timed_with(f, rec):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f(*a, **k)
            e1.record()
            ev.append((e0, e1) + rec(a))   # + (ndoubl, toa, Fourier moments in the call, native-layout run?)
        return timed

    vsm.CoreRT.layer_forward_ = timed_with(orig, lambda a: (a[5], a[7], 1, False))
    vsm.CoreRT.layer_forward_multi_ = timed_with(orig_multi, lambda a: (a[5], a[7], len(a[4]), False))
    vsm.CoreRT.run_layer_native_ = timed_with(orig_native, lambda a: (a[2], a[12], a[1], True))
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r0.record()
    scene.run()
    r1.record()
    torch.cuda.synchronize()
    run_ms = r0.elapsed_time(r1)
    n3, n2 = float(N) ** 3, float(N) ** 2
    step_flops = lambda evs: sum(nmom * S_local * (nd * (12 * n3 + 8 * n2) + (0 if toa else 24 * n3 + 8 * n2))
                                 for _, _, nd, toa, nmom, _nat in evs)   # ALGORITHMIC flops, as written (dense N x N)
    step_ms = sum(e[0].elapsed_time(e[1]) for e in ev)
    step_tflops = step_flops(ev) / (step_ms * 1e-3) / 1e12 if step_ms > 0 else 0.0
    layer_step = {"layers": scene.Nz, "avg_ms": step_ms / max(scene.Nz, 1), "algorithmic_tflops": step_tflops,
                  "frac_of_mfma_peak_algorithmic": step_tflops / PEAK_TFLOPS[cfg["FT"]],
                  "fourier_moments": len(scene.moments)}
    native = sorted(scene._native_moments())
    interval_kernels = None
    if native:
        # The run walks the layers on the native-layout composite (vsm_run_*): a layer step = per class of sub-problems the
        # elemental pre-pass + ONE k_layer_native<RT, KS> launch.  Moments whose Stokes blocks all couple are dense N x N problems
        # (C2: m = 1, 2 on k_layer_native<4, 15>: the dominant kernel); the blocks of m = 0 run as independent sub-problems and
        # execute fewer products than the as-written count.  The dominant kernel is timed by itself: one more walk of the layers
        # with the dense moments only (event pair = its pre-pass + the layer launch), against ITS algorithmic flops.
        sub = lambda mom: scene.N // scene.pol.n * max(bin(g).count("1") for g in stokes_groups(scene.pol.n, scene.coupling[mom["m"]]))
        dense = [i for i in native if sub(scene.moments[i]) == N]
        ev_all, ev = ev, []
        if dense:
            scene._run_layers_native([scene.moments[i] for i in dense], scene._composites[:len(dense)])
            torch.cuda.synchronize()
        ev_dense, ev = ev, ev_all
        if dense:
            k_ms = sum(e[0].elapsed_time(e[1]) for e in ev_dense)
            k_flops = step_flops(ev_dense)
            ks = (N + 3) // 4
            fam = "native32" if cfg["FT"] == "f32" else "native"   # (Float32 models: the FP32 family, vsm_native32.hip)
            kernel_name = "k_layer_%s<%d, %d>" % (fam, native_rt(N), ks)
            interval_kernels = ["k_elemental_" + fam, kernel_name]
            moments_per_launch = len(dense)
            n_launch = len(ev_dense)
        else:   # the dense moments are beyond the native kernels: their layer steps run on the reference-layout kernels
            legacy = [e for e in ev if not e[5]]
            k_ms = sum(e[0].elapsed_time(e[1]) for e in legacy)
            k_flops, n_launch = step_flops(legacy), len(legacy)
            moments_per_launch = max([e[4] for e in legacy], default=1)
        ex_ratio = executed_over_algorithmic(scene, native)
        layer_step["executed_over_algorithmic_flops"] = ex_ratio
        layer_step["frac_of_mfma_peak_executed"] = layer_step["frac_of_mfma_peak_algorithmic"] * ex_ratio
        layer_step["note"] = ("blocks of Stokes components that do not couple (m = 0: (I,Q) and U) run as independent sub-problems, a "
                              "block whose phase matrix is exactly zero in a layer takes a diagonal step: products with exact "
                              "zeros are not formed")
    else:
        k_ms, k_flops, n_launch = step_ms, step_flops(ev), len(ev)
        moments_per_launch = max([e[4] for e in ev], default=1)
    vsm.CoreRT.layer_forward_, vsm.CoreRT.layer_forward_multi_, vsm.CoreRT.run_layer_native_ = orig, orig_multi, orig_native
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    peak = PEAK_TFLOPS[cfg["FT"]]
    if native and dense:
        pass
    elif cfg["FT"] == "f64" and 32 < N <= 64:
        # the event pair of a layer step brackets the fused layer kernel AND its elemental pre-pass (k_elemental_img writes the
        # A-form images the layer kernel copies in): the fraction below charges both to the layer kernel's flops
        kernel_name = "k_layer_strip_mm"
        interval_kernels = ["k_elemental_img", "k_layer_strip_mm"]
    elif cfg["FT"] == "f32" and 64 < N <= 96:
        kernel_name = "k_layer_strip32_mm" if moments_per_launch > 1 else "k_layer_strip32"
        interval_kernels = ["k_elemental_img32", "k_layer_strip32_mm"] if moments_per_launch > 1 else ["k_layer_strip32"]
    else:
        kernel_name = "k_elemental_doubling + k_interaction11"
    # the committed PMC passes cover the default (Rayleigh, m = 0..2) workload of a config only
    build = vsm._lib.build_info()
    tk = ([kernel_name, "k_elemental_%s<%d," % ("native32" if cfg["FT"] == "f32" else "native", native_rt(N))] if native and dense
          else [kernel_name, "k_elemental_img"])
    traffic, traffic_src, traffic_hash, traffic_note = (hbm_traffic_per_launch(tk, cfg, S_local, build["source_hash"])
                                                        if args.variant == "rayleigh" else (None, None, None, "no profile of this variant"))

    if rank == 0:
        flops_pt = scene.flops_per_point()
        pts_per_s = S_total * args.steps / dt
        nds = [ly["nd"] for ly in scene.moments[0]["layers"]]
        line = {
            "metric": "spectral-points/s (whole node) for rt_run, O2-A band",
            "value": pts_per_s, "unit": "spectral-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": cfg["FT"], "data": "synthetic",
            "config": {"workload": "%s: O2-A 759-770 nm, nStokes=3, Nquad=%d (N=%d), %d layers, %d spectral points/GPU, "
                                   "m=0..%d, Rayleigh%s+synthetic O2 absorption, Lambertian 0.15" %
                                   (args.config, model.quad_points.Nquad, N, L, S_local, m_max,
                                    "+HG aerosol (lowest 6 layers)" if args.variant == "aerosol" else ""),
                       "timed_step": "full rt_run-equivalent: H2D of tau_rayl/tau_abs + device layer optics + all moments/layers/"
                                     "surface/post-processing + gather + D2H of R,T",
                       "ranks": world, "collective_backend": "nccl (RCCL)" if world > 1 else "none (1 rank)",
                       "multi_gpu": mg, "timed_region_s": timed_region_s,
                       "N": N, "layers": L, "points_per_gpu": S_local, "fourier_moments": m_max + 1,
                       "ndoubl_per_layer": nds, "algorithmic_gflop_per_point": flops_pt / 1e9,
                       "whole_run_tflops": pts_per_s * flops_pt / 1e12,
                       "whole_run_frac_of_mfma_peak": pts_per_s * flops_pt / 1e12 / (peak * world),
                       "phase_ms (one extra untimed pass with syncs)": {k: 1e3 * v for k, v in phase.items()},
                       "device_pass_ms (run() only, HIP events)": run_ms,
                       "device_pass_points_per_s_per_gpu": S_local / (run_ms * 1e-3)},
            "roofline": {"bound": "mfma", "kernel": kernel_name, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "traffic_profile_head": traffic_hash, "traffic_note": traffic_note, "library": build,
                         "launches": n_launch, "avg_launch_ms": k_ms / max(n_launch, 1),
                         "kernels_in_timed_interval": interval_kernels,
                         "fourier_moments_per_launch": moments_per_launch,
                         "layer_step_all_moments": layer_step},
        }
        if world == 1 and args.config == "C2" and args.variant == "rayleigh" and not args.no_secondary and not args.points:
            import bench_secondary
            del scene
            torch.cuda.empty_cache()
            t_sec = time.perf_counter()
            line["config"]["secondary"] = bench_secondary.run_all(vsm, torch, arch, o2a_atmosphere)
            line["config"]["secondary_wall_s"] = time.perf_counter() - t_sec
        torch.cuda.synchronize()   # (marker: everything on the GPU is done before the CPU baseline starts)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample, L)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def bench_c5(args, vsm, parallel, torch, rank, world, local):
    """`--config C5`: rt_run(RS_type::RRS, model, iBand) on BASELINE.json configs[4] (nStokes = 3, N = 21, 12 layers, K = 40 Raman
    lines, m = 0..2, 20 000 spectral points in TOTAL unless --points gives the points per GPU): every rank computes on its block
    of recipient points extended by the halo of max|shift| donor points (no exchange step), then ONE gather per output array
    (R, T, ieR, ieT).  A step = the whole call: host optics, H2D, device pass, gather, D2H."""
    cfg = dict(CONFIGS["C5"], name="C5")
    S_total = args.points * world if args.points else (args.total_points or cfg["S"])
    L, K = args.layers or cfg["L"], cfg["K"]
    arch = vsm.Architectures.GPU(local)
    vsm._lib.lib()
    rng = np.random.default_rng(20260929)
    dp = np.full(L, 1.0 / L)
    tau_rayl = np.tile(0.3 * dp, (S_total, 1))
    tau_abs = (10.0 ** rng.uniform(-4, 0, (S_total, 1))) * dp[None, :]
    H = vsm.host_model
    model = H.model_from_arrays(arch, cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0075,
                                albedo=0.05, m_max=2)
    model.varpi_Cabannes = 0.96
    N = model.quad_points.Nquad * 3
    shifts = np.unique(np.concatenate([np.arange(-K // 2, 0), np.arange(1, K - K // 2 + 1)]) * 7)
    rs = vsm.CoreRTRaman.RRS(shifts, np.full(len(shifts), 0.04 / len(shifts)), H.get_greek_rayleigh(0.75))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        return vsm.CoreRTRaman.rt_run_sharded(rs, model, rank, world)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    mg = {"rccl_ranks": 1, "per_rank_step_ms": [1e3 * dt / args.steps], "per_rank_device": [torch.cuda.get_device_name(local)]}
    if world > 1:
        dist = torch.distributed
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mg = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
        try:   # (diagnostics only)
            every = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(every, torch.tensor([dt], dtype=torch.float64, device="cuda"))
            names = [None] * world
            dist.all_gather_object(names, "%s (cuda:%d)" % (torch.cuda.get_device_name(local), local))
            mg.update(per_rank_step_ms=[1e3 * float(x.item()) / args.steps for x in every], per_rank_device=names)
        except Exception as ex:
            mg["note"] = "per-rank diagnostics unavailable: %s" % ex
        dt = float(tmax.item())
    if rank == 0:
        n3, n2 = float(N) ** 3, float(N) ** 2
        lods = H.constructCoreOpticalProperties(model, 0)
        nds = [H.get_dtau_ndoubl(np.atleast_1d(lo.tau), np.broadcast_to(np.asarray(lo.varpi), np.atleast_1d(lo.tau).shape),
                                 model.quad_points, np.float64, model.numerics)[1] for lo in lods]
        n1 = np.arange(S_total)
        kin = sum(((n1 + int(sh) >= 0) & (n1 + int(sh) < S_total)).astype(float) for sh in shifts).mean()
        per_m = sum(nd * (12 * n3 + 8 * n2 + kin * (32 * n3 + 20 * n2)) for nd in nds) + L * (24 * n3 + 8 * n2 + kin * (36 * n3 + 16 * n2))
        exe_m = sum(nd * (12 * n3 + 8 * n2 + kin * (20 * n3 + 16 * n2)) for nd in nds) + L * (24 * n3 + 8 * n2 + kin * (18 * n3 + 12 * n2))
        pts = S_total * args.steps / dt
        tf = 3 * per_m * pts / 1e12
        print(json.dumps({
            "metric": "spectral-points/s (whole node) for rt_run(RRS), BASELINE configs[4]", "value": pts, "unit": "spectral-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.points else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C5: rotational Raman (RRS), nStokes=3, N=%d, %d layers, %d spectral points in total, %d Raman lines, "
                                   "m=0..2; ranks own contiguous blocks of recipient points + halo of %d donor points" %
                                   (N, L, S_total, len(shifts), int(np.max(np.abs(shifts)))),
                       "timed_step": "whole rt_run(RRS) per rank (host optics, H2D, device pass) + one gather per output + D2H",
                       "ranks": world, "multi_gpu": mg, "timed_region_s": dt, "algorithmic_gflop_per_point": 3 * per_m / 1e9,
                       "in_band_lines_per_point": kin},
            "roofline": {"bound": "mfma", "kernel": "whole run (k_raman_doubling_chain<21> ~67 %, k_raman_interaction_quad<21> ~17 %)",
                         "achieved": tf, "peak": PEAK_TFLOPS["f64"] * world, "unit": "TFLOP/s", "frac": tf / (PEAK_TFLOPS["f64"] * world),
                         "frac_executed_products": 3 * exe_m * pts / 1e12 / (PEAK_TFLOPS["f64"] * world),
                         "traffic": c5_traffic_per_point() and c5_traffic_per_point() * S_total,
                         "traffic_source": c5_traffic_per_point(True)[1],
                         "traffic_note": "HBM bytes of one whole step (all kernels; FETCH_SIZE x 2 + WRITE_SIZE per point from "
                                         "the newest profiles/rNN/c5/summary.json whose tag sources are unchanged, a 4000-point run, x "
                                         "the points of this run)"}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02/final")


def c5_traffic_per_point(with_source=False):
    """Whole-run HBM bytes per spectral point of the C5 workload from the newest committed PMC passes (profiles/rNN/c5/summary.json)
    -- quoted only while the sources that decide what that tag measures are what they were when it was taken (`tag_sources_hash`,
    the guard of hbm_traffic_per_launch): a profile of other kernels gives None."""
    for rnd in PROFILE_ROUNDS[:3]:
        path = os.path.join(ROOT, "profiles", rnd, "c5", "summary.json")
        try:
            with open(path) as f:
                prof = json.load(f)
            val = float(prof["hbm_bytes_per_point_whole_run"])
        except (OSError, KeyError, ValueError):
            continue
        fresh = (prof.get("tag_sources") and prof.get("tag_sources_hash")
                 and tag_sources_hash(prof["tag_sources"], prof["command"]) == prof["tag_sources_hash"])
        if not fresh:
            try:
                import vsmartmom_jl_amd as vsm
                fresh = (prof.get("library") or {}).get("source_hash") == vsm._lib.build_info()["source_hash"]
            except Exception:
                fresh = False
        src = os.path.relpath(path, ROOT)
        if not fresh:
            return (None, src + " (stale: taken on other sources)") if with_source else None
        return (val, src) if with_source else val
    return (None, None) if with_source else None


def tag_sources_hash(sources, command):
    """tools/profile_any.py's `tag_sources_hash`: SHA-256 over the files a profile tag depends on (+ its command)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted({f for pat in sources.split(",") for f in glob.glob(os.path.join(ROOT, pat.strip()))}):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0" + open(f, "rb").read())
    h.update(command.encode())
    return h.hexdigest()[:16]


def hbm_traffic_per_launch(kernels, cfg, S_local, running_hash=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    rocprofv3 runs by tools/profile_any.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): the per-point
    figure of profiles/<round>/<config>/summary.json (the same command: C2 at the bench's own 10^4 points -- c2_10k -- else a
    4096-point run) times the points of one launch.
    PMC counters cannot be read from inside the timed process, so this is the profiled value, not a live one -- and it is only
    quoted for the build it was taken on: a summary names the library it profiled (`library.source_hash`, csrc/Makefile), and a
    profile of another build gives (None, path, its hash, note).  Returns (bytes per launch | None, path, profile hash, note)."""
    tags = {"C2": ("c2_10k", "c2"), "C4": ("c4",)}.get(cfg.get("name"))   # (c2_10k: the PMC passes at the bench's own batch)
    if tags is None:
        return None, None, None, "no committed PMC profile for this configuration"
    stale = None
    for rnd, tag in [(r, t) for r in PROFILE_ROUNDS for t in tags]:
        path = os.path.join(ROOT, "profiles", rnd, tag, "summary.json")
        try:
            prof = json.load(open(path))
            pts = int(prof["command"].split("--points")[1].split()[0])
            per_point = 0.0
            want = [k.replace(" ", "") for k in kernels]
            for name, rec in prof["kernels"].items():   # the layer kernel and its elemental pre-pass
                if any(k in name.replace(" ", "") for k in want) and "fetch_bytes_per_launch" in rec:
                    per_point += (rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]) / pts
            if per_point <= 0.0:
                continue
            ph = (prof.get("library") or {}).get("source_hash")
            rel = os.path.relpath(path, ROOT)
            if prof.get("tag_sources") and prof.get("tag_sources_hash"):
                # the sources that decide what this tag measures (kernels of the workload + host path) are unchanged since the
                # profile was taken: its per-launch bytes hold for the running build even if other kernel families were edited
                if tag_sources_hash(prof["tag_sources"], prof["command"]) == prof["tag_sources_hash"]:
                    return per_point * S_local, rel, ph, "tag sources unchanged since the profile (tag_sources_hash %s)" % prof["tag_sources_hash"]
            if running_hash is not None and ph != running_hash:
                stale = stale or (None, rel, ph, "the newest committed profile (%s) was taken on library build %s, this run is build %s: "
                                  "traffic withheld" % (rel, ph or "unnamed (before round 4)", running_hash))
                continue
            return per_point * S_local, rel, ph, None
        except Exception:
            pass
    return stale or (None, None, None, "no committed PMC profile for this configuration")


def _cpu_worker(job):
    """One worker of the FP32 CPU baseline: the numpy oracle on its share of the sample (1 BLAS thread per worker)."""
    cfg, L, idx = job
    from oracle import vsm_oracle as O
    FT = np.float64 if cfg["FT"] == "f64" else np.float32
    if idx is None:   # warm-up: import + page-in only
        return 0.0
    tau_rayl, tau_abs = o2a_atmosphere(cfg["S"], L)
    mdl = O.build_model(cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tau_rayl[idx], tau_abs=tau_abs[idx],
                        depol=0.0279, albedo=0.15, m_max=2, FT=FT)
    t0 = time.perf_counter()
    O.rt_run(mdl)
    return time.perf_counter() - t0


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (cpu.max: the GPU boxes expose 256
    hardware threads under a 16-CPU quota -- 256 threads then run 2x slower in total than 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_anchor(cores):
    """the same C + OpenMP port on the ONE shape for which the reference publishes a CPU figure -- noRS rt_run of
    test/test_parameters/Phase1b_RRS_761-764nm.yaml: 103 spectral points, 12 layers, Stokes_IQU, nstreams = 3 (N = 15), Float32,
    0.304 s wall = 340 points/s on a 64-core EPYC 7H12 (dev_notes/phase5_headroom.md:7-14) -- so that the port's rate at C2 can be
    placed against a number of the reference itself.  Synthetic optical depths of that shape (no HITRAN here); the port computes in
    FP64 with ndoubl from the reference's Float32 rule (the 1024 eps(Float32) floor binds: fewer doublings than FP64)."""
    from oracle import vsm_oracle as O, vsm_oracle_c as OC
    S, L = 103, 12
    tau_rayl, tau_abs = o2a_atmosphere(S, L)
    mdl = O.build_model("IQU", 5, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=2)
    N = mdl.quad_points.Nquad * 3
    nthr = min(cores, S)
    OC.rt_run(mdl, nthreads=nthr, ndoubl_float_type=np.float32)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        OC.rt_run(mdl, nthreads=nthr, ndoubl_float_type=np.float32)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": S / med, "unit": "spectral-points/s", "cores": nthr, "wall_s_median_of_5": med,
            "shape": "noRS rt_run, %d points, %d layers, Stokes_IQU, nstreams=3 (N=%d), m=0..2, whole call incl. the numpy host optics" % (S, L, N),
            "reference_published": {"value": 340.0, "wall_s": 0.304, "hardware": "AMD EPYC 7H12 64-core, Julia 1.12.5, Float32",
                                    "source": "dev_notes/phase5_headroom.md:7-14"},
            "note": "the port at the reference's own published CPU shape: the ratio value / 340 says how this box's %d cores and "
                    "the C port compare with the Julia CPU path on 64 EPYC cores (a 103-point call is dominated by fixed costs on "
                    "both sides)" % nthr}


def cpu_baseline(cfg, n_per_core, L):
    """CPU baseline on a bounded sample of the same workload, on ALL host cores of the box.

    FP64 configurations: the C + OpenMP restatement (oracle/vsm_oracle_c.c; one LU per point, threads over the spectral
    axis -- the structure of the reference's CPU path, src/CoreRT/tools/cpu_batched.jl:25-82), `n_per_core` points per core,
    evenly spaced over the band, all layers, all moments.  FP32 configurations: the numpy oracle in single precision, one
    single-threaded worker process per core.  ndoubl of the sample is recomputed from the sample's own max(tau*varpi); with
    Rayleigh-only scattering (spectrally flat tau*varpi) it equals the full batch's."""
    cores = host_cores()
    n_sample = cores * n_per_core
    idx = np.linspace(0, cfg["S"] - 1, n_sample).astype(int)
    if cfg["FT"] == "f64":
        from oracle import vsm_oracle as O, vsm_oracle_c as OC
        OC.lib()
        tau_rayl, tau_abs = o2a_atmosphere(cfg["S"], L)
        mk = lambda ii: O.build_model(cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tau_rayl[ii], tau_abs=tau_abs[ii],
                                      depol=0.0279, albedo=0.15, m_max=2)
        OC.rt_run(mk(idx[:cores]), nthreads=cores)      # thread start-up + page-in, untimed
        t0 = time.perf_counter()
        OC.rt_run(mk(idx[:2 * cores]), nthreads=cores)  # calibration: size the sample for ~15 s of wall time
        rate = 2 * cores / (time.perf_counter() - t0)
        n_per_core = int(min(max(4, round(15.0 * rate / cores)), n_per_core, max(1, cfg["S"] // cores)))
        n_sample = cores * n_per_core
        idx = np.linspace(0, cfg["S"] - 1, n_sample).astype(int)
        mdl = mk(idx)
        t0 = time.perf_counter()
        OC.rt_run(mdl, nthreads=cores)
        dt = time.perf_counter() - t0
        return {"value": n_sample / dt, "unit": "spectral-points/s", "cores": cores, "kind": "port",
                "sample": "%d of the %d spectral points (evenly spaced, %d per core), all %d layers, m=0..2, C + OpenMP "
                          "restatement (oracle/vsm_oracle_c.c, gcc -O3 -march=x86-64-v3), %d threads over the spectral axis, "
                          "%.1f s wall" % (n_sample, cfg["S"], n_per_core, L, cores, dt),
                "anchor": cpu_anchor(cores)}
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    cores = min(cores, 64)
    n_per_core = max(1, n_per_core // 2)
    n_sample = cores * n_per_core
    idx = np.linspace(0, cfg["S"] - 1, n_sample).astype(int)
    shares = [idx[c::cores] for c in range(cores)]
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update({k: "1" for k in saved})
    try:
        with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as pool:
            list(pool.map(_cpu_worker, [(cfg, L, None)] * cores))           # start + import in every worker, untimed
            t0 = time.perf_counter()
            busy = list(pool.map(_cpu_worker, [(cfg, L, sh) for sh in shares]))
            dt = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {"value": n_sample / dt, "unit": "spectral-points/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d spectral points (evenly spaced, %d per core), all %d layers, m=0..2, numpy oracle (FP32), "
                      "%d single-threaded worker processes, %.1f s wall (%.1f s CPU)"
                      % (n_sample, cfg["S"], n_per_core, L, cores, dt, sum(busy))}


if __name__ == "__main__":
    main()
