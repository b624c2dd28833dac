#!/usr/bin/env python3
"""Per-phase cycles of k_ia128 (FP64 interaction kernel for 64 < N <= 128), mean per spectral point over ALL workgroups.
Needs the diagnostic build (make -C vsmartmom.jl_amd/csrc timing):
  VSM_LIB_PATH=vsmartmom.jl_amd/lib_dbg/libvsmartmom_hip_timing.so python tools/phase_timing128.py [pol:l_trunc] [points] [random]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402

NAMES = ["vectors, stage [R+-], barrier", "load r-+ strip", "E2 = R+- r-+ (+ park)", "load t-- strip, Z = R+- t-- (+ park)",
         "barrier, stage [T--], barrier", "V = T-- t-- (+ park)", "S = T-- r-+ (+ park)", "fetch E2, norm, store [E2], series",
         "barrier, stage [t++], barrier", "T21 = t++ G2", "barrier, fetch S, store [S], barrier", "Y = S G2",
         "barrier, store [T21], barrier", "R+- = r+- + T21 Z (load, fetch, product, store)", "T++ = T21 T++ (load, product, store), barrier",
         "store [Y], barrier", "R-+ += Y T++ ; T-- = V + Y Z (loads, fetches, products, store R-+)", "store T--, end barrier"]


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "IQUV:51"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    pol, lt = case.split(":")
    H = vsm.host_model
    N = H.rt_set_streams(int(lt), 40.0, [30.0], H.polarization_type(pol), np.float64).Nquad * H.polarization_type(pol).n
    arch = vsm.Architectures.GPU(0)
    lib = C.CDLL(vsm._lib.LIB_PATH)
    if len(sys.argv) > 3 and sys.argv[3] == "random":   # the tests' random layers: large ||R+- r-+||, deep inverse
        import test_gpu_parity as T
        rng = np.random.default_rng(1)
        comp, add = T._random_layers(rng, N, S, np.float64, None)
        pc, pa = T._upload_layers(vsm, arch, comp, add, np.float64)
        run = lambda: vsm.CoreRT.interaction_("11", pc, pa)
    else:                                               # the O2-A atmosphere of tools/shape_cliff_timing.py, a whole forward run
        import bench
        tau_rayl, tau_abs = bench.o2a_atmosphere(S, 10)
        model = H.model_from_arrays(arch, pol, int(lt), 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279,
                                    albedo=0.15, m_max=2)
        scene = vsm.CoreRT.prepare_scene(model)
        run = lambda: scene.run()
    run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.vsm_debug_phase_cycles_128(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.vsm_debug_phase_cycles_128(buf, 0)
    v = np.array(list(buf), dtype=float)
    npts = max(v[31], 1)
    tot = v[:18].sum() / npts
    print("k_ia128 at N = %d, %d points (%d point-launches): timed region %.3f ms ; cycles per point (thread 0 of every workgroup), sum %.0f"
          % (N, S, int(npts), e0.elapsed_time(e1), tot))
    mf = 64.0 * 4 * ((N + 15) // 16) * ((N + 15) // 16)   # cycles of one wave's MFMAs per product
    for n, x in zip(NAMES, v[:18]):
        print("  %-78s %9.0f  %5.1f %%" % (n, x / npts, 100 * x / npts / tot))
    print("  (one product = %d MFMAs of 64 cycles per wave = %.0f cycles; two waves share a SIMD)" % (mf / 64, mf))
    if v[63] > 0:
        d = v[32:43]
        np_d = v[63]
        dn = ["load r, t, riders, park, store [r], barrier (per point)", "W = r t, riders (+ park W)", "E = r r (+ park r)",
              "norm, store [E], series", "fetch t, barrier, store [t], barrier", "tt = t G", "barrier, store [tt], barrier (+ fetches)",
              "r' = r + tt W", "t' = tt t", "riders, parks, barrier, store [r'], barrier", "outputs (apply_D), end barrier (per point)"]
        totd = d.sum() / np_d
        print("k_dbl128: cycles per point (all doubling steps), sum %.0f" % totd)
        for n, x in zip(dn, d):
            print("  %-78s %9.0f  %5.1f %%" % (n, x / np_d, 100 * x / np_d / totd))


if __name__ == "__main__":
    main()
