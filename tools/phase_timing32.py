#!/usr/bin/env python3
"""Per-phase cycle breakdown of the FP32 strip layer kernel (k_layer_strip32) on a C4-shaped problem (needs the
-DVSM_PHASE_TIMING build: make -C vsmartmom.jl_amd/csrc timing).  Diagnostic tool."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vsmartmom_jl_amd as vsm  # noqa: E402

vsm._lib.LIB_PATH = os.path.join(ROOT, "vsmartmom.jl_amd", "lib_dbg", "libvsmartmom_hip_timing.so")
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    S, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 4
    arch = vsm.Architectures.GPU(0)
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, 60)
    model = vsm.host_model.model_from_arrays(arch, "IQU", 59, 40.0, [30.0], [0.0], tau_rayl=tau_rayl[:, :L], tau_abs=tau_abs[:, :L],
                                             depol=0.0279, albedo=0.15, m_max=0, float_type=np.float32)
    scene = vsm.CoreRT.prepare_scene(model)
    scene.run()
    torch.cuda.synchronize()
    lib = C.CDLL(vsm._lib.LIB_PATH)
    buf = (C.c_ulonglong * 32)()
    lib.vsm_debug_phase_cycles_strip32(None, 1)
    scene.run()
    torch.cuda.synchronize()
    lib.vsm_debug_phase_cycles_strip32(buf, 0)
    v = np.array(list(buf), dtype=float)
    nd = scene.moments[0]["layers"][0]["nd"]
    names = {0: "elemental", 1: "E = r r", 2: "norm + inverse", 3: "tt = t G, store, 2 barriers", 4: "matvec tt j", 5: "tmp, t' (shared A), store, barrier",
             6: "matvec tmp j", 7: "r' product, barrier", 8: "j update, stores, barrier",
             10: "IA stage", 11: "u matvec, E1", 12: "G1 + store", 13: "H, T01, T01 r + stores", 14: "J0- matvec, load T++ R-+",
             15: "R-+ product + store", 16: "T-- product + store, barrier", 17: "stage [R+-], [t]", 18: "G2, z, T21, store",
             19: "J0+ matvec, T++ / tmp products, stores", 20: "R+- product + store"}
    print("N=96 FP32 S=%d layers=%d nd=%d: cycles of workgroup 0 / thread 0" % (S, L, nd))
    for i in sorted(names):
        per = v[i] / L / (nd if 1 <= i <= 8 else 1)
        print("  %-46s %10.0f per %s" % (names[i], per, "step" if 1 <= i <= 8 else "launch"))
    print("  doubling per launch %.0f ; interaction per launch %.0f" % (v[1:9].sum() / L, v[10:21].sum() / (L - 1)))


if __name__ == "__main__":
    main()
