#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_elemental_doubling (needs the -DVSM_PHASE_TIMING build,
vsmartmom.jl_amd/lib_dbg/libvsmartmom_hip_timing.so).  Diagnostic tool, not part of the product."""
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

hip = C.CDLL("libamdhip64.so")


def read_stamps(lib_path, n=32):
    # hipGetSymbolAddress needs the symbol handle; use hipModule-less route: dlsym gives the host shadow var
    lib = C.CDLL(lib_path)
    host_sym = C.c_void_p.in_dll(lib, "_ZN3vsm16vsm_phase_cyclesE") if False else None
    return None


def main():
    S, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 4
    arch = vsm.Architectures.GPU(0)
    tau_rayl, tau_abs = bench.o2a_atmosphere(S, 40)
    tau_rayl, tau_abs = tau_rayl[:, :L], tau_abs[:, :L]
    model = vsm.host_model.model_from_arrays(arch, "IQU", 35, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs,
                                             depol=0.0279, albedo=0.15, m_max=0)
    scene = vsm.CoreRT.prepare_scene(model)
    scene.run()
    torch.cuda.synchronize()
    lib = C.CDLL(vsm._lib.LIB_PATH)
    buf = (C.c_ulonglong * 32)()
    lib.vsm_debug_phase_cycles(None, 1)
    lib.vsm_debug_phase_cycles_strip(None, 1)
    scene.run()
    torch.cuda.synchronize()
    lib.vsm_debug_phase_cycles(buf, 0)
    lib.vsm_debug_phase_cycles_strip(buf, 0)
    names = ["elemental + rider set-up (per launch)", "[E | W] = r [r | t], rider swap-add", "norm, store [E], Horner series",
             "tt = t G, reload t, barrier, store [tt], barrier", "[r' | t'] = [r | 0] + tt [W | t]", "riders", "barrier, store [r] [t], barrier"]
    v = np.array(list(buf)[:7], dtype=float)
    launches = L
    nd = scene.moments[0]["layers"][0]["nd"]
    print("S=%d layers=%d nd=%d ; cycles of workgroup 0 / thread 0 of k_layer_strip" % (S, L, nd))
    for i, (n, x) in enumerate(zip(names, v)):
        per = x / launches / (nd if i else 1)
        print("  %-44s %12.0f total  %10.1f per %s" % (n, x, per, "step" if i else "launch"))
    print("  doubling loop per step: %.0f" % (v[1:].sum() / launches / nd))
    allb = list(buf)
    if allb[31]:
        print("  core clock during the doubling loops of workgroup 0 (s_memtime ticks per s_memrealtime tick at 100 MHz): %.0f MHz"
              % (allb[30] / (allb[31] / 100.0)))
    inames = ["stage [r],[T--], strips", "E1 = r R+-, u", "G1 (series) + store", "H, T01, T01 r + stores", "R-+ update (global)",
              "T-- = T01 t--, J0- (global)", "stage [R+-], [t]", "G2, z, T21 + store", "T21 T++, T21 R+- (global)", "R+- (global)"]
    lib.vsm_debug_phase_cycles_strip(buf, 0)
    w = np.array(list(buf)[8:18], dtype=float)
    ni = L  # L-1 layer interactions + 1 surface
    print("k_ia_strip (per launch, %d launches; two workgroups share the CU, so waits include the other one's work):" % ni)
    for n, x in zip(inames, w):
        print("  %-36s %10.1f" % (n, x / ni))
    print("  sum per launch: %.0f" % (w.sum() / ni))


if __name__ == "__main__":
    main()
