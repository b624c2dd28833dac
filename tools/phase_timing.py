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
                                             depol=0.0279, albedo=0.15, m_max=int(os.environ.get("VSM_PT_MMAX", "2")))
    scene = vsm.CoreRT.prepare_scene(model)
    scene.run()
    torch.cuda.synchronize()
    lib = C.CDLL(vsm._lib.LIB_PATH)
    buf = (C.c_ulonglong * 32)()
    lib.vsm_debug_phase_cycles(None, 1)
    lib.vsm_debug_phase_cycles_strip(None, 1)
    import time
    t0 = time.perf_counter()
    scene.run()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    print("wall time of the pass (%d layer launches + surface): %.3f ms" % (L, wall_ms))
    lib.vsm_debug_phase_cycles(buf, 0)
    lib.vsm_debug_phase_cycles_strip(buf, 0)
    names = ["rider set-up, first A-form stores (per launch)", "[E | W] = r [r | t], rider swap-add", "norm, store [E], Horner series",
             "tt = t G, reload t, barrier, store [tt], barrier", "[r' | t'] = [r | 0] + tt [W | t]", "riders", "barrier, store [r] [t], barrier"]
    allb = list(buf)
    v = np.array(allb[:7], dtype=float)
    nwg = max(allb[28], 1)                      # workgroups that ran elemental + doubling (every workgroup flushes its stamps)
    nd = scene.moments[0]["layers"][0]["nd"]
    print("S=%d layers=%d nd=%d ; cycles per workgroup of k_layer_strip, mean over %d workgroups (all rounds of all launches)" % (S, L, nd, nwg))
    for i, (n, x) in enumerate(zip(names, v)):
        per = x / nwg / (nd if i else 1)
        print("  %-52s %10.1f per %s" % (n, per, "step" if i else "launch"))
    print("  doubling loop per step: %.0f" % (v[1:].sum() / nwg / nd))
    el = [allb[20] / nwg, allb[21] / nwg, allb[22] / nwg, allb[23] / nwg, allb[7] / nwg]
    print("  elemental: tables %.0f, barrier %.0f, elements + sources -> A-forms %.0f, barrier %.0f, strips back %.0f  (sum %.0f per launch)"
          % (el[0], el[1], el[2], el[3], el[4], sum(el)))
    if allb[31]:
        print("  core clock during elemental + doubling (s_memtime ticks per s_memrealtime tick at 100 MHz): %.0f MHz"
              % (allb[30] / (allb[31] / 100.0)))
    inames = ["stage [R+-], [T--]", "[E2|Z] = R+- [r|t--], [S|V] = T-- [r|t--], z", "norm, store [S] [E2], series", "barrier, store [t], barrier",
              "T21 = t G2, Y = S G2", "barrier, stores [T21] [Y], request T++ / R-+, barrier", "[R+-|T++] = [r+-|0] + T21 [Z|T++]",
              "stores R+-, T++, J0+", "[R-+|T--] += Y [T++|Z]", "stores R-+, T--, J0-"]
    w = np.array(allb[8:18], dtype=float)
    ni = max(allb[29], 1)
    print("interaction (mean over %d workgroups):" % ni)
    for n, x in zip(inames, w):
        print("  %-56s %10.1f" % (n, x / ni))
    print("  sum: %.0f" % (w.sum() / ni))
    tot = v.sum() / nwg + w.sum() / ni + sum(el)
    if allb[26]:
        print("  workgroup lifetime (kernel entry -> last store issued): %.0f cycles, mean over %d non-TOA workgroups" % (allb[27] / allb[26], allb[26]))
    print("  layer total per workgroup: %.0f cycles (MFMA issue of %d doubling steps x 6 products + 11: %d)" % (tot, nd, (6 * nd + 11) * 3840))


if __name__ == "__main__":
    main()
