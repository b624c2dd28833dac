#!/usr/bin/env python3
"""Where a workgroup of k_ia_native<4,15> spends its cycles: needs the diagnostic build
(make -C vsmartmom.jl_amd/csrc ia_phases -> lib_dbg/libvsm_ia_phases.so).  Not part of the product.
    VSM_LIB_PATH=$PWD/vsmartmom.jl_amd/lib_dbg/libvsm_ia_phases.so python tools/ia_phases.py [--points S] [--dsym 3] [--refl 0.1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402

NAMES = ["images of R+- / T-- read, [R+-] stored, riders, barrier (a)", "[E2 | Z] = R+- [r-+ | t--]", "store [T--], norm (barrier b)",
         "store [E2], barrier (c), inverse of order 7 in four products", "[S | V] = T-- [r-+ | t--], D r D",
         "barrier (d), store [S] [t++], barrier (e)", "T21 = t++ G2, Y = S G2", "barrier (f), image of T++ through P, store [T21] [Y], barrier (g)",
         "[R+- | T++] products", "R-+ requested, stores R+- T++ issued", "[R-+ | T--] products (R-+ arrives underneath)",
         "stores R-+ T-- issued and drained (the wait is the stamp's)", "kernel head: vectors, images of r-+ t++ by DMA, images of R+- T-- requested"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--dsym", type=int, default=3)
    ap.add_argument("--refl", type=float, default=0.1)
    a = ap.parse_args()
    N, S = 60, a.points
    arch = vsm.Architectures.GPU(0)
    CR = vsm.CoreRT
    dev = torch.device("cuda:0")
    pc = CR.make_composite_layer(np.float64, arch, (N, N), S)
    pa = CR.make_added_layer(np.float64, arch, (N, N), S, d_symmetric=a.dsym)

    def refl(scale):
        return torch.rand((S, N, N), dtype=torch.float64, device=dev) * (scale / N)

    def trans():
        d = 0.3 + 0.65 * torch.rand((S, N), dtype=torch.float64, device=dev)
        return torch.diag_embed(d) + torch.rand((S, N, N), dtype=torch.float64, device=dev) * (0.05 / N)
    init = dict(R_mp=refl(a.refl), R_pm=refl(a.refl), T_pp=trans(), T_mm=trans(), J0_p=torch.rand((S, N), dtype=torch.float64, device=dev),
                J0_m=torch.rand((S, N), dtype=torch.float64, device=dev))
    for k in ("r_mp", "r_pm")[:1 if a.dsym else 2]:
        getattr(pa, k).copy_(refl(0.75 * a.refl))
    for k in ("t_pp", "t_mm")[:1 if a.dsym else 2]:
        getattr(pa, k).copy_(trans())
    lib = C.CDLL(vsm._lib.LIB_PATH)
    buf = (C.c_ulonglong * 16)()
    for it in range(3):
        for k, v in init.items():
            getattr(pc, k).copy_(v)
        torch.cuda.synchronize()
        lib.vsm_debug_ia_phases(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        CR.interaction_("11", pc, pa)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    lib.vsm_debug_ia_phases(buf, 0)
    v = np.array(list(buf), dtype=float)
    nwg = max(v[15], 1.0)
    tot = v[:13].sum() / nwg
    print("k_ia_native<4,15,%s> S=%d: %.4f ms per launch; s_memtime ticks (shader clock) of wave 0 per workgroup, mean over %d workgroups"
          % ("true" if a.dsym else "false", S, ms, nwg))
    for i in [12] + list(range(12)):
        print("  %-62s %9.1f  %5.1f %%" % (NAMES[i], v[i] / nwg, 100 * v[i] / nwg / tot))
    print("  total %.1f ticks per workgroup" % tot)


if __name__ == "__main__":
    main()
