#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_raman_doubling_wave (workgroup 0 / wave 0), from a -DRW_PHASE_TIMING build:
   tools/variantsrw.sh phase "-DRW_PHASE_TIMING" ; python tools/raman_phase_timing.py
Diagnostic tool, not part of the product."""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VSM_LIB_PATH", os.path.join(ROOT, "vsmartmom.jl_amd", "lib_dbg", "libv_phase.so"))

NAMES = ["stage ier/iet/r0", "X, r1 iet (72 MFMA)", "stage gt0, X image", "X gt0, iet gt0 (48)", "stage gr0", "X gr0 (24)",
         "stage grt0", "iet grt0 (24)", "rider algebra, WA image", "stage t0", "W3, ttg W1, ttg W3 (72)", "outputs", "", "", "",
         "loop head"]


def main():
    import torch
    torch.zeros(1, device="cuda")
    lib = C.CDLL(os.environ["VSM_LIB_PATH"])
    sys.argv = [sys.argv[0], "--points", "1024"]
    lib.vsm_debug_rw_phase(None, 1)
    runpy.run_path(os.path.join(ROOT, "tools", "raman_timing.py"), run_name="__main__")
    buf = (C.c_ulonglong * 16)()
    lib.vsm_debug_rw_phase(buf, 0)
    tot = sum(buf)
    for i, n in enumerate(NAMES):
        if n:
            print("%-28s %12d  %5.1f %%" % (n, buf[i], 100.0 * buf[i] / max(tot, 1)))
    print("total stamped cycles (s_memtime units): %d" % tot)


if __name__ == "__main__":
    main()
