#!/usr/bin/env python3
"""vsm_batched_mul in isolation: time per launch, TFLOP/s and HBM GB/s (three N x N operands per product) over a sweep of N."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vsmartmom_jl_amd as vsm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2000)
    ap.add_argument("--sizes", default="30,48,60,64,72,96,112,128")
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    FT = np.float64 if a.dtype == "f64" else np.float32
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    for N in [int(x) for x in a.sizes.split(",")]:
        A = torch.randn((a.batch, N, N), dtype=dt, device="cuda")
        B = torch.randn((a.batch, N, N), dtype=dt, device="cuda")
        for _ in range(2):
            Cm = vsm.CoreRT.batched_mul(A, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            Cm = vsm.CoreRT.batched_mul(A, B)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byts = 3.0 * N * N * A.element_size() * a.batch
        print("N=%3d %s batch=%d: %.1f us per launch (incl. the allocation of C), %.1f TFLOP/s, %.2f TB/s" % (
            N, a.dtype, a.batch, ms * 1e3, 2.0 * N ** 3 * a.batch / ms / 1e9, byts / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
