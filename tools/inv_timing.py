#!/usr/bin/env python3
"""vsm_batch_inv in isolation: time per launch and per matrix for a sweep of N (matrices I - E with small E, the RT case)."""
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
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--sizes", default="30,60,64,96,128")
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    FT = np.float64 if a.dtype == "f64" else np.float32
    arch = vsm.Architectures.GPU(0)
    rng = np.random.default_rng(0)
    for N in [int(x) for x in a.sizes.split(",")]:
        A = (np.eye(N)[None] - 0.3 * rng.random((a.batch, N, N)) / N).astype(FT)
        tA = vsm.CoreRT.to_device_matrix(A, arch, FT)
        tX = torch.empty_like(tA)
        for _ in range(2):
            vsm.CoreRT.batch_inv_(tX, tA)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps):
            vsm.CoreRT.batch_inv_(tX, tA)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("N=%3d %s batch=%d: %.3f ms per launch, %.2f TFLOP/s (2 N^3 per matrix), %.2f us per pivot step and launch" % (
            N, a.dtype, a.batch, ms, 2.0 * N ** 3 * a.batch / ms / 1e9, ms * 1e3 / N), flush=True)


if __name__ == "__main__":
    main()
