#!/usr/bin/env python3
"""Thread scaling of the C + OpenMP CPU restatement on the box's host cores (C2 shape); prints points/s per thread count."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import vsm_oracle as O, vsm_oracle_c as OC

cfg, L = bench.CONFIGS["C2"], 40
tr, ta = bench.o2a_atmosphere(cfg["S"], L)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("no cpu.max", e)
for nt in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128, 256]:
    idx = np.linspace(0, cfg["S"] - 1, 2 * nt).astype(int)
    m = O.build_model(cfg["pol"], cfg["l_trunc"], 40.0, [30.0], [0.0], tau_rayl=tr[idx], tau_abs=ta[idx], depol=0.0279, albedo=0.15, m_max=2)
    t = time.perf_counter(); OC.rt_run(m, nthreads=nt); dt = time.perf_counter() - t
    print("threads %4d: %d points in %.2f s = %.1f points/s (%.2f per thread)" % (nt, len(idx), dt, len(idx) / dt, len(idx) / dt / nt), flush=True)
