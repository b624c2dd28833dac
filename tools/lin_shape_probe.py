"""Linearized run of one large FP64 shape (N = 112) for profiling the operator chain of 64 < N <= 128: python tools/lin_shape_probe.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, bench
import vsmartmom_jl_amd as vsm
arch = vsm.Architectures.GPU(0)
S, L = 2000, 10
tau_rayl, tau_abs = bench.o2a_atmosphere(S, L)
H = vsm.host_model
model = H.model_from_arrays(arch, "IQUV", 51, 40.0, [30.0], [0.0], tau_rayl=tau_rayl, tau_abs=tau_abs, depol=0.0279, albedo=0.15, m_max=2)
ls = vsm.CoreRTLin.SceneLin(model, H.LinModel([tau_abs]), 0, 1, 1)
ls.run(); torch.cuda.synchronize()
t0 = time.perf_counter(); ls.run(); torch.cuda.synchronize(); print("lin N=112", S / (time.perf_counter() - t0), "points/s")
