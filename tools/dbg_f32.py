import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import vsmartmom_jl_amd as vsm
from oracle import vsm_oracle as O
from oracle import vsm_oracle_lin as OL
arch = vsm.Architectures.GPU(0)
CR, CL = vsm.CoreRT, vsm.CoreRTLin
FT = np.float32 if sys.argv[1] == "f32" else np.float64
for N in [int(x) for x in sys.argv[2:]]:
    rng = np.random.default_rng(3)
    S, P = 3, 2
    refl = lambda sc, lead=(S,): (sc * rng.random(lead + (N, N)) / N).astype(FT)
    trans = lambda lead=(S,): (np.eye(N) * rng.uniform(0.3, 0.95, lead + (N, 1)) + 0.05 * rng.random(lead + (N, N)) / N).astype(FT)
    conv_v = vsm.Architectures.array_type(arch)
    cm = lambda x: CR.to_device_matrix(x, arch, FT)
    pc, pa = CR.CompositeLayer(FT, arch, N, S), CR.AddedLayer(FT, arch, N, S)
    pcl, pal = CL.CompositeLayerLin(FT, arch, P, N, S), CL.AddedLayerLin(FT, arch, P, N, S)
    for k in ("R_mp", "R_pm", "T_pp", "T_mm"):
        getattr(pc, k).copy_(cm(refl(1.0) if k[0] == "R" else trans()))
        getattr(pcl, k).copy_(conv_v((0.1 * rng.standard_normal((P, S, N, N))).astype(FT)))
    for k in ("r_mp", "t_pp", "r_pm", "t_mm"):
        getattr(pa, k).copy_(cm(refl(1.0) if k[0] == "r" else trans()))
        getattr(pal, "ap_" + k).copy_(conv_v((0.1 * rng.standard_normal((P, S, N, N))).astype(FT)))
    print("N", N, "interaction_lin f32 ...", flush=True)
    CL.interaction_lin_("11", pc, pcl, pa, pal)
    torch.cuda.synchronize()
    print("  ok", float(pc.R_mp.abs().max()), flush=True)
    pol = vsm.host_model.polarization_type("I")
    expk = conv_v(np.full(S, 0.9, dtype=FT))
    dall = conv_v(np.zeros((P, S), dtype=FT))
    print("N", N, "doubling_lin f32 ...", flush=True)
    CL.doubling_allparams_(pol, expk, 2, pa, pal, dall, 0.7, P)
    torch.cuda.synchronize()
    print("  ok", float(pa.r_mp.abs().max()), flush=True)
