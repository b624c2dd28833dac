"""Spectral-axis sharding of `rt_run` over the GPUs of one node.

The reference has no multi-device code; its only parallel axis is the
spectral one, and every operator of the elastic path is independent per spectral
point.  So: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI
on ROCm, "gloo" in the CPU tests), each rank owns a contiguous block of spectral
points, NO collective on the data path, and one gather of R/T at the end.

`ndoubl` and the scattering-interface tags are batch-global in the reference
(rt_kernel.jl:197,282-283; compEffectiveLayerProperties.jl:87).  `Scene` derives them
from the full spectral axis on every rank, so a sharded run reproduces the
single-device run exactly.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(S: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous blocks of ceil(S/world) points (the last ranks may get fewer / none)."""
    per = -(-S // world)
    lo = min(rank * per, S)
    return lo, min(lo + per, S)


def shard_slice(S: int, rank: int, world: int) -> slice:
    lo, hi = shard_bounds(S, rank, world)
    return slice(lo, hi)


def raman_halo_slices(S: int, owned: slice, shifts) -> Tuple[slice, slice]:
    """Raman (RRS) sharding: recipient point n1 reads elastic fields of the donor n0 = n1 + shift
    (src/Inelastic/inelastic_helper.jl:19-26), so a rank that owns `owned` computes on the slice extended by
    max|shift| points on each side (clipped to the band) and needs no exchange.  Shifts that can never be in band
    (|shift| >= S) do not widen the halo.  Returns (extended slice of the full axis, owned points inside it)."""
    lo, hi, _ = owned.indices(S)
    if hi <= lo:
        return slice(lo, lo), slice(0, 0)
    sh = [abs(int(x)) for x in np.asarray(shifts).ravel() if abs(int(x)) < S]
    h = max(sh) if sh else 0
    elo, ehi = max(0, lo - h), min(S, hi + h)
    return slice(elo, ehi), slice(lo - elo, hi - elo)


def init_process_group_from_env(backend: Optional[str] = None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the environment (torchrun)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if backend == "nccl":
            # create the RCCL communicator with every rank present: the data path's gather is a group of point-to-point
            # transfers in which a rank with an empty block takes no part
            dist.all_reduce(torch.zeros(1, device="cuda"))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def gather_spectral(local, S_total: int, rank: int, world: int, dst: int = 0):
    """Gather per-rank results whose FIRST axis is the local spectral block onto `dst` -- the ONE collective step of the data path.

    `local` is a torch tensor of shape (S_local, ...) (device tensor under nccl, CPU under gloo).  Returns the concatenated
    (S_total, ...) tensor on `dst`, None elsewhere.  The blocks are ragged when world does not divide S_total (the last ranks
    own fewer points, or none), so the step is one group of point-to-point transfers (ncclGroupStart/End under RCCL): every rank
    sends exactly its block, `dst` receives each block straight into its rows of the output -- no padding to a common block
    size, no concatenation copy."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    local = local.contiguous()
    if local.shape[0] != shard_bounds(S_total, rank, world)[1] - shard_bounds(S_total, rank, world)[0]:
        raise ValueError("gather_spectral: rank %d holds %d points, its block has %d" % (
            rank, local.shape[0], shard_bounds(S_total, rank, world)[1] - shard_bounds(S_total, rank, world)[0]))
    if rank != dst:
        if local.shape[0] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst)]):
                w.wait()
        return None
    out = torch.empty((S_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    ops = []
    for r in range(world):
        lo, hi = shard_bounds(S_total, r, world)
        if hi <= lo:
            continue
        if r == dst:
            out[lo:hi] = local
        else:
            ops.append(dist.P2POp(dist.irecv, out[lo:hi], r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


def pack_RT(R, T):
    """R, T (S_local, nStokes, nVZA) -> ONE buffer (S_local, 2 nStokes nVZA) per rank, so that the gather is one step."""
    import torch
    flat = lambda t: t.reshape(t.shape[0], int(np.prod(t.shape[1:])))     # (an empty block keeps its column count)
    return torch.cat([flat(R), flat(T)], dim=1)


def unpack_RT(g, n: int, nV: int):
    """The gathered buffer (S, 2 n nV) -> host arrays R, T in the reference's axis order [nVZA, nStokes, nSpec]."""
    g = g.detach().cpu().numpy()
    S, k = g.shape[0], n * nV
    return (g[:, :k].reshape(S, n, nV).transpose(2, 1, 0).copy(), g[:, k:2 * k].reshape(S, n, nV).transpose(2, 1, 0).copy())


def rt_run_sharded(model, executor: Optional[Callable] = None, rank: int = 0, world: int = 1, dst: int = 0):
    """rt_run over this rank's spectral block + ONE gather of the packed R | T on `dst` (what bench.make_step times).

    `executor(model, spec_slice) -> (R, T)` returns tensors shaped (S_local, nStokes, nVZA); the default
    is the HIP engine (`CoreRT.prepare_scene(model, slice).run()`).  Returns host arrays
    [nVZA, nStokes, nSpec] on `dst`, (None, None) on the other ranks."""
    S = model.tau_rayl.shape[0]
    sl = shard_slice(S, rank, world)
    if executor is None:
        from . import core_rt

        def executor(mdl, s):
            scene = core_rt.prepare_scene(mdl, s)
            return scene.run()
    R, T = executor(model, sl)
    n, nV = int(R.shape[1]), int(R.shape[2])
    g = gather_spectral(pack_RT(R, T), S, rank, world, dst)
    if rank != dst:
        return None, None
    return unpack_RT(g, n, nV)


def rt_run_lin_sharded(model, lin_model, NAer: int, NGas: int, NSurf: int, executor: Optional[Callable] = None, rank: int = 0,
                       world: int = 1, dst: int = 0):
    """rt_run(model, lin_model, NAer, NGas, NSurf) over this rank's spectral block + ONE gather of R, T, Rdot, Tdot on `dst`
    (rt_run_lin.jl:185-190,324: Rdot/Tdot [nVZA, nStokes, nSpec, Nparams]).

    `executor(model, lin_model, spec_slice) -> (R, T, Rd, Td)` returns tensors whose spectral axis is the local block:
    R/T (S_local, nStokes, nVZA), Rd/Td (P, S_local, nStokes, nVZA); the default is the HIP engine
    (`CoreRTLin.SceneLin(model, lin_model, ..., slice).run()`).  The four arrays travel in one buffer per rank
    ([S_local, (2 + 2P) nStokes nVZA]), so the data path has exactly one collective.  Returns host arrays in the
    reference's axis order on `dst`, Nones elsewhere."""
    import torch
    S = model.tau_rayl.shape[0]
    sl = shard_slice(S, rank, world)
    if executor is None:
        from . import core_rt_lin

        def executor(mdl, lin, s):
            return core_rt_lin.SceneLin(mdl, lin, NAer, NGas, NSurf, s).run()
    R, T, Rd, Td = executor(model, lin_model, sl)
    P = Rd.shape[0]
    Sl, n, nV = R.shape
    packed = torch.cat([R.reshape(Sl, n * nV), T.reshape(Sl, n * nV), Rd.permute(1, 0, 2, 3).reshape(Sl, P * n * nV),
                        Td.permute(1, 0, 2, 3).reshape(Sl, P * n * nV)], dim=1).contiguous()
    g = gather_spectral(packed, S, rank, world, dst)
    if rank != dst:
        return None, None, None, None
    g = g.detach().cpu().numpy()
    k = n * nV
    Rg, Tg = g[:, :k].reshape(S, n, nV), g[:, k:2 * k].reshape(S, n, nV)
    Rdg = g[:, 2 * k:(2 + P) * k].reshape(S, P, n, nV)
    Tdg = g[:, (2 + P) * k:].reshape(S, P, n, nV)
    return (Rg.transpose(2, 1, 0).copy(), Tg.transpose(2, 1, 0).copy(), Rdg.transpose(3, 2, 0, 1).copy(),
            Tdg.transpose(3, 2, 0, 1).copy())
