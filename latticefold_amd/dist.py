"""Multi-GPU exchange for the column-sharded Ajtai commitment (BASELINE configs[3]).

Each rank holds A[:, shard] and the matching slice of every witness; `sharded_commit` computes the partial commitment on the
rank's GPU, exchanges the partials with ONE all-gather (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests) and adds
them mod p locally -- RCCL has no modular reduction and ncclSum on canonical u64 residues would wrap mod 2^64.
"""
import ctypes as C

import numpy as np

from . import api


def column_shard(n, rank, world):
    """contiguous column range of rank (high index bits: keeps sumcheck pairs (2j,2j+1) local)"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def allgather_modsum(partial, group=None, ring="goldilocks"):
    """partial: uint64 array of canonical residues (any shape) -> elementwise sum over ranks mod p (of `ring`)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    flat = np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1)
    t = torch.from_numpy(flat.view(np.int64).copy())            # bit pattern; torch has no uint64 collectives
    if backend == "nccl":
        t = t.cuda()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    stacked = torch.stack(parts).cpu().numpy().view(np.uint64)
    out = np.zeros_like(flat)
    rc = api._lib().lf_modsum_ring(stacked.ctypes.data_as(api.u64p), world, flat.size, out.ctypes.data_as(api.u64p), api.RING_IDS[ring])
    if rc != 0:
        raise api.LfError(rc, "lf_modsum_ring")
    return out.reshape(np.shape(partial))


def sharded_commit(scheme_shard, f_shard, group=None):
    """scheme_shard: api.AjtaiCommitmentScheme over this rank's column slice; f_shard: (n_local,24) or (batch,n_local,24)."""
    return allgather_modsum(scheme_shard.commit_ntt(f_shard), group, scheme_shard.ctx.ring if hasattr(scheme_shard.ctx, "ring") else "goldilocks")


def init_sharding(ctx, rank, world, transport="auto"):
    """Put `ctx` into intra-step sharding mode over the default torch.distributed process group.
    transport "rccl": the library's own RCCL communicators (lf_dist_init; device buffers, no Python in the data path) -- the ids are
    created on rank 0 and broadcast through torch.distributed; "host": torch.distributed all_gather through a Python callback, one
    process group per lane (gloo in the CPU/one-GPU tests); "auto": rccl when the backend is nccl, else host."""
    import torch
    import torch.distributed as dist
    if transport == "auto":
        transport = "rccl" if dist.get_backend() == "nccl" else "host"
    if transport == "rccl":
        ids = [None]
        if rank == 0:
            try:
                ids = [api.dist_unique_ids()]
            except api.LfError:          # RCCL not loadable on this box: every rank falls back to the host transport together
                ids = [b""]
        dist.broadcast_object_list(ids, src=0)
        if ids[0]:
            # lf_dist_init checks the two-lane schedule itself (both communicators' first collectives issued concurrently from the two threads of a fold step).
            # Every rank must run the SAME schedule: the ranks' outcomes are combined (minimum); a rank whose handshake timed out has lost its communicators,
            # so then every rank makes new ones from fresh ids, without the handshake, and the conservative one-thread schedule is used
            try:
                ctx.dist_init(rank, world, ids[0])
                mine = ctx.dist_two_lanes()
            except api.LfError:
                mine = -1
            t = torch.tensor([mine], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            agreed = int(t.item())
            if agreed < 0:
                import os
                ids = [api.dist_unique_ids() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                os.environ["LF_DIST_NO_HANDSHAKE"] = "1"
                ctx.dist_init(rank, world, ids[0])
                agreed = 0
            ctx.dist_two_lanes(agreed)
            return "rccl"
        transport = "host"
    if transport == "host":
        g0, g1 = dist.new_group(), dist.new_group()   # collective calls: every rank creates both groups in the same order
        ctx.set_sharding(rank, world, make_allgather(g0), make_allgather(g1))
        ctx._lf_groups = (g0, g1)
    return transport


def make_allgather(group=None):
    """all-gather of a uint64 vector for Context.set_sharding: RCCL on device tensors with backend "nccl", host tensors with gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)

    def allgather(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.uint64).view(np.int64).copy())
        if backend == "nccl":
            t = t.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        return torch.stack(parts).cpu().numpy().view(np.uint64)

    return allgather
