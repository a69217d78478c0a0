"""Column sharding across GPUs and the diagnostics gather.

The column step has no cross-column dependence (reference
src/mckpp_physics_driver_mod.F90:46-63), so ranks own disjoint column sets and
never exchange data inside a step.  Columns are dealt round-robin so that the
data-dependent iteration counts (6..200 passes) average out per rank.  The only
collective is the gather-to-root of diagnostics at output cadence (over RCCL on
GPUs; the same code runs over gloo in the CPU tests).
"""
import numpy as np


def shard_indices(ntotal, rank, world):
    """Global column indices owned by `rank` (round-robin)."""
    return np.arange(rank, ntotal, world)


def unshard(parts, ntotal):
    """Inverse of round-robin sharding: parts[r] holds rank r's columns (leading axis)."""
    world = len(parts)
    first = np.asarray(parts[0])
    out = np.empty((ntotal,) + first.shape[1:], dtype=first.dtype)
    for r, p in enumerate(parts):
        out[r::world] = np.asarray(p)
    return out


def gather_to_root(local, dist, device=None, dst=0):
    """Gather equally- or unequally-sized per-rank arrays to `dst` with
    torch.distributed; returns the list of per-rank numpy arrays on dst, else None."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[: int(s.item())].cpu().numpy() for b, s in zip(bufs, sizes)]
