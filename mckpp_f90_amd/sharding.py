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


class AgreementError(RuntimeError):
    """The ranks could not even agree on whether a block of collectives may run: the job must end."""


def agree(dist, ok, device=None, timeout_s=120.0):
    """Collective: True when EVERY rank calls it with ok=True.  An all_reduce(MAX) of a failed flag, waited for with a
    timeout; if the reduction itself fails or does not complete, AgreementError - nothing that follows could be
    matched up with the other ranks any more."""
    import datetime

    import torch

    t = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=device if device is not None else "cpu")
    try:
        work = dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True)
        done = work.wait(timeout=datetime.timedelta(seconds=timeout_s))
        if done is False:
            raise AgreementError(f"all_reduce of the failure flag did not complete within {timeout_s:.0f} s")
        return float(t.item()) == 0.0
    except AgreementError:
        raise
    except Exception as e:   # noqa: BLE001
        raise AgreementError(f"all_reduce of the failure flag failed: {type(e).__name__}: {e}") from e


def guarded_block(dist, local, collective, device=None, timeout_s=120.0):
    """A block of a bench line that must not be able to take the line down, made of a part that only touches this
    rank (`local()`: downloads, kernel runs - it may fail on one rank alone) and a part of collectives
    (`collective(x)`, x what local returned - a failure there reaches every rank, or none ever returns).  Every rank
    learns whether ANY rank's local part failed before the first collective is entered (one that skipped them would
    leave the others inside a gather until the process group's timeout), and again after the collectives.  Returns
    (result, None) or (None, error text); raises AgreementError when the ranks cannot agree (the caller exits
    non-zero: better no line than a hung job)."""
    err, x = None, None
    try:
        x = local()
    except Exception as e:   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(dist, err is None, device, timeout_s):
        return None, err or "another rank failed in the local part of this block (collectives skipped on every rank)"
    out = None
    try:
        out = collective(x)
    except Exception as e:   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not agree(dist, err is None, device, timeout_s):
        return None, err or "another rank failed in the collectives of this block"
    return out, None
