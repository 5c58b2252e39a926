"""Multi-GPU sharding of a batch of IVPs: one process per GPU, no data-path collective.

The path shards trivially: trajectories are independent, so each rank integrates its
own slice with its own host controller. The only exchange is the final gather of per-trajectory
results, done with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU).
torch is imported lazily and only here: it is plumbing, not the product.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def interleaved_indices(total: int, world: int, rank: int):
    """Round-robin assignment (rank, rank + world, ...): balances tolerance sweeps, whose step counts
    grow monotonically along the batch."""
    return np.arange(rank, total, world)


def gather_results(local: np.ndarray, device=None) -> np.ndarray:
    """All-gather equally-shaped per-rank result arrays along axis 0; identity without a process group."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return local
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0).cpu().numpy()
