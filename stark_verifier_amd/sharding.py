"""Batch sharding across the GPUs of one node (one process per GPU).

The hot path partitions: polynomials, branches and proofs are independent units (the
reference verifies branches one by one in a loop, src/merkle_tree.rs:35-41, and has no state
shared between proofs, src/lib.rs:99), so every rank works on a contiguous slice with no
data-path collective.  The only exchange is the final gather of per-unit accept flags.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_accept_bits(local_accept, total: int, device=None):
    """All ranks contribute their slice of accept flags; every rank gets the full [total] uint8 vector.

    Uses torch.distributed (RCCL when the process group is "nccl", gloo on CPU).  With a single
    process (no initialised process group) it returns the input unchanged.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(local_accept, dtype=np.uint8)
    world, rank = dist.get_world_size(), dist.get_rank()
    if isinstance(local_accept, torch.Tensor):
        t = local_accept.to(torch.uint8)
    else:
        t = torch.from_numpy(np.ascontiguousarray(local_accept, dtype=np.uint8))
    if device is not None:
        t = t.to(device)
    # all_gather wants equal sizes: pad every slice to the largest one
    width = (total + world - 1) // world
    padded = torch.zeros(width, dtype=torch.uint8, device=t.device)
    padded[: t.numel()] = t
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = np.empty(total, dtype=np.uint8)
    for r, p in enumerate(parts):
        lo, hi = shard_range(total, r, world)
        out[lo:hi] = p[: hi - lo].cpu().numpy()
    return out
