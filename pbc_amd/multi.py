"""Multi-GPU use of the batched pairing path: a contiguous range split of the batch, one
process per GPU, no data-path collective.  torch.distributed is used only to
bring the disjoint result slices back to rank 0 ("host-side gather") and for timing barriers.
"""
import numpy as np


def range_split(n, world):
    """Contiguous ceil(n/world) ranges; products are never split across ranks."""
    per = -(-n // world)
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world)]


def element_pairing_sharded(compute, g1, g2, k=1, group=None):
    """Every rank passes the same full input arrays (n*k records each); rank r computes units
    [lo_r, hi_r) with `compute(g1_slice, g2_slice)` (normally Pairing.element_pairing or a
    k-term Pairing.element_prod_pairing bound to the rank's GPU) and rank 0 returns the
    concatenated (n, len_GT) result, other ranks None."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    l1 = g1.shape[-1]
    n = g1.reshape(-1, l1).shape[0] // k
    lo, hi = range_split(n, world)[rank]
    part = compute(g1.reshape(-1, l1)[lo * k:hi * k], g2.reshape(-1, g2.shape[-1])[lo * k:hi * k])
    parts = [None] * world if rank == 0 else None
    dist.gather_object(np.ascontiguousarray(part), parts, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([p for p in parts if len(p)], axis=0) if n else part


def max_over_ranks(seconds, device=None, group=None):
    """The job's step time is the slowest rank's."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
