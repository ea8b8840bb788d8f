"""Multi-GPU plumbing: reads shard across ranks, the index is replicated, no data-path collective.

The reference's mapping threads pull read lists from one shared queue and never exchange data
(current/align2/AbstractMapThread.java:390,574); one process per GPU does the same with a static split.
torch.distributed is only used for the start barrier and for the max-over-ranks step time.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_seed(base_seed, rank):
    """Read-generator seed of a rank (weak scaling: every rank draws its own, equally sized, stream)."""
    return base_seed + 1000 * rank


def shard_range(n_total, rank, world):
    """Contiguous slice [lo, hi) of a fixed read list owned by a rank (strong-scaling helper)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def max_over_ranks(value, dist, device=None):
    """The slowest rank's time: what a whole-job throughput must be divided by."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
