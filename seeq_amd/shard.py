"""Line sharding across GPUs.

Lines are independent (no state crosses seeqStringMatch calls: reference
libseeq.c:236-247 resets everything per line), so the path shards by
contiguous line ranges with NO data-path collective.  The only exchange is
the global count: one all-reduce(sum) of three u64 per scan (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests) plus, for global line
numbers, an exclusive prefix of the per-rank line counts.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous [first, first+count) of n_total items for `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def reduce_counts(counts, device=None, group=None):
    """Sum (nlines, nmatchlines, nhits) over ranks; returns a dict with the global values.
    No-op when torch.distributed is not initialised (single GPU)."""
    keys = ("nlines", "nmatchlines", "nhits")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: int(counts[k]) for k in keys}
    t = torch.tensor([int(counts[k]) for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def line_base(nlines_local, device=None, group=None):
    """Number of counted lines on lower ranks = global line number of this rank's line 0
    (needed to print global line numbers; records themselves stay rank-local)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([int(nlines_local)], dtype=torch.int64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    return int(sum(int(v.item()) for v in allv[:rank]))
