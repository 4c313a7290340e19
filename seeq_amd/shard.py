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


def reduce_counts(counts, device=None, group=None, force=False):
    """Sum (nlines, nmatchlines, nhits) over ranks; returns a dict with the global values.
    No-op when torch.distributed is not initialised (single GPU) or the group has one rank -- unless `force`
    (the collective then runs over the one rank: the RCCL rehearsal of bench.py --force-dist)."""
    keys = ("nlines", "nmatchlines", "nhits")
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return {k: int(counts[k]) for k in keys}
    t = torch.tensor([int(counts[k]) for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def line_base(nlines_local, device=None, group=None, force=False):
    """Number of counted lines on lower ranks = global line number of this rank's line 0
    (needed to print global line numbers; records themselves stay rank-local)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([int(nlines_local)], dtype=torch.int64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    return int(sum(int(v.item()) for v in allv[:rank]))


# ---------------------------------------------------------------------------------------------------------------
# Real files: byte ranges cut at '\n', per-shard results merged into whole-file results
# ---------------------------------------------------------------------------------------------------------------
def cut_at_newlines(buf, world):
    """`world` contiguous byte ranges [lo, hi) of `buf` (bytes / bytearray / 1-D uint8 array) that partition it and
    end right after a '\\n' (the last one at the end of the buffer): every line belongs to exactly one shard, whole --
    the per-line contract of the reference loop (seeq.c:361-377: getline, then seeqStringMatch on that line alone,
    libseeq.c:236-247 resets all matcher state per call).  Shards may be empty when the buffer has fewer lines."""
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8) if not hasattr(buf, "dtype") else buf
    n = int(a.shape[0])
    cuts = [0]
    for r in range(1, world):
        want = max(cuts[-1], (n * r) // world)
        # the first newline at or after the even cut (search in growing windows: lines are short or very long)
        pos, step = want, 1 << 16
        nl = -1
        while pos < n:
            hit = np.flatnonzero(a[pos:pos + step] == 10)
            if hit.size:
                nl = pos + int(hit[0])
                break
            pos += step
            step *= 4
        cuts.append(n if nl < 0 else nl + 1)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def merge_results(parts):
    """parts: per shard, in shard order, dicts {nlines, nmatchlines, nhits, records (k x 4: line, start, end, dist;
    line 1-based within the shard, FASTA headers not counted)}.  Returns the whole buffer's dict: counts summed,
    records concatenated with every shard's lines renumbered behind the counted lines of the shards before it
    (reference seeq.c:377: sqfile->line counts non-header lines from the start of the file)."""
    import numpy as np
    out = dict(nlines=0, nmatchlines=0, nhits=0)
    recs = []
    for p in parts:
        r = np.asarray(p.get("records", np.zeros((0, 4), dtype=np.uint64)), dtype=np.uint64).reshape(-1, 4).copy()
        r[:, 0] += out["nlines"]
        recs.append(r)
        for k in ("nlines", "nmatchlines", "nhits"):
            out[k] += int(p[k])
    out["records"] = np.concatenate(recs, axis=0) if recs else np.zeros((0, 4), dtype=np.uint64)
    return out
