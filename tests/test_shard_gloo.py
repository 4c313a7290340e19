"""CPU: the N>1 path (line sharding + count all-reduce + line-number base) with world_size 2 over gloo.
On the GPU box the same code runs over RCCL; here each rank counts its shard with the oracle."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT

PAT, TAU, N, LEN = "GATGTAGCGCGATTAGCCTG", 3, 6001, 150


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    from seeq_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    first, count = shard.shard_range(N, rank, world)
    buf = orc.synth_reads(first, count, LEN, PAT, TAU)
    res = orc.buffer_scan(PAT, TAU, buf, 1)
    local = dict(nlines=res["nlines"], nmatchlines=res["nmatchlines"], nhits=len(res["records"]))
    total = shard.reduce_counts(local)
    base = shard.line_base(local["nlines"])
    out.put((rank, first, count, local, total, base))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from seeq_amd import shard
    for n in (0, 1, 7, 100, 6001):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and sum(c for _, c in r) == n
            assert all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(c for _, c in r) - min(c for _, c in r) <= 1


def test_two_rank_count_reduce(oracle):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = oracle.buffer_scan(PAT, TAU, oracle.synth_reads(0, N, LEN, PAT, TAU), 1)
    for rank, first, count, local, total, base in got:
        assert total == dict(nlines=N, nmatchlines=whole["nmatchlines"], nhits=len(whole["records"]))
        assert base == first                      # one line per read: line base == first read index
    assert sum(g[3]["nmatchlines"] for g in got) == whole["nmatchlines"]
