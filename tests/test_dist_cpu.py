"""world_size-2 gloo test of the sharding plumbing (runs on CPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from bbmap_amd import dist as D
    from bbmap_amd import workload as W
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert D.env_rank() == (rank, world, rank)
    ref = W.make_reference(50000, seed=1, pad=500)
    reads, jobs, truth = W.make_reads_and_jobs(ref, 2000, seed=D.shard_seed(2, rank), pad=500)
    dist.barrier()
    slowest = D.max_over_ranks(1.0 + rank, dist)           # rank r "took" 1+r seconds
    total = D.sum_over_ranks(len(jobs), dist)
    lo, hi = D.shard_range(1001, rank, world)
    q.put((rank, slowest, total, int(truth["start"][:50].sum()), lo, hi))
    dist.destroy_process_group()


def test_two_ranks_shard_reads_and_reduce_time():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [2.0, 2.0]               # max over ranks
    assert [g[2] for g in got] == [4000.0, 4000.0]         # units summed over ranks
    assert got[0][3] != got[1][3]                          # different read streams per rank
    assert (got[0][4], got[0][5], got[1][4], got[1][5]) == (0, 500, 500, 1001)   # disjoint, covering slices
