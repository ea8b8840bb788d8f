"""Two processes, each mapping its shard of a pair list with the device mapper (both on the one GPU of the test box, gloo for the
rendezvous): the concatenated outputs equal the single-process result.  This is the multi-GPU data path in miniature: pairs are
independent, every rank holds its own replica of the index, nothing is exchanged but the barrier and the step time."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
L, K, PAIRS = 150, 12, 600


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _lists(out):
    """per read: (nsites, records) with the overflow tier's lists folded in (nsites == -3 in the main list points there)"""
    tier = out.get("overflow")
    where = {}
    if tier is not None:
        where = {int(r): i for i, r in enumerate(tier["read_ids"])}
    res = []
    for r in range(len(out["nsites"])):
        n = int(out["nsites"][r])
        if n == -3:
            i = where[r]
            n = int(tier["nsites"][i])
            res.append((n, tier["sites"][i][:max(0, n)].copy()))
        else:
            res.append((n, out["sites"][r][:max(0, n)].copy()))
    return res


def _map(reads, ref, max_sites=32):
    from bbmap_amd.index import DeviceIndex
    from bbmap_amd.mapper import Mapper
    from bbmap_amd import workload as W
    di = DeviceIndex.build([ref], k=K)
    offs = W.make_offsets(L, K, 1.9)
    mp_ = Mapper(di, reads.size // L, L, offs, [100 * K] * len(offs), paired=True, max_sites=max_sites)
    mp_.load_reads(reads)
    mp_.step()
    out = mp_.fetch(with_match=False)
    st = mp_.stats()
    mp_.close()
    di.close()
    out["stats"] = st
    return out


def _data(repeats=False):
    from bbmap_amd import workload as W
    if repeats:        # 60 % of the sequence in 3 repeat families: many reads have more candidate sites than slots
        ref = W.make_reference(200000, seed=11, pad=2000, repeat_frac=0.6, families=3)
        reads, _ = W.make_pairs(ref, PAIRS, read_len=L, seed=10, pad=2000, hard_frac=0.05)
        return ref, reads
    ref = W.make_reference(150000, seed=31, pad=2000, repeat_frac=0.1)
    reads, _ = W.make_pairs(ref, PAIRS, read_len=L, seed=9, pad=2000, hard_frac=0.1)
    return ref, reads


def _worker(rank, world, port, q, max_sites=32):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bbmap_amd import dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, reads = _data(repeats=max_sites != 32)
    lo, hi = D.shard_range(PAIRS, rank, world)                       # this rank's pairs
    out = _map(reads.reshape(-1, 2 * L)[lo:hi].reshape(-1), ref, max_sites)
    dist.barrier()
    if max_sites == 32:
        q.put((rank, lo, hi, out["nsites"].copy(), out["sites"].copy()))
    else:
        q.put((rank, lo, hi, _lists(out), int(out["stats"]["reads_reprobed"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda g: g[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref, reads = _data()
    whole = _map(reads, ref)
    nsites = np.concatenate([g[3] for g in got])
    sites = np.concatenate([g[4] for g in got])
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == (0, PAIRS // 2, PAIRS // 2, PAIRS)
    assert (nsites == whole["nsites"]).all()
    for f in sites.dtype.names:
        if f not in ("match_job", "reserved"):                      # match_job indexes a rank's own fill log
            for r in range(len(nsites)):
                n = max(0, int(nsites[r]))
                assert (sites[f][r, :n] == whole["sites"][f][r, :n]).all(), (f, r)


def test_two_ranks_equal_one_process_with_the_overflow_tier():
    """The same with 4 slots per read, so that on every rank some lists outgrow their slots and are mapped by that rank's overflow
    tier: the lists a host would read (main list, or the tier's through read_ids) are the single process's, whichever rank and
    whichever tier produced them."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, 4)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda g: g[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref, reads = _data(repeats=True)
    whole = _map(reads, ref, 4)
    assert all(g[4] > 0 for g in got) and whole["stats"]["reads_reprobed"] == sum(g[4] for g in got), ([g[4] for g in got], whole["stats"])     # every rank's tier had work
    lists = got[0][3] + got[1][3]
    want = _lists(whole)
    assert len(lists) == len(want) == 2 * PAIRS
    for r, ((n, s), (wn, ws)) in enumerate(zip(lists, want)):
        assert n == wn, r
        for f in s.dtype.names:
            if f not in ("match_job", "reserved"):
                assert (s[f] == ws[f]).all(), (f, r)
