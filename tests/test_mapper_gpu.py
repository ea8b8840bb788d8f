"""GPU tests of the device-resident mapper (bbmap_*): single-ended processRead and paired processReadPair (pairing, trimming,
ungapped scores, tip deletions, scoreSlow rounds with the exact minScore sequence and wider refills, rescue) against the CPU
restatement oracle/mapper_oracle.c -- site lists field by field, every fill's window / minScore / score vector / visited-cell
count / traceback string, and which fill each site took its limits from."""
import ctypes as C

import numpy as np
import pytest

from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper
from oracle import oracle as O
from tests.mapper_check import compare, gpu_fills

pytestmark = pytest.mark.gpu


def _run(ref, reads, L, k, paired, max_sites=32, cap=64, **cfg):
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    n = reads.size // L
    mp = Mapper(di, n, L, offs, ks, paired=paired, max_sites=max_sites, **cfg)
    mp.load_reads(reads)
    mp.step()
    out = mp.fetch()
    st = mp.stats()
    oi = O.OracleIndex([ref], k=k)
    params = O.map_default_params(**{k_: v for k_, v in cfg.items() if k_ in ("tipSearchDist", "trimList", "doRescue", "averagePairDist")})
    if paired:
        oi.s.p.quitAfterTwoPerfects = 0
        r = reads.reshape(-1, L)
        orc = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), L, offs, ks, params=params, cap=cap, match_stride=4200)
    else:
        orc = O.map_batch(oi, reads, None, L, offs, ks, params=params, cap=cap, match_stride=4200)
    mp.close()
    di.close()
    return out, orc, st, n


def test_single_ended_matches_oracle():
    L, k = 150, 12
    ref = W.make_reference(300000, seed=5, pad=2000, repeat_frac=0.15)
    reads, _, _ = W.make_reads_and_jobs(ref, 3000, read_len=L, seed=9, pad=2000, long_del_frac=0.3, hard_frac=0.05)
    out, orc, st, n = _run(ref, reads, L, k, paired=False)
    assert st["reads_overflowed"] == 0
    bad = compare(out, orc, n, paired=False)
    assert not bad, "\n".join(bad[:20])
    assert st["fills"] > 0.1 * n and st["gapped_fills"] > 0        # the workload reaches both fill logs
    assert st["rounds"] >= 2                                         # and some reads need more than one fill


def test_paired_matches_oracle_with_rescue():
    L, k = 150, 12
    ref = W.make_reference(300000, seed=6, pad=2000, repeat_frac=0.15)
    reads, truth = W.make_pairs(ref, 2000, read_len=L, seed=4, pad=2000, hard_frac=0.08)
    out, orc, st, n = _run(ref, reads, L, k, paired=True)
    assert st["reads_overflowed"] == 0
    bad = compare(out, orc, n, paired=True)
    assert not bad, "\n".join(bad[:20])
    assert st["rescue_scans"] > 20 and st["rescue_fills"] > 5       # rescue really ran
    resc = sum(int(out["sites"][r, : max(0, out["nsites"][r])]["rescued"].sum()) for r in range(n))
    assert resc > 5
    # planted pairs come back where they were drawn from
    top = out["sites"][:, 0]
    ok1 = (out["nsites"][0::2] > 0) & (np.abs(top["start"][0::2] - truth["start1"]) <= 40) & (top["strand"][0::2] == truth["strand1"])
    assert ok1.mean() > 0.97


def test_paired_without_tip_search_and_trimming():
    L, k = 100, 11
    ref = W.make_reference(120000, seed=8, pad=1000, repeat_frac=0.3)
    reads, _ = W.make_pairs(ref, 800, read_len=L, seed=5, pad=1000, hard_frac=0.1)
    out, orc, st, n = _run(ref, reads, L, k, paired=True, tipSearchDist=0, trimList=0)
    assert st["reads_overflowed"] == 0
    bad = compare(out, orc, n, paired=True)
    assert not bad, "\n".join(bad[:20])


def _repeat_workload(paired):
    L, k = 150, 12
    ref = W.make_reference(200000, seed=11, pad=2000, repeat_frac=0.6, families=3)
    if paired:
        reads, _ = W.make_pairs(ref, 700, read_len=L, seed=2, pad=2000, hard_frac=0.05)
    else:
        reads, _, _ = W.make_reads_and_jobs(ref, 1500, read_len=L, seed=2, pad=2000)
    return ref, reads, L, k


def test_overflow_is_reported_not_dropped():
    """Without the overflow tier a read with more candidate sites than max_sites is flagged (nsites = -1, counted), never
    passed off as unmapped."""
    ref, reads, L, k = _repeat_workload(False)
    out, orc, st, n = _run(ref, reads, L, k, paired=False, max_sites=4, cap=256, reserved=(C.c_int32 * 4)(0, -1, 0, 0))
    over = out["nsites"] < 0
    assert over.sum() == st["reads_overflowed"] and over.sum() > 0 and st["reads_reprobed"] == 0 and "overflow" not in out
    # every read the device did map is identical to the oracle; the oracle finds sites for the flagged ones
    good = [r for r in range(n) if not over[r]]
    bad = compare(out, orc, n, paired=False, reads_range=good)
    assert not bad, "\n".join(bad[:20])
    assert all(orc["nsites1"][r] != 0 for r in np.nonzero(over)[0])


@pytest.mark.parametrize("paired", [False, True])
def test_overflow_tier_maps_what_max_sites_cannot_hold(paired):
    """The reference's site list has no capacity (BBIndex.java:1537-1604): reads whose list does not fit max_sites are mapped
    again by the overflow tier (long lists), pairs as pairs, and the result is the oracle's for EVERY read."""
    ref, reads, L, k = _repeat_workload(paired)
    out, orc, st, n = _run(ref, reads, L, k, paired=paired, max_sites=4, cap=1024)
    moved = out["nsites"] == -3
    assert st["reads_overflowed"] == 0 and (out["nsites"] >= -3).all() and not (out["nsites"] == -1).any()
    assert moved.sum() == st["reads_reprobed"] > (20 if paired else 50)
    t = out["overflow"]
    assert sorted(t["read_ids"].tolist()) == np.nonzero(moved)[0].tolist()
    assert int(t["nsites"].max()) > 4                                    # lists longer than the main capacity exist
    if paired:                                                           # mates travel together
        assert (t["read_ids"][0::2] % 2 == 0).all() and (t["read_ids"][1::2] == t["read_ids"][0::2] + 1).all()
    bad = compare(out, orc, n, paired=paired)
    assert not bad, "\n".join(bad[:20])


def test_overflow_tier_too_small_keeps_the_rest_flagged():
    ref, reads, L, k = _repeat_workload(False)
    out, orc, st, n = _run(ref, reads, L, k, paired=False, max_sites=4, cap=1024, reserved=(C.c_int32 * 4)(0, 10, 0, 0))
    moved, left = out["nsites"] == -3, out["nsites"] == -1
    assert moved.sum() == 10 == st["reads_reprobed"] and left.sum() == st["reads_overflowed"] > 0
    assert np.nonzero(moved)[0].max() < np.nonzero(left)[0].min()       # the tier takes the first ones in read order
    bad = compare(out, orc, n, paired=False, reads_range=[r for r in range(n) if not left[r]])
    assert not bad, "\n".join(bad[:20])


def test_fills_ahead_of_time_do_not_change_anything():
    """The same batch with and without fills ahead of time (bbmap_config.reserved[0]): identical sites and identical fill
    sequences; the strict mode needs one round per candidate site, the default a handful."""
    L, k = 150, 12
    ref = W.make_reference(200000, seed=12, pad=2000, repeat_frac=0.5, families=12)
    reads, _, _ = W.make_reads_and_jobs(ref, 1500, read_len=L, seed=3, pad=2000, hard_frac=0.3)
    res = {}
    for strict in (0, 1):
        di = DeviceIndex.build([ref], k=k)
        offs = O.make_offsets(L, k, 1.9)
        ks = [100 * k] * len(offs)
        mp = Mapper(di, 1500, L, offs, ks, paired=False, max_sites=48, reserved=(C.c_int32 * 4)(strict, 0, 0, 0))
        mp.load_reads(reads)
        mp.step()
        res[strict] = (mp.fetch(), mp.stats())
        mp.close()
        di.close()
    (a, sa), (b, sb) = res[0], res[1]
    assert sb["fills_dropped"] == 0 and sa["rounds"] < sb["rounds"] and sb["rounds"] >= 6
    assert (a["nsites"] == b["nsites"]).all()
    for f in a["sites"].dtype.names:
        if f not in ("match_job", "reserved"):
            assert (a["sites"][f] == b["sites"][f]).all(), f
    fa, fb = gpu_fills(a), gpu_fills(b)
    assert fa.keys() == fb.keys()
    for key in fa:
        for f in ("kind", "refStartLoc", "refEndLoc", "minScore", "score", "iterations", "match"):
            assert fa[key][f] == fb[key][f], (key, f)


def test_packed_site_lists_equal_the_padded_ones():
    """bbmap_pack_sites_device: the site lists without their empty slots, as a host would copy them back."""
    import torch
    ref, reads, L, k = _repeat_workload(True)                 # lists of every length, reads without a list, reads in the tier
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    n = reads.size // L
    mp = Mapper(di, n, L, offs, [100 * k] * len(offs), paired=True, max_sites=16)
    mp.load_reads(reads)
    mp.step()
    out = mp.fetch(with_match=False)
    dev = mp.dev
    counts = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    packed = torch.zeros(n * 16 * 128, dtype=torch.uint8, device=dev)
    mp.pack_sites(counts, offsets, packed)
    torch.cuda.synchronize()
    c, o = counts.cpu().numpy(), offsets.cpu().numpy()
    ns = np.maximum(out["nsites"], 0)
    assert (c[:n] == ns).all() and c[n] == 0
    assert (o == np.concatenate([[0], np.cumsum(ns)])).all() and o[n] > n // 2 and ns.max() > 4 and (out["nsites"] < 0).any()
    from bbmap_amd.mapper import MSITE_DTYPE
    recs = packed.cpu().numpy()[: int(o[n]) * 128].view(MSITE_DTYPE)
    want = np.concatenate([out["sites"][r, : ns[r]] for r in range(n)])
    assert recs.tobytes() == want.tobytes()
    # a destination that is too small is never written past its end (the caller compares offsets[n] with its capacity)
    small = torch.zeros(100 * 128 + 256, dtype=torch.uint8, device=dev)
    mp.pack_sites(counts, offsets, small[: 100 * 128])
    torch.cuda.synchronize()
    assert int(small[100 * 128:].sum()) == 0 and small.cpu().numpy()[: 100 * 128].tobytes() == want[:100].tobytes()
    mp.close()
    di.close()


@pytest.mark.parametrize("paired", [False, True])
def test_fill_logs_grow_instead_of_refusing_the_batch(paired):
    """The reference's lists of fills have no capacity.  With logs that start at 64 entries (jobsPerRead = -64) the rounds run
    into the end of both logs: the reads that find no room ask again after the host has grown the log, rescue
    reserves its entries beforehand, and every site list, fill and traceback string still equals the oracle's."""
    L, k = 150, 12
    ref = W.make_reference(300000, seed=5, pad=2000, repeat_frac=0.15)
    if paired:
        reads, _ = W.make_pairs(ref, 1500, read_len=L, seed=4, pad=2000, hard_frac=0.08)
    else:
        reads, _, _ = W.make_reads_and_jobs(ref, 3000, read_len=L, seed=9, pad=2000, long_del_frac=0.3, hard_frac=0.05)
    out, orc, st, n = _run(ref, reads, L, k, paired=paired, jobsPerRead=-64)
    assert st["log_growths"] >= 1 and st["fills"] > 64 and st["gapped_fills"] > (0 if paired else 64)
    assert st["reads_overflowed"] == 0
    bad = compare(out, orc, n, paired=paired)
    assert not bad, "\n".join(bad[:20])


def test_host_buffer_entry_returns_packed_lists_with_the_overflow_tier():
    """bbmap_map_batch (what a host without device memory calls, e.g. the JNI glue): the same lists as the device-resident call, packed,
    with the overflow tier's lists appended behind the others -- and a too small array reports how many records there are."""
    ref, reads, L, k = _repeat_workload(True)
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    n = reads.size // L
    mp = Mapper(di, n, L, offs, ks, paired=True, max_sites=4)
    mp.load_reads(reads)
    mp.step()
    out, st = mp.fetch(with_match=False), mp.stats()
    assert st["reads_reprobed"] > 20 and st["reads_overflowed"] == 0
    recs = np.zeros(n, np.dtype([("bases_off", "<i8"), ("keys_off", "<i8"), ("len", "<i4"), ("nkeys", "<i4")]))
    recs["bases_off"] = np.arange(n, dtype=np.int64) * L
    recs["len"] = L
    recs["nkeys"] = len(offs)
    keyinfo = np.array(list(offs) + list(ks), np.int32)
    ns, po, sites, total = mp.map_batch_host(recs, reads, np.zeros(reads.size, np.int8), keyinfo, 64 * n)
    assert total == len(sites) and (ns >= 0).all()
    tier = out["overflow"]
    where = {int(r): i for i, r in enumerate(tier["read_ids"])}
    for r in range(n):
        m = int(out["nsites"][r])
        want = out["sites"][r][:max(m, 0)] if m != -3 else tier["sites"][where[r]][:int(tier["nsites"][where[r]])]
        got = sites[po[r]:po[r] + ns[r]]
        assert len(got) == len(want), r
        for f in want.dtype.names:
            if f not in ("match_job", "reserved"):
                assert (got[f] == want[f]).all(), (r, f)
    assert int(ns.max()) > 4                                   # (a list longer than the main capacity came through)
    ns2, po2, sites2, total2 = mp.map_batch_host(recs, reads, np.zeros(reads.size, np.int8), keyinfo, 10)
    assert total2 == total and len(sites2) == 10 and (ns2 == ns).all()
    mp.close()
    di.close()


def test_average_pair_dist_follows_the_host_between_batches():
    """AVERAGE_PAIR_DIST moves while BBMap runs (DYNAMIC_INSERT_LENGTH, BBMapThread.java:1307-1309); the host carries the running value
    and sets it between batches (bbmap_set_average_pair_dist).  Two batches on ONE context, the second with another value: each equals
    the oracle run with that value, and the value matters (paired scores differ between the two)."""
    L, k = 150, 12
    ref = W.make_reference(300000, seed=6, pad=2000, repeat_frac=0.15)
    reads, _ = W.make_pairs(ref, 1500, read_len=L, seed=4, pad=2000, hard_frac=0.08)
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    n = reads.size // L
    mp = Mapper(di, n, L, offs, ks, paired=True, max_sites=32)
    mp.load_reads(reads)
    oi = O.OracleIndex([ref], k=k)
    oi.s.p.quitAfterTwoPerfects = 0
    r = reads.reshape(-1, L)
    outs = []
    for apd in (100, 260):
        mp.set_average_pair_dist(apd)
        mp.step()
        out = mp.fetch()
        orc = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), L, offs, ks, params=O.map_default_params(averagePairDist=apd), cap=64, match_stride=4200)
        bad = compare(out, orc, n, paired=True)
        assert not bad, "averagePairDist %d\n" % apd + "\n".join(bad[:20])
        outs.append(out)
    assert (outs[0]["sites"]["pairedScore"] != outs[1]["sites"]["pairedScore"]).any()
    mp.close()
    di.close()


def test_contexts_of_several_threads_share_one_index():
    """BBMap's thread model: every mapping thread has its own read lists and, here, its own bbmap_ctx; the index is one.  Four contexts
    on ONE DeviceIndex, stepped at the same time from four threads (each on a stream of its own), must give what each gives alone --
    site lists, fills and final records.  (Until round 4 the persistent probe waves' work queue belonged to the index context: two
    probes in flight took reads from each other's queue and half the reads came back unmapped.)"""
    import threading

    import torch
    L, k, T = 150, 12, 4
    ref = W.make_reference(400000, seed=11, pad=2000, repeat_frac=0.15)
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    mps, alone = [], []
    for t in range(T):
        reads, _ = W.make_pairs(ref, 1500, read_len=L, seed=20 + t, pad=2000, hard_frac=0.08)
        mp = Mapper(di, reads.size // L, L, offs, ks, paired=True, max_sites=32)
        mp.load_reads(reads)
        mp.step()
        o = mp.fetch()
        alone.append((o["nsites"].copy(), o["sites"].copy(), o["final"].copy(), len(o["jobs"]), len(o["gjobs"])))
        mps.append(mp)
    assert sum(int((a[2]["mapped"] > 0).sum()) for a in alone) > 0.95 * T * 3000
    streams = [torch.cuda.Stream() for _ in mps]
    errs = []

    def run(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(3):
                    mps[i].step()
        except Exception as e:                                   # (a failure inside a thread must fail the test)
            errs.append(repr(e))
    th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i, mp in enumerate(mps):
        o = mp.fetch()
        ns, st, fin, nj, ng = alone[i]
        assert np.array_equal(o["nsites"], ns), i
        for f in ("chrom", "strand", "start", "stop", "score", "slowScore", "pairedScore", "perfect", "rescued"):
            assert np.array_equal(o["sites"][f], st[f]), (i, f)
        for f in fin.dtype.names:
            if f not in ("match_off", "reserved"):
                assert np.array_equal(o["final"][f], fin[f]), (i, f)
        assert (len(o["jobs"]), len(o["gjobs"])) == (nj, ng)
        mp.close()
    di.close()
