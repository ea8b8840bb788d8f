"""configs[1], configs[2] and configs[3] at their real reference sizes (E. coli-sized, 4.6 Mbp, single-ended; chr21-sized, 46.7 Mbp;
hg38-shaped, 24 chromosomes, 3.09 Gbp, four index blocks, 10 % repeat families; k=13; 150-bp reads, paired in the last two), checked through properties that do not need the oracle on the whole batch: a batch mapped twice gives the same bytes; a batch mapped
in two halves gives the lists the whole batch gives (pairs are independent, which is what lets reads shard across GPUs); the
planted pairs come back where they were drawn from; and a sample of the batch is identical to the CPU oracle, site by site and
fill by fill (the oracle probes the device-built index arrays, exported block by block, as bench.py does)."""
import numpy as np
import numpy.lib.recfunctions  # noqa: F401  (np.lib.recfunctions)
import pytest

import bench as B
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper
from oracle.oracle import map_batch
from tests.mapper_check import compare

pytestmark = pytest.mark.gpu
L, K, N = 150, 13, 120000            # reads (60,000 pairs)


def _map(di, reads, offs, paired=True):
    n = reads.size // L
    mp = Mapper(di, n, L, offs, [100 * K] * len(offs), paired=paired, max_sites=32)
    mp.load_reads(reads)
    mp.step()
    out, st = mp.fetch(), mp.stats()
    mp.step()
    again = mp.fetch(with_match=False)
    mp.close()
    return out, st, again


@pytest.mark.parametrize("name", ["ecoli", "chr21", "hg38"])
def test_full_size_reference_workload(name):
    lens, paired, _ = B.WORKLOADS[name]
    chroms, _ = B.shared_reference(name, lens, 0.0 if name == "ecoli" else 0.1, 0, 1)
    if paired and len(chroms) == 1:
        pairs, truth = W.make_pairs(chroms[0], N // 2, read_len=L, seed=3, lo=B.LEAD_N.get(name, [0])[0])     # (chr21: not from its leading N-run)
        truth["chrom"] = np.ones(N // 2, np.int64)
    else:
        pairs, truth = B.make_batch(chroms, N, paired, 3, with_truth=True)      # single-ended reads / pairs from all 24 chromosomes, mixed
    di = DeviceIndex.build(chroms, k=K)
    offs = W.make_offsets(L, K, 1.9)
    out, st, again = _map(di, pairs, offs, paired)
    ns = out["nsites"]
    assert st["reads_overflowed"] == 0 and (ns != -1).all()

    # 1. the same batch again: the same lists (fills land in the log in whatever order their threads get there, so the log
    #    position a site remembers is not part of the comparison; neither are the slots past a list's end)
    def norm(o):
        s_ = o["sites"].copy()
        s_["match_job"] = 0
        s_["reserved"] = 0
        s_[np.arange(s_.shape[1])[None, :] >= np.maximum(o["nsites"], 0)[:, None]] = np.zeros((), s_.dtype)
        return s_
    assert (again["nsites"] == ns).all() and norm(again).tobytes() == norm(out).tobytes()

    # 2. the two halves of the batch, mapped separately (reads the overflow tier mapped are compared through its own lists)
    def lists(o, r):
        if o["nsites"][r] == -3:
            t = o["overflow"]
            i = int(np.nonzero(t["read_ids"] == r)[0][0])
            return t["sites"][i, : t["nsites"][i]]
        return o["sites"][r, : max(0, o["nsites"][r])]
    half = (N // 4) * 2
    for lo, hi in ((0, half), (half, N)):
        part, _, _ = _map(di, pairs.reshape(-1, L)[lo:hi].reshape(-1), offs, paired)
        step = max(1, (hi - lo) // 4000)
        for r in list(range(0, hi - lo, step)) + [int(x) for x in np.nonzero(part["nsites"] == -3)[0][:50]]:
            a, b = lists(part, r), lists(out, lo + r)
            drop = ["match_job", "reserved"]                      # job numbers differ between the two runs' logs
            fa = np.lib.recfunctions.drop_fields(a, drop, usemask=False) if len(a) else a
            fb = np.lib.recfunctions.drop_fields(b, drop, usemask=False) if len(b) else b
            assert len(a) == len(b) and fa.tobytes() == fb.tobytes(), "read %d of the half starting at %d" % (r, lo)

    # 3. planted pairs come back where they were drawn from (one chromosome: the generator's coordinates are at hand), and
    #    nearly every mate ends with a site that passes the reference's minimum score ratio
    top = out["sites"][:, 0]
    minScore = int(np.float32(0.56) * np.float32(70 + (L - 1) * 100))
    assert ((ns > 0) & (top["slowScore"] >= minScore)).mean() > 0.99
    if paired:
        ok1 = (ns[0::2] > 0) & (top["chrom"][0::2] == truth["chrom"]) & (np.abs(top["start"][0::2] - truth["start1"]) <= 40) & (top["strand"][0::2] == truth["strand1"])
        ok2 = (ns[1::2] > 0) & (top["chrom"][1::2] == truth["chrom"]) & (np.abs(top["start"][1::2] - truth["start2"]) <= 40) & (top["strand"][1::2] == truth["strand2"])
        assert ok1.mean() > 0.97 and ok2.mean() > 0.97, (ok1.mean(), ok2.mean())
    else:
        ok = (ns > 0) & (top["chrom"] == truth["chrom"]) & (np.abs(top["start"] - truth["start1"]) <= 40) & (top["strand"] == truth["strand1"])
        assert ok.mean() > 0.97, ok.mean()
    # ... and in the records BBMap would print (the final alignment stage)
    fin = out["final"]
    if paired:
        okf = (fin["mapped"][0::2] > 0) & (fin["chrom"][0::2] == truth["chrom"]) & (np.abs(fin["start"][0::2] - truth["start1"]) <= 40) & (fin["strand"][0::2] == truth["strand1"])
    else:
        okf = (fin["mapped"] > 0) & (fin["chrom"] == truth["chrom"]) & (np.abs(fin["start"] - truth["start1"]) <= 40) & (fin["strand"] == truth["strand1"])
    assert okf.mean() > 0.97, okf.mean()

    # 4. a sample against the oracle
    oi = B.oracle_index(di, chroms, K)
    cnt = 400
    r = pairs.reshape(-1, L)
    if paired:
        oi.s.p.quitAfterTwoPerfects = 0
        orc = map_batch(oi, r[0:cnt:2].copy(), r[1:cnt:2].copy(), L, offs, [100 * K] * len(offs), cap=1024, match_stride=4200)
    else:
        orc = map_batch(oi, r[:cnt].copy(), None, L, offs, [100 * K] * len(offs), cap=1024, match_stride=4200)
    good = [i for i in range(cnt) if ns[i] >= 0 or ns[i] == -3]
    bad = compare(out, orc, cnt, paired, reads_range=good)
    assert not bad, "\n".join(bad[:10])
    di.close()
