"""CPU tests of the index-probe oracle (restatement only: no runnable reference exists for this path).
They pin invariants the reference documents: a planted perfect read is found at its true site with
score == maxQuality and perfect=true; the product-side index builder produces the same arrays."""
import numpy as np

from bbmap_amd.index import HostIndex
from oracle.oracle import OracleIndex, make_offsets
from tests.index_problems import make_genome, make_reads, revcomp


def test_builder_matches_oracle_build():
    chroms = [make_genome(3, 60000), make_genome(4, 20000)]
    for k, cb in ((10, None), (9, 1)):
        hi = HostIndex(chroms, k=k, chromBits=cb)
        oi = OracleIndex(chroms, k=k, chromBits=cb)
        assert hi.nblocks == oi.s.nblocks
        for b in range(hi.nblocks):
            st, si = oi.block_arrays(b)
            assert np.array_equal(st, hi.starts[b]) and np.array_equal(si, hi.sites[b])
        assert np.array_equal(oi.counts(), hi.counts)
        assert np.array_equal(np.array(oi.s.lengthHistogram[:]), hi.length_histogram)
        assert {n: getattr(oi.s.p, n) for n in hi.params} == hi.params


def test_make_offsets_shape():
    offs = make_offsets(150, 13, 1.9)
    assert offs[0] == 0 and offs[-1] == 150 - 13 and len(offs) == 22
    assert all(b > a for a, b in zip(offs, offs[1:]))
    assert make_offsets(13, 13, 1.9) == [0]


def test_planted_perfect_reads_are_found_perfect():
    G = make_genome(11, 200000, repeats=False)
    ix = OracleIndex([G], k=11)
    import random
    rng = random.Random(2)
    offs = make_offsets(150, 11)
    for _ in range(100):
        st = rng.randrange(1000, len(G) - 1000)
        rd = G[st:st + 150]
        strand = rng.random() < 0.5
        bp = revcomp(rd) if strand else rd
        res = ix.find(bp, revcomp(bp), [0] * 150, [1100] * len(offs), offs)
        best = max(res, key=lambda r: r["score"])
        assert best["score"] == 70 + 149 * 100 and best["perfect"] == 1 and best["semiperfect"] == 1
        assert (best["chrom"], best["strand"], best["start"], best["stop"]) == (1, 1 if strand else 0, st, st + 149)
        assert best["hits"] == len(offs)


def test_mutated_reads_mostly_found_near_truth():
    genomes = [make_genome(21, 150000), make_genome(22, 80000)]
    ix = OracleIndex(genomes, k=11)
    reads = make_reads(5, genomes, 300, k=11)
    found = total = 0
    for bp, bm, bs, ks, offs, truth in reads:
        res = ix.find(bp, bm, bs, ks, offs)
        total += 1
        if any(r["chrom"] == truth[0] and r["strand"] == truth[1] and abs(r["start"] - truth[2]) <= 12 for r in res):
            found += 1
        for r in res:
            assert r["start"] <= r["stop"] and r["hits"] >= 1
            assert not r["perfect"] or r["semiperfect"]
    assert found > 0.8 * total
