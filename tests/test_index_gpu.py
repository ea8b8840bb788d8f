"""GPU parity tests: the HIP index probe (through the C ABI) vs the CPU oracle, SiteScore by SiteScore."""
import os

import pytest

from bbmap_amd.index import HostIndex, DeviceIndex
from oracle.oracle import OracleIndex
from tests.index_problems import make_genome, make_reads

pytestmark = pytest.mark.gpu


def run_case(genomes, k, chromBits, reads, tweak=None, cap=48):
    """Both probe kernels (one read per wavefront; one read per lane) against the oracle, SiteScore by SiteScore."""
    hi = HostIndex(genomes, k=k, chromBits=chromBits)
    oi = OracleIndex(genomes, k=k, chromBits=chromBits)
    if tweak:
        for name, val in tweak.items():
            hi.params[name] = val
            setattr(oi.s.p, name, val)
    di = DeviceIndex(hi)
    exp = []
    for bp, bm, bs, ks, offs, truth in reads:
        try:
            exp.append(oi.find(bp, bm, bs, ks, offs, cap=cap))
        except RuntimeError:                       # more than `cap` sites: the kernels must report the overflow too
            exp.append(None)
    # the wavefront kernel exists in two variants (with / without batched pops and bulk skips, picked by average list
    # length); BBIDX_LONG_LISTS forces one, so every case runs through both of them and through the per-lane kernel
    # ... and with the LDS sizing for short reads (reads longer than the announced 160 bases must come back through the
    # per-lane kernel, unchanged)
    for kind, variant, maxlen in (("auto", "1", 600), ("auto", "0", 600), ("auto", "1", 160), ("auto", "0", 160), ("lane", "", 600)):
        di.set_kernel(kind)
        di.set_max_read_len(maxlen)
        os.environ["BBIDX_LONG_LISTS"] = variant
        try:
            got = di.find_batch([(bp, bs, ks, offs) for bp, bm, bs, ks, offs, t in reads], max_sites=cap)
        finally:
            os.environ.pop("BBIDX_LONG_LISTS", None)
        for i, (bp, bm, bs, ks, offs, truth) in enumerate(reads):
            assert got[i] == exp[i], "%s kernel (long-list variant %r, max read length %d), read %d (truth %s): %s != %s" % (
                kind, variant, maxlen, i, truth, got[i], exp[i])
    di.close()
    return sum(bool(e) for e in exp)


def test_single_chromosome_k13():
    genomes = [make_genome(31, 250000)]
    reads = make_reads(7, genomes, 600, k=13)
    assert run_case(genomes, 13, None, reads) > 400


def test_two_blocks_small_k():
    # chromBits=1 -> two chromosomes per block: three chromosomes span two blocks
    genomes = [make_genome(41, 90000), make_genome(42, 60000), make_genome(43, 50000)]
    reads = make_reads(8, genomes, 500, k=10)
    assert run_case(genomes, 10, 1, reads) > 300


def test_short_reads_few_keys_and_no_prescan():
    genomes = [make_genome(51, 120000)]
    reads = make_reads(9, genomes, 300, read_len=40, k=11, density=1.5)
    run_case(genomes, 11, None, reads)
    run_case(genomes, 11, None, reads[:150], tweak={"prescanQscore": 0, "quitAfterTwoPerfects": 0})


def test_long_reads():
    genomes = [make_genome(61, 200000)]
    reads = make_reads(10, genomes, 150, read_len=400, k=13)
    run_case(genomes, 13, None, reads)


def test_kfilter_and_many_keys():
    # kfilter > 1 exercises the contig bookkeeping of calcAffineScore; density 3 gives > 64 keys for 400-bp reads,
    # which the wavefront kernel hands to the per-lane kernel
    genomes = [make_genome(71, 150000)]
    reads = make_reads(11, genomes, 300, k=12)
    run_case(genomes, 12, None, reads, tweak={"kfilter": 20})
    long_reads = make_reads(12, genomes, 60, read_len=400, k=13, density=3.0)
    assert max(len(r[4]) for r in long_reads) > 64
    run_case(genomes, 13, None, long_reads)


def test_dense_repeats_long_lists():
    # a genome made mostly of diverged repeat copies: long k-mer lists, many candidate sites per read, greedy trimming
    import random
    rng = random.Random(5)
    fam = [bytes(rng.choice(b"ACGT") for _ in range(500)) for _ in range(4)]
    body = bytearray()
    while len(body) < 120000:
        cp = bytearray(rng.choice(fam))
        for _ in range(rng.randint(0, 25)):
            cp[rng.randrange(len(cp))] = rng.choice(b"ACGT")
        body += cp + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 300)))
    genomes = [b"N" * 400 + bytes(body) + b"N" * 400]
    reads = make_reads(13, genomes, 400, k=11)
    run_case(genomes, 11, None, reads, cap=160)
    run_case(genomes, 11, None, reads[:200], tweak={"maxUsableLength": 4000, "maxUsableLength2": 8000}, cap=24)


def test_device_build_matches_host_builder():
    """bbidx_build (emit -> radix sort -> scan -> COUNTS / clumpy / histogram / tunables on the device) against the numpy
    builder array by array, then probes through the device-built index against the oracle."""
    import numpy as np
    genomes = [make_genome(81, 120000), make_genome(82, 70000), make_genome(83, 40000)]
    for k, cb in ((12, None), (10, 1)):
        hi = HostIndex(genomes, k=k, chromBits=cb, backend="numpy")
        di = DeviceIndex.build(genomes, k=k, chromBits=cb)
        assert di.host.chromBits == hi.chromBits and di.host.nblocks == hi.nblocks
        for b in range(hi.nblocks):
            starts, sites, counts, hist = di.export_block(b)
            assert np.array_equal(starts, hi.starts[b])
            assert np.array_equal(sites, hi.sites[b])
            assert np.array_equal(counts, hi.counts)
            assert np.array_equal(hist, hi.length_histogram)
        for name, val in hi.params.items():
            assert di.host.params[name] == int(val), name
        oi = OracleIndex(genomes, k=k, chromBits=cb)
        reads = make_reads(14, genomes, 200, k=k)
        got = di.find_batch([(bp, bs, ks, offs) for bp, bm, bs, ks, offs, t in reads], max_sites=48)
        for (bp, bm, bs, ks, offs, t), g in zip(reads, got):
            assert g == oi.find(bp, bm, bs, ks, offs, cap=48)
        di.close()


def test_long_lists_small_k_two_blocks():
    # the regime of a large genome in miniature: k=9 over 3 x 1.2 Mbp (262,144 keys, ~14 sites per key and block) in two
    # blocks, so every read's 22+ lists hold hundreds of interleaved entries: batched pops, bulk skips and the exit rules
    # of the heap walk all run for real, through every kernel variant
    import random
    rng = random.Random(77)
    genomes = [b"N" * 400 + bytes(rng.choice(b"ACGT") for _ in range(1200000)) + b"N" * 400 for _ in range(3)]
    reads = make_reads(21, genomes, 120, k=9)
    tweak = {"maxUsableLength": 400, "maxUsableLength2": 800}      # keep the long lists (analyzeIndex would drop most of them)
    assert run_case(genomes, 9, 1, reads, tweak=tweak, cap=64) > 40


def _hard_reads(seed, genomes, n, k, read_len=150, sub=(0.08, 0.16)):
    """Reads riddled with substitutions (a few keys hit, so the walk starts at a hit cutoff of 1 and every list entry is a
    site), some with Ns, some hanging over either end of their chromosome, with and without base / key scores."""
    import random
    from oracle.oracle import make_offsets
    from tests.index_problems import revcomp
    rng = random.Random(seed)
    out = []
    for i in range(n):
        ci = rng.randrange(len(genomes))
        G = genomes[ci]
        L = read_len
        where = rng.random()
        if where < 0.1:
            st = rng.randrange(380, 420)                       # around the first defined base
        elif where < 0.2:
            st = len(G) - L - rng.randrange(380, 420)          # around the last one
        else:
            st = rng.randrange(400, len(G) - L - 400)
        rd = bytearray(G[st:st + L])
        for p in range(L):
            if rng.random() < rng.uniform(*sub):
                rd[p] = rng.choice(b"ACGT")
        if rng.random() < 0.2:
            rd[rng.randrange(L)] = ord("N")
        rd = bytes(rd)
        if b"N" * 20 in rd:
            continue
        strand = 1 if rng.random() < 0.5 else 0
        bp = revcomp(rd) if strand else rd
        offs = make_offsets(L, k, 1.9)
        if rng.random() < 0.5:
            ks, bs = [100 * k] * len(offs), [0] * L
        else:
            ks, bs = [rng.randint(100 * k // 8, 100 * k) for _ in offs], [rng.randint(0, 30) for _ in range(L)]
        out.append((bp, revcomp(bp), bs, ks, offs, (ci + 1, strand, st)))
    return out


def test_few_hit_keys_walk_at_cutoff_one():
    # reads with 8-16 % substitutions: a handful of keys hit, the hit cutoff is 1, so the walk treats every list entry as a site
    # (quick-score filter, then extendScore from that one key).  Long lists (k=9, two blocks) and ordinary ones (k=13).
    import random
    rng = random.Random(5)
    genomes = [b"N" * 400 + bytes(rng.choice(b"ACGT") for _ in range(1200000)) + b"N" * 400 for _ in range(3)]
    tweak = {"maxUsableLength": 400, "maxUsableLength2": 800}
    assert run_case(genomes, 9, 1, _hard_reads(31, genomes, 90, 9), tweak=tweak, cap=64) > 20
    g13 = [make_genome(9, 300000), make_genome(10, 200000)]
    assert run_case(g13, 13, 1, _hard_reads(32, g13, 200, 13, sub=(0.05, 0.14)), cap=64) > 60
