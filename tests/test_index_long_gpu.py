"""GPU parity tests of the long-read probe kernel (index_probe_long.hip: one read per wavefront, up to 6016 bases and 2047 keys):
 * with BBIndex's constants on the problems of test_index_gpu.py (a third, structurally different kernel against the same oracle),
   including reads with hundreds of keys, which neither older kernel takes;
 * with BBIndexPacBio's constants (mapPacBio) against the oracle compiled with -DORC_PACBIO: device-built index vs the oracle's
   own build, then SiteScore lists of PacBio-like reads of up to 6000 bases, keys placed by bbkeys_make as quickMap places them."""
import numpy as np
import pytest

from bbmap_amd import keys as K
from bbmap_amd.index import HostIndex, DeviceIndex, PROFILE_PACBIO
from oracle.oracle import OracleIndex
from tests.index_problems import make_genome, make_reads, revcomp

pytestmark = pytest.mark.gpu


def _compare(di, oi, reads, cap):
    exp = []
    for bp, bm, bs, ks, offs, truth in reads:
        try:
            exp.append(oi.find(bp, bm, bs, ks, offs, cap=cap))
        except RuntimeError:
            exp.append(None)
    got = di.find_batch([(bp, bs, ks, offs) for bp, bm, bs, ks, offs, t in reads], max_sites=cap)
    for i, r in enumerate(reads):
        assert got[i] == exp[i], "long kernel, read %d (truth %s, %d keys): %s != %s" % (i, r[5], len(r[4]), got[i], exp[i])
    return sum(bool(e) for e in exp)


def test_bbmap_constants_match_the_oracle_and_the_other_kernels():
    genomes = [make_genome(31, 250000)]
    hi = HostIndex(genomes, k=13)
    oi = OracleIndex(genomes, k=13)
    di = DeviceIndex(hi)
    di.set_kernel("long")
    assert _compare(di, oi, make_reads(7, genomes, 400, k=13), 48) > 250
    # hundreds of keys per read: 600-base reads at density 6
    many = make_reads(12, genomes, 60, read_len=600, k=13, density=6.0)
    assert max(len(r[4]) for r in many) > 250
    _compare(di, oi, many, 48)
    di.close()
    # three chromosomes over two blocks, k = 10, no prescan / no early quit
    genomes = [make_genome(41, 90000), make_genome(42, 60000), make_genome(43, 50000)]
    hi = HostIndex(genomes, k=10, chromBits=1)
    oi = OracleIndex(genomes, k=10, chromBits=1)
    for name, val in {"prescanQscore": 0, "quitAfterTwoPerfects": 0}.items():
        hi.params[name] = val
        setattr(oi.s.p, name, val)
    di = DeviceIndex(hi)
    di.set_kernel("long")
    _compare(di, oi, make_reads(8, genomes, 300, k=10), 48)
    di.close()


def pacbio_piece(rng, G, lo, hi, err=(0.10, 0.16)):
    """A piece of a PacBio-like read: substitutions, deletions and insertions at a total rate drawn from `err`."""
    L = int(rng.integers(lo, hi))
    st = int(rng.integers(600, len(G) - L - 1200))
    e = rng.uniform(*err)
    src = np.frombuffer(G[st:st + L + L // 4], np.uint8)
    x = rng.random(len(src))
    keep = x >= e * 0.35
    sub = (x >= e * 0.35) & (x < e * 0.55)
    piece = src.copy()
    piece[sub] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(sub.sum()))]
    piece = piece[keep]
    ins_at = np.nonzero(rng.random(len(piece)) < e * 0.45)[0]
    piece = np.insert(piece, ins_at, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(ins_at))])[:L]
    return piece.tobytes(), st


def test_pacbio_profile_index_and_probe():
    genomes = [make_genome(81, 400000), make_genome(82, 250000)]
    oi = OracleIndex(genomes, profile="pacbio")
    di = DeviceIndex.build(genomes, profile=PROFILE_PACBIO)
    assert di.host.k == 12
    # the device-built index equals the oracle's: arrays and every derived tunable
    for name in ("k", "chromBits", "maxIndel", "maxIndel2", "maxUsableLength", "maxUsableLength2", "maxHitsReduction2", "maximumMaxHitsReduction",
                 "hitReductionDiv", "maxAverageListToSearch", "maxAverageListToSearch2", "maxShortestListToSearch", "pointsPerSite"):
        assert di.host.params[name] == getattr(oi.s.p, name), name
    assert di.host.params["profile"] == PROFILE_PACBIO and di.host.params["maxIndel"] == 100
    starts, sites, counts, hist = di.export_block(0)
    ostarts, osites = oi.block_arrays(0)
    assert np.array_equal(starts, ostarts) and np.array_equal(sites, osites) and np.array_equal(counts, oi.counts())
    assert np.array_equal(hist, np.ctypeslib.as_array(oi.s.lengthHistogram))
    cfg = K.default_config(K.PROFILE_PACBIO)
    rng = np.random.default_rng(5)
    reads = []
    for i in range(40):
        ci = int(rng.integers(0, 2))
        lo, hi_ = ((200, 900), (1500, 3000), (5000, 6001))[i % 3]
        rd, st = pacbio_piece(rng, genomes[ci], lo, hi_, err=(0.02, 0.06) if i % 5 == 0 else (0.10, 0.16))
        if i % 7 == 3:
            rd = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), len(rd)))            # junk: no site
        strand = i & 1
        bp = revcomp(rd) if strand else rd
        offs, ks, bs = K.make_keys(bp, None, cfg)
        reads.append((bp, revcomp(bp), bs.tolist(), ks, offs, (ci + 1, strand, st)))
    assert max(len(r[4]) for r in reads) >= 1300
    found = _compare(di, oi, reads, 64)
    assert found >= 30
    # the true site is among the sites of nearly every non-junk read
    exp = [oi.find(bp, bm, bs, ks, offs, cap=64) for bp, bm, bs, ks, offs, t in reads]
    hits = 0
    for (bp, bm, bs, ks, offs, (chrom, strand, st)), sites in zip(reads, exp):
        hits += any(s["chrom"] == chrom and s["strand"] == strand and abs(s["start"] - st) < 400 for s in sites)
    assert hits >= 30
    di.close()
