"""GPU parity tests: the HIP index probe (through the C ABI) vs the CPU oracle, SiteScore by SiteScore."""
import pytest

from bbmap_amd.index import HostIndex, DeviceIndex
from oracle.oracle import OracleIndex
from tests.index_problems import make_genome, make_reads

pytestmark = pytest.mark.gpu


def run_case(genomes, k, chromBits, reads, tweak=None):
    hi = HostIndex(genomes, k=k, chromBits=chromBits)
    oi = OracleIndex(genomes, k=k, chromBits=chromBits)
    if tweak:
        for name, val in tweak.items():
            hi.params[name] = val
            setattr(oi.s.p, name, val)
    di = DeviceIndex(hi)
    got = di.find_batch([(bp, bs, ks, offs) for bp, bm, bs, ks, offs, t in reads], max_sites=48)
    nonempty = 0
    for i, (bp, bm, bs, ks, offs, truth) in enumerate(reads):
        exp = oi.find(bp, bm, bs, ks, offs, cap=48)
        assert got[i] is not None, "read %d: probe reported overflow/unsupported" % i
        assert got[i] == exp, "read %d (truth %s): %s != %s" % (i, truth, got[i], exp)
        nonempty += bool(exp)
    di.close()
    return nonempty


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
