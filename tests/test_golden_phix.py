"""configs[0] of BASELINE.json and the only fixture the reference holds for this path: phix174_ill.ref.fa.gz + sample1.fq.gz /
sample2.fq.gz (100 synthetic 100-bp read pairs whose names carry the true strand, start and stop, written by the reference's
own generator).  What it pins: the END RESULT of the whole path -- probe, ungapped scores, scoreSlow DP, score2's limits, mate
rescue -- lands on the coordinates the reference's generator recorded, under the reference's own two correctness rules
(AbstractMapThread.isCorrectHit / isCorrectHitLoose, current/align2/AbstractMapThread.java:2692-2717, thresh = CORRECT_THRESH = 0).
What it does not pin: intermediate values (scores, visited-cell counts, match strings); for those the oracle remains a
restatement.  The reads' qualities are ignored (key offsets and key scores as for quality-less input), so the few reads whose
placement depends on quality-weighted keys may differ from a real BBMap run; the counts below are what the restatement gives
and are asserted as floors.
CPU test: the oracle against the truth.  GPU tests: the device mapper equals the oracle on all 200 reads, single-ended and
paired, and therefore meets the same truth."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_phix import phix_reference, sample_reads

L, K = 100, 13


def _score_against_truth(sites, nsites, truth):
    top = sites[:, 0]
    mapped = nsites > 0
    same = mapped & (top["chrom"] == 1) & (top["strand"] == truth["strand"])
    strict = same & (top["start"] == truth["start"]) & (top["stop"] == truth["stop"])          # isCorrectHit, thresh 0
    loose = same & ((top["start"] == truth["start"]) | (top["stop"] == truth["stop"]))          # isCorrectHitLoose, thresh 0
    return int(mapped.sum()), int(strict.sum()), int(loose.sum())


def _oracle(paired):
    ref = phix_reference()
    assert len(ref) == 5386 + 16000
    r1, t1 = sample_reads(1)
    r2, t2 = sample_reads(2)
    oi = O.OracleIndex([ref], k=K)
    offs = O.make_offsets(L, K, 1.9)
    ks = [100 * K] * len(offs)
    if paired:
        oi.s.p.quitAfterTwoPerfects = 0
        out = O.map_batch(oi, r1.reshape(-1), r2.reshape(-1), L, offs, ks)
    else:
        out = O.map_batch(oi, r1.reshape(-1), None, L, offs, ks)
    return ref, (r1, t1), (r2, t2), offs, ks, out


def test_oracle_single_ended_meets_the_fixture_truth():
    _, (r1, t1), _, _, _, out = _oracle(False)
    mapped, strict, loose = _score_against_truth(out["sites1"], out["nsites1"], t1)
    assert mapped >= 99 and strict >= 86 and loose >= 96, (mapped, strict, loose)
    # a read the generator left unmutated comes back perfect at exactly its origin
    top = out["sites1"][:, 0]
    perfect = (out["nsites1"] > 0) & (top["slowScore"] == 70 + 99 * 100)
    assert perfect.sum() >= 5
    assert ((top["start"] == t1["start"]) & (top["stop"] == t1["stop"]) & (top["strand"] == t1["strand"]))[perfect].all()


def test_oracle_paired_meets_the_fixture_truth():
    _, (r1, t1), (r2, t2), _, _, out = _oracle(True)
    m1 = _score_against_truth(out["sites1"], out["nsites1"], t1)
    m2 = _score_against_truth(out["sites2"], out["nsites2"], t2)
    assert m1[0] >= 99 and m1[1] >= 86 and m1[2] >= 96, m1
    assert m2[0] >= 99 and m2[1] >= 83 and m2[2] >= 95, m2
    assert out["stats"][2] > 50                      # quickRescue scans ran


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
def test_device_mapper_equals_oracle_on_the_fixture(paired):
    from bbmap_amd.index import DeviceIndex
    from bbmap_amd.mapper import Mapper
    from tests.mapper_check import compare
    ref, (r1, t1), (r2, t2), offs, ks, orc = _oracle(paired)
    if paired:
        reads = np.empty((200, L), np.uint8)
        reads[0::2], reads[1::2] = r1, r2
    else:
        reads = r1
    n = len(reads)
    di = DeviceIndex.build([ref], k=K)
    mp = Mapper(di, n, L, offs, ks, paired=paired, max_sites=32)
    mp.load_reads(reads)
    mp.step()
    out, st = mp.fetch(), mp.stats()
    mp.close()
    di.close()
    assert st["reads_overflowed"] == 0
    bad = compare(out, orc, n, paired)
    assert not bad, "\n".join(bad[:20])
    if paired:
        m1 = _score_against_truth(out["sites"][0::2], out["nsites"][0::2], t1)
        m2 = _score_against_truth(out["sites"][1::2], out["nsites"][1::2], t2)
        assert m1[1] >= 86 and m1[2] >= 96 and m2[1] >= 83 and m2[2] >= 95, (m1, m2)
    else:
        m = _score_against_truth(out["sites"], out["nsites"], t1)
        assert m[0] >= 99 and m[1] >= 86 and m[2] >= 96, m
