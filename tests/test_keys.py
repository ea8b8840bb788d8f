"""The probe's per-read inputs as the product's host code makes them (bbkeys_*: AbstractMapThread.quickMap's key stage,
current/align2/AbstractMapThread.java:659-728 -> QualityTools.makeKeyProbs / KeyRing.makeOffsets3 / makeKeyScores /
makeByteScoreArray).  Checked against answers derived by hand from the Java (no sibling restatement involved)."""
import numpy as np

from bbmap_amd import keys as K


def _read(n, seed=1):
    return np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(seed).integers(0, 4, n)]


def test_quality_less_150_bases_bbmap_defaults():
    # keyDen2 = max(minKeyDensity 1.5, 15 * 13 / 150 = 1.3) = 1.5, capped by keyDensity 1.9 (AbstractMapThread.java:663-665);
    # desiredKeysFromDensity = ceil(150 * 1.5 / 13) = 18 (KeyRing.java:276); every key probability is 0, so left = 0, right = 137,
    # interval = 137 / 17 = 8.0588..., offsets = round(8.0588 i) (KeyRing.java:447-493); key scores = BASE_KEY_HIT_SCORE = 1300
    # (baseKeyScore 162 + round(1138 * 1), AbstractMapThread.java:713-719); base scores 0 (QualityTools.java:164-181)
    offs, ks, bs = K.make_keys(_read(150))
    assert offs == [0, 8, 16, 24, 32, 40, 48, 56, 64, 73, 81, 89, 97, 105, 113, 121, 129, 137]
    assert set(ks) == {1300} and not bs.any()


def test_quality_less_lengths():
    # 100 bases: keyDen2 = min(1.9, max(1.5, 1.95)) = 1.9 -> ceil(100 * 1.9 / 13) = 15 keys, first 0, last 87
    offs, _, _ = K.make_keys(_read(100))
    assert len(offs) == 15 and offs[0] == 0 and offs[-1] == 87 and offs == sorted(set(offs))
    # shorter than k: quickMap returns before making keys (AbstractMapThread.java:645)
    assert K.make_keys(_read(12))[0] == []
    # mapPacBio: density floor 2.8 whatever maxDesiredKeys says -> ceil(6000 * 2.8 / 12) = 1400 keys, key score 100 * 12
    cfg = K.default_config(K.PROFILE_PACBIO)
    offs, ks, _ = K.make_keys(_read(6000), cfg=cfg)
    assert len(offs) == 1400 and offs[0] == 0 and offs[-1] == 5988 and set(ks) == {1200}
    assert all(b > a for a, b in zip(offs, offs[1:]))
    # a mostly undefined read is discarded (DISCARD_MOSTLY_UNDEFINED_READS, :651-654)
    rd = _read(150).copy()
    rd[10:100] = ord("N")
    assert K.make_keys(rd)[0] == []


def test_semiperfect_mode_refuses_reads_with_an_undefined_base():
    # `if(PERFECTMODE || SEMIPERFECTMODE){if(r.containsUndefined()){return -1;}}` (AbstractMapThread.java:650-651): one N is enough there,
    # while the default mode only discards MOSTLY undefined reads (:652-654)
    rd = _read(150, 7).copy()
    rd[40] = ord("N")
    assert len(K.make_keys(rd)[0]) > 10
    assert K.make_keys(rd, cfg=K.default_config(semiperfectMode=1))[0] == []
    assert len(K.make_keys(_read(150, 7), cfg=K.default_config(semiperfectMode=1))[0]) > 10


def test_qualities_move_keys_and_lower_scores():
    rd = _read(150, 2)
    q = np.full(150, 35, np.uint8)
    q[60:80] = 2                                     # a stretch of bases that are probably wrong
    offs, ks, bs = K.make_keys(rd, q)
    # PROB_CORRECT[35] = 1 - 10^-3.5: a clean 13-mer has error probability 1 - (1 - 10^-3.5)^13 = 0.004103 ->
    # key score 162 + round(1138 * (1 - 0.004103)) = 162 + 1133 = 1295; base score round(100 * 0.99968) - 100 = 0
    assert ks[0] == 1295 and bs[0] == 0
    # PROB_CORRECT[2] = 1 - 10^-0.2 = 0.36904: base score round(36.9) - 100 = -63
    assert set(bs[60:80].tolist()) == {-63}
    # keys overlapping the bad stretch score far lower, and none of them is as good as a clean one
    low = [s for o, s in zip(offs, ks) if o + 13 > 60 and o < 80]
    assert low and max(low) < 1295 and min(low) < 400
    assert offs == sorted(offs) and len(set(offs)) == len(offs)
    # an unreadable read: every key almost surely wrong -> probAllErrors > 0.5 -> not probed (:723)
    assert K.make_keys(rd, np.full(150, 2, np.uint8))[0] == []


def test_batch_layout():
    reads = [_read(150, 3), _read(100, 4), _read(10, 5), _read(400, 6)]
    recs, blob, bs, ki = K.make_batch(reads)
    assert recs["len"].tolist() == [150, 100, 10, 400] and recs["bases_off"].tolist() == [0, 150, 250, 260]
    assert recs["nkeys"].tolist()[2] == 0 and len(blob) == 660
    for i, r in enumerate(reads):
        offs, ks, _ = K.make_keys(r)
        o, n = int(recs["keys_off"][i]), int(recs["nkeys"][i])
        assert ki[o:o + n].tolist() == offs and ki[o + n:o + 2 * n].tolist() == ks
