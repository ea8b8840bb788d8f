"""CPU tests of the mapper restatement (oracle/mapper_oracle.c): planted reads and pairs come back at the coordinates they were
drawn from, rescue finds mates the probe missed, list invariants hold.  Parity of this restatement with the reference is pinned
by the reference-held fixture test (tests/test_golden_phix.py), not here."""
import numpy as np

from bbmap_amd import workload as W
from oracle import oracle as O


def _setup(L=150, k=12, size=150000, seed=5):
    ref = W.make_reference(size, seed=seed, pad=2000, repeat_frac=0.1)
    oi = O.OracleIndex([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    return ref, oi, offs, [100 * k] * len(offs)


def test_single_ended_reads_map_to_their_origin():
    ref, oi, offs, ks = _setup()
    reads, _, truth = W.make_reads_and_jobs(ref, 600, seed=3, pad=2000)
    # (the lists as scoreSlow leaves them: the final stage lowers a list's scores when several sites tie, applyClearzone3)
    out = O.map_batch(oi, reads, None, 150, offs, ks, params=O.map_default_params(finalStage=0))
    n1, top = out["nsites1"], out["sites1"][:, 0]
    assert (n1 > 0).mean() > 0.99
    assert ((n1 > 0) & (np.abs(top["start"] - truth["start"]) <= 40)).mean() > 0.98
    for r in range(600):                                   # lists end sorted by score; scores are consistent
        s = out["sites1"][r, : n1[r]]
        assert (np.diff(s["score"]) <= 0).all()
        assert (s["score"] == s["slowScore"]).all()
    perfect = ~truth["imperfect"]
    assert (top["slowScore"][perfect & (n1 > 0)] == 70 + 149 * 100).all()


def test_pairs_rescue_and_fill_log():
    ref, oi, offs, ks = _setup(seed=6)
    oi.s.p.quitAfterTwoPerfects = 0
    reads, truth = W.make_pairs(ref, 500, seed=4, pad=2000, hard_frac=0.1)
    r = reads.reshape(-1, 150)
    out = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), 150, offs, ks, params=O.map_default_params(finalStage=0))
    n1, n2 = out["nsites1"], out["nsites2"]
    t1, t2 = out["sites1"][:, 0], out["sites2"][:, 0]
    assert ((n1 > 0) & (np.abs(t1["start"] - truth["start1"]) <= 40) & (t1["strand"] == truth["strand1"])).mean() > 0.97
    assert ((n2 > 0) & (np.abs(t2["start"] - truth["start2"]) <= 40) & (t2["strand"] == truth["strand2"])).mean() > 0.97
    log = out["log"]
    assert (log["kind"] == 2).sum() > 3 and out["stats"][2] > 10          # slowRescue fills, quickRescue scans
    # fills are numbered per read without holes
    for rd in np.unique(log["read"]):
        seqs = np.sort(log["seq"][log["read"] == rd])
        assert seqs.tolist() == list(range(len(seqs)))
    # a rescued site is paired with its anchor
    for p in range(500):
        for s in out["sites1"][p, : n1[p]]:
            if s["rescued"]:
                assert s["pairedScore"] > 0


def test_ratios_follow_the_reference_formulas():
    L = O.lib()
    import ctypes as C
    L.orc_ratio_paired.restype = C.c_float
    L.orc_ratio_paired.argtypes = [C.c_float]
    L.orc_ratio_pre_rescue.restype = C.c_float
    L.orc_ratio_pre_rescue.argtypes = [C.c_float]
    assert abs(L.orc_ratio_paired(0.56) - 0.448) < 1e-6               # max(0.56 * .80, 1 - 0.44 * 1.4)
    assert abs(L.orc_ratio_pre_rescue(0.56) - 0.336) < 1e-6           # max(0.56 * .60, 1 - 0.44 * 1.8)
