"""CPU tests of the final alignment stage's restatement (oracle/final_stage.inc): invariants the reference asserts on its own results
(a match string consumes exactly the read and spans exactly [start, stop]: SiteScore.lengthsAgree, Read.CHECKSITES), planted reads
come back with the CIGAR they were given, the rare branches are reachable through final_reads.  Parity of this restatement with the
reference is pinned by the reference-held fixture (tests/test_golden_phix.py)."""
import numpy as np

from bbmap_amd import workload as W
from oracle import oracle as O
from tests.final_problems import edge_reads, perturb, plant_edge_sites, tip_reads

L, K = 150, 12


def _consumes(m):
    read = sum(m.count(c) for c in b"mSNIXYC")
    ref = sum(m.count(c) for c in b"mSNDXYC")
    return read, ref


def _setup(seed=15, size=200000, pad=2000):
    ref = W.make_reference(size, seed=seed, pad=pad, repeat_frac=0.1)
    oi = O.OracleIndex([ref], k=K)
    offs = O.make_offsets(L, K, 1.9)
    return ref, oi, offs, [100 * K] * len(offs)


def test_final_records_are_consistent_single_ended():
    ref, oi, offs, ks = _setup()
    reads, _, truth = W.make_reads_and_jobs(ref, 1500, read_len=L, seed=9, pad=2000, long_del_frac=0.3)
    out = O.map_batch(oi, reads, None, L, offs, ks, cap=64, match_stride=4200, threads=8)
    f, fm = out["final1"], out["fmatch1"]
    assert (f["mapped"] > 0).mean() > 0.99
    kinds = np.bincount(out["log"]["kind"], minlength=7)
    assert kinds[3] > 300                                      # realign_new's fills are in the log
    for i in range(len(f)):
        if not f["mapped"][i]:
            assert f["match_len"][i] == 0 and f["start"][i] == -1 and f["mapScore"][i] == 0
            continue
        ml = int(f["match_len"][i])
        assert ml > 0
        if ml > fm.shape[1]:
            continue
        m = fm[i][:ml].tobytes()
        rd, rf = _consumes(m)
        assert rd == L, (i, m)
        assert rf == f["stop"][i] - f["start"][i] + 1, (i, m, f[i])
        assert set(m) <= set(b"mSNDIXYC")
        if f["perfect"][i]:
            assert m == b"m" * L
        top = out["sites1"][i][0]                              # the record is the top site's
        assert (top["start"], top["stop"], top["strand"]) == (f["start"][i], f["stop"][i], f["strand"][i])
    # reads the generator left unmutated come home perfect at their origin
    perfect = ~truth["imperfect"]
    assert (f["perfect"][perfect] == 1).mean() > 0.99
    assert (f["start"][perfect] == truth["start"][perfect]).mean() > 0.99


def test_planted_deletion_shows_in_the_match_string():
    ref, oi, offs, ks = _setup(seed=21)
    rng = np.random.default_rng(4)
    reads, want = [], []
    for i in range(200):
        a = int(rng.integers(3000, len(ref) - 4000))
        d = int(rng.integers(5, 60))
        p = int(rng.integers(30, 120))
        reads.append(np.concatenate([ref[a:a + p], ref[a + p + d:a + d + L]]))
        want.append((a, a + d + L - 1, d))
    reads = np.stack(reads)
    out = O.map_batch(oi, reads.reshape(-1), None, L, offs, ks, cap=64, match_stride=4200, threads=8)
    f, fm = out["final1"], out["fmatch1"]
    ok = 0
    for i, (a, b, d) in enumerate(want):
        if not f["mapped"][i]:
            continue
        m = fm[i][:f["match_len"][i]].tobytes()
        if f["start"][i] == a and f["stop"][i] == b and m.count(b"D") == d and b"D" * d in m:
            ok += 1
    assert ok >= 190, ok


def test_pairs_final_flags():
    ref, oi, offs, ks = _setup(seed=6)
    oi.s.p.quitAfterTwoPerfects = 0
    reads, truth = W.make_pairs(ref, 600, read_len=L, seed=4, pad=2000, hard_frac=0.1)
    r = reads.reshape(-1, L)
    out = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), L, offs, ks, cap=64, match_stride=4200, threads=8)
    f1, f2 = out["final1"], out["final2"]
    assert (f1["paired"] == f2["paired"]).all()                # a pair is paired on both mates or on neither
    assert f1["paired"].mean() > 0.95
    both = (f1["paired"] > 0)
    assert (f1["strand"][both] != f2["strand"][both]).all() and (f1["chrom"][both] == f2["chrom"][both]).all()
    assert (f1["rescued"].sum() + f2["rescued"].sum()) > 5
    for f, fm in ((f1, out["fmatch1"]), (f2, out["fmatch2"])):
        for i in range(len(f)):
            if f["mapped"][i] and f["match_len"][i] <= fm.shape[1]:
                rd, rf = _consumes(fm[i][:f["match_len"][i]].tobytes())
                assert rd == L and rf == f["stop"][i] - f["start"][i] + 1


def test_rare_branches_are_reachable():
    """final_reads over damaged lists reaches fixXY, clipTipIndels, toLocalAlignment, the recursion, the re-sort loop and (small MSA)
    the third fill + fillUnlimited; the GPU test compares exactly these runs with the device."""
    ref = W.make_reference(200000, seed=18, pad=0, repeat_frac=0.1)
    reads = tip_reads(ref, 2400, L, 3, 300)
    er, info = edge_reads(ref, 400, L, 4)
    reads = np.concatenate([reads, er])
    oi = O.OracleIndex([ref], k=K)
    offs = O.make_offsets(L, K, 1.9)
    ks = [100 * K] * len(offs)
    pre = O.map_batch(oi, reads.reshape(-1), None, L, offs, ks, cap=32, match_stride=4200, threads=8, params=O.map_default_params(finalStage=0))
    s, ns = perturb(pre["sites1"], pre["nsites1"], 5, len(ref))
    plant_edge_sites(s, ns, 2400, info)
    recs = np.zeros(len(ns), O.READ_DTYPE)
    recs["bases_off"] = np.arange(len(ns)) * L
    recs["len"] = L
    for cols, need in ((3000, ("clip_tip_indels", "fix_xy", "to_local_clipped", "realign_recursion", "second_realign", "later_site_matched")),
                       (250, ("resort_loop",))):
        O.final_branch_counts(oi)
        out = O.final_reads(oi, recs, reads.reshape(-1), s, ns, params=O.map_default_params(msaMaxColumns=cols, alignColumns=cols))
        b = O.final_branch_counts(oi)
        for name in need:
            assert b[name] > 0, (cols, name, b)
        kinds = np.bincount(out["log"]["kind"], minlength=7)
        assert kinds[4] > 100
        if cols == 250:
            assert kinds[5] > 10 and kinds[6] >= 1, kinds
        f, fm = out["final"], out["fmatch"]
        for i in range(len(f)):                                # lengthsAgree holds on every record, clipped ones included
            if f["mapped"][i] and 0 < f["match_len"][i] <= fm.shape[1]:
                rd, rf = _consumes(fm[i][:f["match_len"][i]].tobytes())
                assert rd == L and rf == f["stop"][i] - f["start"][i] + 1, (cols, i, fm[i][:f["match_len"][i]].tobytes(), f[i])


def test_a_tighter_min_score_can_change_a_fill():
    """Why no kernel here ever substitutes a tighter minScore for the caller's: the pruning of fillLimitedX
    (MultiStateAligner11tsJNI.c fillLimitedX, restated in oracle/msa11ts_oracle.c) is not admissible.  Re-running realign_new's own
    fills on tip-damaged reads with minScore set 400 points BELOW the score the fill found returns null for some and, for a few, a
    different non-null alignment with a lower score -- so a result obtained under a tighter bound proves nothing about the result under
    the caller's bound (the narrow-window kernel's BBMSA_NO_ITERATIONS shortcut of rounds 2-3 and a 'known score' bound for the
    final stage were both withdrawn over this)."""
    from tests.test_final_gpu import _problem
    ref, oi, reads, recs, offs, ks, s, ns = _problem(False, 1200, 40, 3)
    orc = O.final_reads(oi, recs, reads.reshape(-1), s, ns, paired=False, params=O.map_default_params(msaMaxColumns=601, alignColumns=601))
    msa = O.OracleMSA(601, 601)
    refb = ref.tobytes()
    comp = np.full(256, ord("N"), np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    tried = null = other = 0
    for e in orc["log"]:
        if e["kind"] not in (3, 4, 5) or e["ngaps"] or e["score_len"] <= 0:
            continue
        sc = int(e["score"][0])
        if sc - 400 <= int(e["minScore"]):
            continue
        rd = reads[int(e["read"])]
        if e["strand"]:
            rd = comp[rd[::-1]]
        a, b = int(e["refStartLoc"]), int(e["refEndLoc"])
        loose, _ = msa.fillAndScoreLimited(rd.tobytes(), refb, a, b, int(e["minScore"]))
        assert [int(x) for x in loose[:3]] == [int(x) for x in e["score"][:3]]              # (the log's result is reproducible)
        tight, _ = msa.fillAndScoreLimited(rd.tobytes(), refb, a, b, sc - 400)
        tried += 1
        if tight is None:
            null += 1
        elif [int(x) for x in tight[:3]] != [int(x) for x in loose[:3]]:
            other += 1
            assert int(tight[0]) < int(loose[0])
    assert tried > 500 and null > 20 and other > 5, (tried, null, other)
