"""configs[0] of BASELINE.json and the only fixture the reference holds for this path: phix174_ill.ref.fa.gz + sample1.fq.gz /
sample2.fq.gz (100 synthetic 100-bp read pairs whose names carry the true strand, start and stop, written by the reference's
own generator, with qualities).  Six runs: sample1 / sample2 single-ended and the pairs, each with keys placed from the qualities
(GENERATE_KEY_SCORES_FROM_QUALITY: makeKeyProbs / makeOffsets3 / makeKeyScores / makeByteScoreArray, the product's bbkeys_*) and as
for quality-less input.

What is pinned, and by what:
 * END RESULT vs the fixture's truth (reference-held): the reference's own two correctness rules (AbstractMapThread.isCorrectHit /
   isCorrectHitLoose, current/align2/AbstractMapThread.java:2692-2717, thresh 0) as floors -- applied, as the reference applies them
   (calcStatistics1 at the end of processRead), to the FINAL records: the coordinates after genMatchString -> realign_new,
   clipTipIndels, the ambiguity policy and the minimum-score cut (FLOORS_FINAL); and, for the stage-by-stage view, to the site lists
   as scoreSlow and rescue leave them (finalStage = 0; FLOORS);
 * the TRUTH-WINDOW property (reference-held truth + the DP): the fill scoreSlow would issue for a site at exactly the coordinates
   in a read's name never scores above what the mapper found for that read -- a mapper may beat the generator's placement, never
   lose to it (a wrong tie-break or window in the flow would show here);
 * a per-read table (tests/golden/phix_expected.json, written by scripts/make_phix_expected.py from the oracle at the commit that
   made it): every read's top site (strand, start, stop, slowScore) in all six runs, compared field by field -- not reference
   output, a regression pin: a change that moves one read fails.
What it does not pin: intermediate values (visited-cell counts, match strings); for those the oracle remains a restatement.
CPU tests: the oracle.  GPU tests: the device mapper equals the oracle on every run (site lists, fills, traceback strings), so it
meets the same truth, table and property."""
import json
import os

import numpy as np
import pytest

from tests.golden_phix import (HERE, PACBIO_MSA, fixture_runs, fixture_runs_pacbio, phix_reference, sample_reads, truth_window_jobs,
                               truth_window_jobs_pacbio, truth_window_scores, truth_window_scores_pacbio)

FLOORS = {      # (mapped, strict, loose) of a sample's 100 reads, as the restatement gives them: asserted as floors against the truth
    ("se", 1): (99, 86, 98), ("se", 2): (98, 79, 95),       # mapped on their own (sample2 with quality-placed keys: 95 loose, 96 without)
    ("pe", 1): (99, 86, 98), ("pe", 2): (98, 82, 98),       # as pairs (rescue brings three of sample2's reads home)
}
FLOORS_FINAL = {  # the same on the records BBMap prints.  Lower than FLOORS where a read's best site ends below MINIMUM_ALIGNMENT_SCORE_RATIO
                  # (`r.mapScore<maxSwScore*MINIMUM_ALIGNMENT_SCORE_RATIO -> clearMapping`, BBMapThread.java:697-699) and is reported unmapped
    ("se", 1): (94, 83, 93), ("se", 2): (91, 78, 91),
    ("pe", 1): (98, 85, 97), ("pe", 2): (94, 78, 94),
}


def _final_against_truth(fin, truth):
    mapped = fin["mapped"] > 0
    same = mapped & (fin["chrom"] == 1) & (fin["strand"] == truth["strand"])
    strict = same & (fin["start"] == truth["start"]) & (fin["stop"] == truth["stop"])
    loose = same & ((fin["start"] == truth["start"]) | (fin["stop"] == truth["stop"]))
    return int(mapped.sum()), int(strict.sum()), int(loose.sum())


def _score_against_truth(sites, nsites, truth):
    top = sites[:, 0]
    mapped = nsites > 0
    same = mapped & (top["chrom"] == 1) & (top["strand"] == truth["strand"])
    strict = same & (top["start"] == truth["start"]) & (top["stop"] == truth["stop"])          # isCorrectHit, thresh 0
    loose = same & ((top["start"] == truth["start"]) | (top["stop"] == truth["stop"]))          # isCorrectHitLoose, thresh 0
    return int(mapped.sum()), int(strict.sum()), int(loose.sum())


def _split(name, out):
    """{sample: (sites, nsites)} of one run's output"""
    if name.startswith("pe"):
        return {1: (out["sites"][0::2], out["nsites"][0::2]), 2: (out["sites"][1::2], out["nsites"][1::2])}
    return {int(name[2]): (out["sites"], out["nsites"])}


@pytest.fixture(scope="module")
def runs():
    r = fixture_runs()
    for v in r.values():
        v["out"] = v["oracle"](final_stage=0)          # the lists as scoreSlow and rescue leave them
        v["final"] = v["oracle"]()                     # the whole of processRead / processReadPair
    return r


@pytest.fixture(scope="module")
def expected():
    return json.load(open(os.path.join(HERE, "phix_expected.json")))


def _check_truth(name, out):
    for which, (sites, nsites) in _split(name, out).items():
        _, truth = sample_reads(which)
        got = _score_against_truth(sites, nsites, truth)
        floor = FLOORS[(name[:2], which)]
        assert all(g >= f for g, f in zip(got, floor)), (name, which, got, floor)


def _check_final(name, out, expected, check_truth=True):
    """the final records: floors against the truth (isCorrectHit on what BBMap prints) and the per-read table, match strings included"""
    fin, blob = out["final"], out.get("final_match")
    if check_truth:
        for which in ((1, 2) if name.startswith("pe") else (int(name[2]),)):
            f = fin[which - 1::2] if name.startswith("pe") else fin
            _, truth = sample_reads(which)
            got = _final_against_truth(f, truth)
            floor = FLOORS_FINAL[(name[:2], which)]
            assert all(g >= fl for g, fl in zip(got, floor)), (name, which, got, floor)
    for i, exp in enumerate(expected["final"][name]):
        f = fin[i]
        ml = int(f["match_len"])
        if blob is not None:                           # device: (records, packed strings)
            m = blob[int(f["match_off"]): int(f["match_off"]) + ml].tobytes().decode()
        else:
            m = out["fmatch"][i][:ml].tobytes().decode()
        got = [int(f["mapped"]), int(f["strand"]), int(f["start"]), int(f["stop"]), int(f["mapScore"]), int(f["paired"]), int(f["ambiguous"]), m]
        assert got == exp, "%s read %d final record: %s, table %s" % (name, i, got, exp)


def _check_table(name, out, expected):
    top = out["sites"][:, 0]
    for i, exp in enumerate(expected["runs"][name]):
        ns = int(out["nsites"][i])
        got = [ns, int(top["strand"][i]), int(top["start"][i]), int(top["stop"][i]), int(top["slowScore"][i])] if ns > 0 else [ns, 0, 0, 0, 0]
        assert got == exp, "%s read %d: %s, table %s" % (name, i, got, exp)


def _check_truth_window(name, out, tw):
    """The mapper's best slowScore of a read is at least the score of the fill against the read's truth window whenever the read's
    site list holds a site (on the truth's strand) whose span covers the truth's: that site's own fill window contains the truth
    window, so its optimum cannot be lower (and a limit raised by an earlier, better site only proves the point).  A read without
    such a site is a PROBE miss -- the heuristic never proposed the locus -- and is returned, not failed."""
    worse, uncovered = [], []
    for which, (sites, nsites) in _split(name, out).items():
        _, truth = sample_reads(which)
        for i, t in enumerate(tw["sample%d" % which]):
            if t is None:
                continue                                   # the truth window itself does not reach the minimum score
            n = max(int(nsites[i]), 0)
            best = int(sites[i]["slowScore"][:n].max()) if n else -1
            if best >= t[0]:
                continue
            s = sites[i][:n]
            covered = bool(((s["strand"] == truth["strand"][i]) & (s["start"] <= t[1]) & (s["stop"] >= t[2])).any()) if n else False
            (worse if covered else uncovered).append((name, which, i, best, t))
    assert not worse, worse
    return uncovered


def test_oracle_meets_the_fixture_truth_in_all_six_runs(runs):
    for name, r in runs.items():
        _check_truth(name, r["out"])
    # the rescue stage ran in the paired runs, and qualities changed the inputs (key scores below the maximum)
    assert runs["pe_qual"]["out"]["stats"][2] > 50
    ki = runs["se1_qual"]["inputs"][3]
    assert (ki < 1300).sum() > 100 and (runs["se1_qual"]["inputs"][2] < 0).sum() > 1000
    # a read the generator left unmutated comes back perfect at exactly its origin
    out = runs["se1_noqual"]["out"]
    _, t1 = sample_reads(1)
    top = out["sites"][:, 0]
    perfect = (out["nsites"] > 0) & (top["slowScore"] == 70 + 99 * 100)
    assert perfect.sum() >= 5
    assert ((top["start"] == t1["start"]) & (top["stop"] == t1["stop"]) & (top["strand"] == t1["strand"]))[perfect].all()


def test_oracle_equals_the_per_read_table(runs, expected):
    for name, r in runs.items():
        _check_table(name, r["out"], expected)
        _check_final(name, r["final"], expected)


def test_final_records_on_the_fixture(runs):
    """What the final alignment stage does to the fixture's reads, stated as numbers: it never moves a strict hit away from the truth,
    every mapped read's match string spans exactly [start, stop] and consumes exactly the read, and the reads it reports unmapped are
    those whose best alignment stays below MINIMUM_ALIGNMENT_SCORE_RATIO."""
    min_score = int(np.float32(0.56) * np.float32(70 + 99 * 100))
    for name, r in runs.items():
        pre, fin = r["out"], r["final"]["final"]
        for which, (sites, nsites) in _split(name, pre).items():
            f = fin[which - 1::2] if name.startswith("pe") else fin
            fm = r["final"]["fmatch"][which - 1::2] if name.startswith("pe") else r["final"]["fmatch"]
            _, truth = sample_reads(which)
            top = sites[:, 0]
            for i in range(100):
                was_strict = nsites[i] > 0 and top["strand"][i] == truth["strand"][i] and top["start"][i] == truth["start"][i] and top["stop"][i] == truth["stop"][i]
                if f["mapped"][i]:
                    m = fm[i][:f["match_len"][i]].tobytes()
                    assert sum(m.count(c) for c in b"mSNIXYC") == 100
                    assert sum(m.count(c) for c in b"mSNDXYC") == f["stop"][i] - f["start"][i] + 1
                    if was_strict and not name.startswith("pe"):
                        assert (f["start"][i], f["stop"][i]) == (truth["start"][i], truth["stop"][i]), (name, which, i)
                elif nsites[i] > 0 and not name.startswith("pe"):
                    assert top["slowScore"][i] < min_score + 300, (name, which, i, int(top["slowScore"][i]))


def test_mapper_never_loses_to_the_truth_window(runs, expected):
    tw = {"sample1": truth_window_scores(1), "sample2": truth_window_scores(2)}
    assert tw == expected["truth_window"]                  # the fills themselves are pinned too
    assert sum(t is not None for t in tw["sample1"]) >= 90    # (the rest: the truth window itself stays below the minimum score)
    missed = []
    for name, r in runs.items():
        missed += _check_truth_window(name, r["out"], tw)
    # One read loses to its truth window, and not in the DP: sample2's read 43 (an 85-base deletion) with quality-placed keys -- the
    # probe proposes [11721, 11864], 46 bases left of the truth and 87 short of its end, and that window's fill stays below the
    # minimum score; as a mate, rescue finds [11767, 11863] (5,135 points), again short of the deletion.  With the quality-less
    # key offsets the same read comes home at 6,993 points.  Every other read of every run holds the property.
    assert sorted((m[0], m[1], m[2]) for m in missed) == [("pe_qual", 2, 43), ("se2_qual", 2, 43)], missed


# ---------------------------------------------------------------------------------------------- mapPacBio's classes on the same fixture
# BBIndexPacBio / BBMapThreadPacBio / MultiStateAligner9PacBio have no reference-held vector of their own; the fixture's 100-base reads are
# legal input to mapPacBio.sh, so its truth pins these classes' constants too: floors by the reference's correctness rules, the
# truth-window property with the 9PacBio scores, a per-read table.
FLOORS_PACBIO = {1: (100, 80, 99), 2: (98, 78, 97)}          # (mapped, strict, loose) per sample; keys from qualities or not: the same
# reads whose list holds no site covering the truth (BBIndexPacBio's MAX_INDEL is 100: a 100-200-base deletion is not ONE site there,
# the probe reports the longer half): the heuristic's misses, not the DP's -- listed so that a new one shows
PACBIO_UNCOVERED = {1: [4, 9, 23, 24, 32, 48, 62, 77], 2: [44, 46, 47, 62, 63, 73, 79, 85]}


@pytest.fixture(scope="module")
def pacbio_runs():
    r = fixture_runs_pacbio()
    for v in r.values():
        v["out"] = v["oracle"]()
    return r


def _check_pacbio(name, out, expected, tw):
    which = int(name[2])
    _, truth = sample_reads(which)
    got = _score_against_truth(out["sites"], out["nsites"], truth)
    assert all(g >= f for g, f in zip(got, FLOORS_PACBIO[which])), (name, got)
    top = out["sites"][:, 0]
    for i, exp in enumerate(expected["pacbio_runs"][name]):
        ns = int(out["nsites"][i])
        g = [ns, int(top["strand"][i]), int(top["start"][i]), int(top["stop"][i]), int(top["slowScore"][i])] if ns > 0 else [ns, 0, 0, 0, 0]
        assert g == exp, "%s read %d: %s, table %s" % (name, i, g, exp)
    worse, uncovered = [], []
    for i, t in enumerate(tw["sample%d" % which]):
        if t is None:
            continue
        n = max(int(out["nsites"][i]), 0)
        best = int(out["sites"][i]["slowScore"][:n].max()) if n else -1
        if best >= t[0]:
            continue
        s = out["sites"][i][:n]
        covered = bool(((s["strand"] == truth["strand"][i]) & (s["start"] <= t[1]) & (s["stop"] >= t[2])).any()) if n else False
        (worse if covered else uncovered).append(i)
    assert not worse, (name, worse)
    assert uncovered == PACBIO_UNCOVERED[which], (name, uncovered)


def test_pacbio_classes_meet_the_fixture_truth(pacbio_runs, expected):
    tw = {"sample1": truth_window_scores_pacbio(1), "sample2": truth_window_scores_pacbio(2)}
    assert tw == expected["pacbio_truth_window"]
    for name, r in pacbio_runs.items():
        _check_pacbio(name, r["out"], expected, tw)
        assert len(r["out"]["log"]) > 50                   # MultiStateAligner9PacBio fills ran


@pytest.mark.gpu
def test_device_pacbio_mapper_equals_oracle_on_the_fixture(pacbio_runs, expected):
    """The four runs through the device's mapPacBio profile (long-read probe kernel, strip-tiled 9PacBio DP): equal to the oracle compiled
    with -DORC_PACBIO fill by fill, hence to the floors, the table and the property; the truth-window fills on the device equal the oracle's."""
    from bbmap_amd import msa as M
    from bbmap_amd.index import DeviceIndex, PROFILE_PACBIO
    from bbmap_amd.mapper import Mapper
    from tests.mapper_check import compare
    ref = phix_reference()
    di = DeviceIndex.build([ref], profile=PROFILE_PACBIO)
    tw = expected["pacbio_truth_window"]
    for name, r in pacbio_runs.items():
        recs, blob, bs, ki, _ = r["inputs"]
        mp = Mapper.from_records(di, recs, blob, bs, ki, paired=False, max_sites=32, profile=PROFILE_PACBIO, msaMaxColumns=PACBIO_MSA["msaMaxColumns"])
        mp.step()
        out, st = mp.fetch(), mp.stats()
        mp.close()
        assert st["reads_overflowed"] == 0
        bad = compare(out, r["out"], len(recs), False)
        assert not bad, name + "\n" + "\n".join(bad[:20])
        _check_pacbio(name, out, expected, tw)
    di.close()
    al = M.MultiStateAligner9PacBio(160, 7600)
    for which in (1, 2):
        probs = [(rd, ref.tobytes(), a, b, floor) for rd, a, b, floor in truth_window_jobs_pacbio(which)]
        got = al.align(probs, M.FILL_AND_SCORE_LIMITED)
        for g, t in zip(got, tw["sample%d" % which]):
            assert (None if g["score"] is None else g["score"][:3]) == t
    al.ctx.close()


@pytest.mark.gpu
def test_device_mapper_equals_oracle_on_the_fixture(runs, expected):
    """All six runs through bbmap_map_batch_device: equal to the oracle fill by fill, hence to the table, the truth and the property;
    the truth-window fills on the device equal the oracle's."""
    from bbmap_amd import msa as M
    from bbmap_amd.index import DeviceIndex
    from bbmap_amd.mapper import Mapper
    from tests.mapper_check import compare
    ref = phix_reference()
    di = DeviceIndex.build([ref], k=13)
    tw = expected["truth_window"]
    for name, r in runs.items():
        recs, blob, bs, ki, paired = r["inputs"]
        mp = Mapper.from_records(di, recs, blob, bs, ki, paired=paired, max_sites=32, finalStage=0)
        mp.step()
        out, st = mp.fetch(), mp.stats()
        mp.close()
        assert st["reads_overflowed"] == 0
        bad = compare(out, r["out"], len(recs), paired)
        assert not bad, name + "\n" + "\n".join(bad[:20])
        _check_truth(name, out)
        _check_table(name, out, expected)
        _check_truth_window(name, out, tw)
        # the whole flow: final records and match strings equal the oracle's, hence the table's and the floors on what BBMap prints
        mp = Mapper.from_records(di, recs, blob, bs, ki, paired=paired, max_sites=32)
        mp.step()
        out, st = mp.fetch(), mp.stats()
        mp.close()
        assert st["final_fills"] > 50
        bad = compare(out, r["final"], len(recs), paired)
        assert not bad, name + " (final stage)\n" + "\n".join(bad[:20])
        _check_final(name, out, expected)
    di.close()
    msa = M.MultiStateAligner11ts(maxRows=601, maxColumns=3000)
    for which in (1, 2):
        probs = [(rd, ref.tobytes(), a, b, floor) for rd, a, b, floor in truth_window_jobs(which)]
        got = msa.align(probs, M.FILL_AND_SCORE_LIMITED)
        for g, t in zip(got, tw["sample%d" % which]):
            assert (None if g["score"] is None else g["score"][:3]) == t
    msa.ctx.close()
