"""CPU tests: the oracle (oracle/msa11ts_oracle.c) against the reference's known answers and
against hand-derived cases.  No GPU, no reference checkout needed."""
import json
import os
import random

import pytest

from oracle.oracle import OracleMSA, score_no_indels, score_no_indels_match, calc_affine_score, lib
from tests.problems import survey_problem_stream, max_quality

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")))


def test_known_answer_fill_unlimited():
    g = GOLD["fillUnlimited_34x67"]
    m = OracleMSA()
    res, it = m.fill_unlimited_raw(g["read"].encode(), g["ref"].encode(), g["refStartLoc"], g["refEndLoc"])
    assert res == g["result"]
    assert it == g["iterations"]
    # hand check from SURVEY 8c: 34 matches (70+33*100) - one deletion (472) - the MATCH2->MATCH downgrade after it (30)
    assert res[3] == 3370 - 472 - 30


def _limited_known_problem():
    random.seed(7)
    g = "".join(random.choice("ACGT") for _ in range(400)).encode()
    rd = bytearray(g[100:250])
    rd[40] = ord("A") if rd[40] != ord("A") else ord("C")
    del rd[90:92]
    return bytes(rd), g


def test_known_answer_fill_limited():
    k = GOLD["fillLimitedX_148x158"]
    rd, g = _limited_known_problem()
    assert int(0.56 * max_quality(len(rd))) - 120 == k["minScore"]
    m = OracleMSA()
    res, it = m.fill_limited_raw(rd, g, k["refStartLoc"], k["refEndLoc"], k["minScore"])
    assert res == k["result"]
    assert it == k["iterations"]
    # Java-side walkers on the same matrix: one substitution at read 40, a 2-base deletion after read 90
    m.s.rows, m.s.columns = 148, 158
    tb = m.traceback(rd, g, 96, 253, 148, 154, 0)
    assert tb == b"m" * 40 + b"S" + b"m" * 49 + b"DD" + b"m" * 58
    assert m.score(rd, g, 96, 253, 148, 154, 0) == [13978, 100, 249, 148, 154, 0]


def test_survey_400_problem_statistic():
    """400 problems, dirty == clean matrix, visited fraction mean .535 min .215 max .850."""
    k = GOLD["stale_matrix_experiment_400"]
    dirty = OracleMSA()
    n, diff, frac = 0, 0, []
    for rd, G, a, b, ms, bw, bwr in survey_problem_stream():
        dirty.s.bandwidth, dirty.s.bandwidthRatio = bw, bwr
        r_dirty = dirty.fill_limited_raw(rd, G, a, b, ms)
        clean = OracleMSA(bandwidth=bw, bandwidthRatio=bwr)
        r_clean = clean.fill_limited_raw(rd, G, a, b, ms)
        n += 1
        frac.append(r_clean[1] / (len(rd) * (b - a + 1)))
        diff += r_dirty != r_clean
    assert n == k["problems"] and diff == k["dirty_vs_clean_differences"]
    assert "%.3f" % (sum(frac) / len(frac)) == k["visited_fraction_mean"]
    assert "%.3f" % min(frac) == k["visited_fraction_min"]
    assert "%.3f" % max(frac) == k["visited_fraction_max"]


def test_tables_match_java_static_init():
    L = lib()
    ins, insc, sub = L.orc_points_ins_array(), L.orc_points_ins_array_c(), L.orc_points_sub_array()
    assert [ins[i] for i in (1, 2, 5, 6, 20, 21, 603)] == [-395, -39, -39, -23, -23, -8, -8]
    assert [sub[i] for i in (1, 2, 5, 6, 603)] == [-127, -51, -51, -25, -25]
    assert insc[1] == -395 and insc[5] == -395 - 4 * 39 and insc[20] == -395 - 4 * 39 - 15 * 23
    off = L.orc_pointsoff_ins_array_c()
    assert off[20] == insc[20] * 2048
    assert L.orc_calc_del_score_offset(1) == -472 * 2048
    assert L.orc_calc_del_score_offset(90) == (-472 - 4 * 33 - 15 * 9 - 60 - 3) * 2048
    b2n = L.orc_base_to_number()
    assert [b2n[ord(c)] for c in "ACGTUacgtuN-"] == [0, 1, 2, 3, 3, 0, 1, 2, 3, 3, -1, -1]


def test_perfect_read_scores_max_quality():
    rng = random.Random(5)
    ref = bytes(rng.choice(b"ACGT") for _ in range(600))
    for L0 in (60, 100, 150, 250):
        rd = ref[200:200 + L0]
        m = OracleMSA()
        sv, mx = m.fillAndScoreLimited(rd, ref, 196, 200 + L0 + 3, int(0.56 * max_quality(L0)))
        assert sv[:3] == [max_quality(L0), 200, 200 + L0 - 1]
        assert m.traceback(rd, ref, 196, 200 + L0 + 3, mx[0], mx[1], mx[2]) == b"m" * L0
        assert score_no_indels(rd, ref, 200) == max_quality(L0)


def test_below_min_returns_null_and_unshifted_score():
    rng = random.Random(9)
    ref = bytes(rng.choice(b"ACGT") for _ in range(600))
    rd = bytes(rng.choice(b"ACGT") for _ in range(150))      # unrelated read
    m = OracleMSA()
    res, it = m.fill_limited_raw(rd, ref, 100, 260, int(0.7 * max_quality(150)))
    assert res[4] == 1 and res[3] < 0 and res[3] % 2048 == 0    # left unshifted on failure
    assert 0 < it < 150 * 161
    assert m.fillLimited(rd, ref, 100, 260, int(0.7 * max_quality(150))) is None


def test_java_gate_falls_back_to_unlimited():
    rng = random.Random(11)
    ref = bytes(rng.choice(b"ACGT") for _ in range(300))
    rd = ref[100:140]                                          # rows+cols < 90
    m = OracleMSA()
    out = m.fillLimited(rd, ref, 98, 141, 1000)
    assert out == [40, 42, 0, max_quality(40)]
    assert m.iterationsUnlimited == 40 * 44 and m.iterationsLimited == 0
    m2 = OracleMSA()
    m2.fillLimited(ref[50:200], ref, 46, 203, 0)               # minScore < 1
    assert m2.iterationsUnlimited == 150 * 158


def test_score_no_indels_and_match_string():
    ref = b"NNACGTACGTTTGACCANN"
    rd = b"ACGTACCTTTGACNA"
    s, ms = score_no_indels_match(rd, ref, 2)
    #            A C G T A C C T T T G A C N A
    assert ms == b"mmmmmmSmmmmmmNm"
    # a read N costs 0 and does not leave match mode (MultiStateAligner11tsJNI.java:1154-1156), so the last base earns MATCH2
    assert s == 70 + 5 * 100 - 127 + 70 + 5 * 100 + 0 + 100
    assert score_no_indels(rd, ref, 2) == s
    assert score_no_indels_match(rd, ref, 10)[0] == -99999


def test_calc_affine_score_simple():
    n = 20
    loc = [101] * n            # locArray holds the implied read start per base (BBIndex.extendScore)
    bs = [0] * n
    assert calc_affine_score(loc, bs) == max_quality(n)
    loc2 = list(loc)
    loc2[7] = -1                                                # one substitution
    assert calc_affine_score(loc2, bs) == 70 + 6 * 100 - 127 + 70 + 11 * 100
