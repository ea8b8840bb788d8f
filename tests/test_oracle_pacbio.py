"""CPU checks of the PacBio variant of the DP restatement (oracle/liboracle_pacbio.so): the constants and the
hand-derivable values of current/align2/MultiStateAligner9PacBio.java."""
import numpy as np

from oracle.oracle import OracleMSA, lib_pacbio


def test_pacbio_tables_and_offsets():
    L = lib_pacbio()
    # calcDelScoreOffset / calcInsScoreOffset (:2254-2310), 9 score-offset bits
    assert L.orc_calc_del_score_offset(1) == -292 * 512
    assert L.orc_calc_del_score_offset(5) == (-292 - 4 * 37) * 512
    assert L.orc_calc_del_score_offset(30) == (-292 - 4 * 37 - 15 * 17 - 10 * 2) * 512
    assert L.orc_calc_del_score_offset(90) == (-292 - 4 * 37 - 15 * 17 - 60 * 2 - ((90 - 80 + 3) // 4) * 1) * 512
    assert L.orc_calc_ins_score_offset(1) == -205 * 512
    assert L.orc_calc_ins_score_offset(25) == (-205 - 4 * 42 - 15 * 23 - 5 * 8) * 512


def test_pacbio_column_zero_follows_the_constructor():
    om = OracleMSA(40, 50, scheme="9pacbio")
    W = 51
    col0 = np.ctypeslib.as_array(om.s.packed, shape=(3 * 41 * W,))[:41 * W:W] // 512
    exp = [0, -205]
    for i in range(2, 41):
        exp.append(exp[-1] + (-42 if i < 5 else (-23 if i < 20 else -8)))     # :91-98, tiers by `i<LIMIT`
    assert col0.tolist() == exp


def test_pacbio_perfect_and_one_substitution():
    om = OracleMSA(120, 160, scheme="9pacbio")
    rng = np.random.default_rng(4)
    g = bytes(rng.choice(list(b"ACGT"), 300).astype(np.uint8))
    rd = g[100:200]
    r = om.fillUnlimited(rd, g, 90, 215)
    assert r == [100, 110, 0, 90 + 99 * 100]
    assert om.traceback(rd, g, 90, 215, r[0], r[1], r[2]) == b"m" * 100
    bad = bytearray(rd)
    bad[50] = ord("A") if bad[50] != ord("A") else ord("C")
    r = om.fillUnlimited(bytes(bad), g, 90, 215)
    # 50 matches (90 + 49*100), SUB after a streak > 1 (-137), then match restarts at 90 and 48 more at 100
    assert r[3] == (90 + 49 * 100) - 137 + 90 + 48 * 100
