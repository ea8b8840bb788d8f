"""CPU tests of the BandedAligner oracle (oracle/banded_oracle.c)."""
import json
import os
import random

from oracle.oracle import banded_align

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_known_answers.json")))


def test_known_answer_from_survey_c_semantics():
    g = GOLD["bandedAlignForward"]
    e, rv = banded_align(0, g["query"].encode(), g["ref"].encode(), g["qstart"], g["rstart"], g["maxEdits"],
                         bool(g["exact"]), g["maxWidth"], variant=0)
    assert e == g["edits"] and rv == g["returnVals"]


def test_identical_sequences_have_zero_edits_in_every_direction():
    rng = random.Random(3)
    s = bytes(rng.choice(b"ACGT") for _ in range(80))
    for variant in (0, 1):
        assert banded_align(0, s, s, 0, 0, 5, True, 21, variant)[0] == 0
        assert banded_align(2, s, s, 79, 79, 5, True, 21, variant)[0] == 0
        rc = bytes({65: 84, 67: 71, 71: 67, 84: 65}[b] for b in reversed(s))
        assert banded_align(1, s, rc, 79, 0, 5, True, 21, variant)[0] == 0
        assert banded_align(3, s, rc, 0, 79, 5, True, 21, variant)[0] == 0


def test_edit_counts_and_semantic_differences():
    q = b"ACGTTGCAAGCTTAGGCTTAACCGTTAGCA"
    r = b"ACGTTGCAAGGTTAGGCTTAACCGTTAGCA"      # one substitution
    for variant in (0, 1):
        e, rv = banded_align(0, q, r, 0, 0, 4, True, 9, variant)
        assert e == 1 and rv[2] == len(q) - 1
    # the C adds the off-centre distance, the Java clamps to it: an indel shows the difference
    r2 = b"ACGTTGCAGCTTAGGCTTAC"
    q2 = b"ACGTTGCAAGCTTAGGCTTA"
    assert banded_align(0, q2, r2, 0, 0, 5, True, 11, 0)[0] == 2
    assert banded_align(0, q2, r2, 0, 0, 5, True, 11, 1)[0] == 1


def test_swap_rule_is_symmetric_in_locations():
    rng = random.Random(5)
    a = bytes(rng.choice(b"ACGT") for _ in range(60))
    b = a[:50]
    e1, rv1 = banded_align(0, a, b, 0, 0, 6, True, 13, 1)      # query longer: swapped internally
    e2, rv2 = banded_align(0, b, a, 0, 0, 6, True, 13, 1)
    assert e1 == e2 and rv1[0] == rv2[1] and rv1[1] == rv2[0]
