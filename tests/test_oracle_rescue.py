"""CPU tests of the rescue-scan oracle (AbstractMapThread.quickRescue restated): hand-derived cases."""
from oracle.oracle import quick_rescue
from tests.rescue_problems import make_problems


def test_quick_rescue_known_cases():
    ref = b"N" * 20 + b"ACGTTGCAAGCTTAGGCTTAACGGATCCGATTACAGGCTAAGCTTCGATCGGATATCGGCTAGCTAGGCTTAAGG" * 3 + b"N" * 20
    read = ref[60:100]
    r = quick_rescue(read, ref, 20, 30, 200, True, 58, 10)
    # exact copy at 60; the unit repeats every 75 bases, so 135 also matches: the scan stops at idealStart + |60 - 58|
    assert r == dict(start=60, stop=99, score=70 + 100 * 39, mismatches=0, perfect=1, semiperfect=1, contig=0)
    r = quick_rescue(read, ref, 20, 200, 200, False, 140, 10)      # searching left from 200: 135 comes first and is nearer
    assert r["start"] == 135 and r["perfect"] == 1
    bad = bytearray(read); bad[10] = ord("N"); bad[30] = ord("A") if bad[30] != ord("A") else ord("C")
    r = quick_rescue(bytes(bad), ref, 20, 30, 200, True, 58, 10)
    # two mismatches (the N and the substitution); runs of 10 and 19 complete before a mismatch, the trailing 9 do not count
    assert (r["start"], r["mismatches"], r["contig"], r["perfect"], r["semiperfect"]) == (60, 2, 19, 0, 0)
    assert r["score"] == 70 + 100 * (40 - 1 - 2)
    # the reference starts with minMismatches = maxAllowedMismatches + 1 and accepts "<=": one more than "allowed" passes
    assert quick_rescue(bytes(bad), ref, 20, 30, 200, True, 58, 1)["mismatches"] == 2
    assert quick_rescue(bytes(bad), ref, 20, 30, 200, True, 58, 0) is None
    assert quick_rescue(read[:9], ref, 20, 30, 200, True, 58, 3) is None            # reads shorter than 10 are not rescued


def test_quick_rescue_generated_set_is_consistent():
    ref, probs = make_problems(3, 200)
    found = 0
    for b, ch, loc, sd, right, ideal, mam in probs:
        r = quick_rescue(b, ref, 0, loc, sd, right, ideal, mam)
        if r is None:
            continue
        found += 1
        mm = sum(1 for j in range(len(b)) if b[j] != ref[r["start"] + j] or b[j] == ord("N"))
        assert mm == r["mismatches"] <= mam + 1
        lo = max(0, loc) if right else max(0, loc - sd)
        hi = min(len(ref) - len(b), loc + sd) if right else min(len(ref) - len(b), loc)
        assert lo <= r["start"] <= hi
    assert found > 80
