"""GPU parity tests: HIP BandedAligner (through the C ABI) vs the CPU oracle, bit for bit."""
import random

import pytest

from bbmap_amd import banded as B
from oracle.oracle import banded_align
from tests.problems import rand_seq, mutate

pytestmark = pytest.mark.gpu

COMP = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}


def revcomp(s):
    return bytes(COMP.get(b, b) for b in reversed(s))


def make_problems(seed, n, width, lens=(30, 80, 150, 400), with_n=True):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        L = rng.choice(lens)
        a = rand_seq(rng, L)
        b = mutate(rng, a, max_events=5, n_prob=0.15 if with_n else 0.0)
        if rng.random() < 0.15:
            b = rand_seq(rng, max(5, L + rng.randint(-10, 10)))       # unrelated: early exit
        if rng.random() < 0.3:
            b = b + rand_seq(rng, rng.randint(1, 30))                  # length mismatch -> swap rule
        d = rng.randrange(4)
        me = rng.choice([0, 1, 2, 3, 5, 8, 12, 20, 40])
        ex = rng.random() < 0.5
        q, r = a, b
        if d in (1, 3):
            r = revcomp(b)
        if d == 0:
            qs, rs = rng.choice([0, 0, 0, 3]), rng.choice([0, 0, 0, 2])
        elif d == 1:
            qs, rs = len(q) - 1 - rng.choice([0, 0, 4]), rng.choice([0, 0, 3])
        elif d == 2:
            qs, rs = len(q) - 1 - rng.choice([0, 0, 2]), len(r) - 1 - rng.choice([0, 0, 5])
        else:
            qs, rs = rng.choice([0, 0, 3]), len(r) - 1 - rng.choice([0, 0, 2])
        out.append((d, q, r, qs, rs, me, ex))
    return out


def check(problems, width, semantics):
    al = B.BandedAligner(width, semantics)
    got = al.align_batch(problems)
    for k, (p, g) in enumerate(zip(problems, got)):
        d, q, r, qs, rs, me, ex = p
        e, rv = banded_align(d, q, r, qs, rs, me, ex, al.maxWidth, semantics)
        ctx = "job %d dir=%d qlen=%d rlen=%d qs=%d rs=%d maxEdits=%d exact=%s" % (k, d, len(q), len(r), qs, rs, me, ex)
        assert int(g["status"]) == 0, ctx
        assert int(g["edits"]) == e, ctx
        assert [int(g[f]) for f in ("lastQueryLoc", "lastRefLoc", "lastRow", "lastEdits", "lastOffset")] == rv, ctx
    al.close()


def test_known_answer_from_survey():
    al = B.BandedAligner(11, B.SEMANTICS_JNI_C)
    g = al.align_batch([(0, b"ACGTTGCAAGCTTAGGCTTA", b"ACGTTGCAGCTTAGGCTTAC", 0, 0, 5, True)])[0]
    assert int(g["edits"]) == 2
    assert [int(g[f]) for f in ("lastQueryLoc", "lastRefLoc", "lastRow", "lastEdits", "lastOffset")] == [19, 18, 19, 2, 1]


@pytest.mark.parametrize("semantics", [B.SEMANTICS_JNI_C, B.SEMANTICS_JAVA])
@pytest.mark.parametrize("width", [3, 11, 21, 33, 63, 101, 201])
def test_random_problems_all_directions(width, semantics):
    check(make_problems(1000 + width, 400, width), width, semantics)


def test_short_and_degenerate_sequences():
    probs = [(0, b"A", b"A", 0, 0, 0, True), (0, b"A", b"C", 0, 0, 3, True), (2, b"ACG", b"ACG", 2, 2, 1, True),
             (0, b"NNNN", b"ACGT", 0, 0, 2, False), (0, b"NNNN", b"ACGT", 0, 0, 2, True),
             (1, b"ACGTN", b"NACGT", 4, 0, 2, False), (3, b"acgt", b"ACGT", 0, 3, 2, True),
             (0, b"ACGT" * 300, b"ACGT" * 300, 0, 0, 10, True)]
    for sem in (0, 1):
        check(probs, 21, sem)


def test_quadruple_matches_reference_orchestration():
    """alignQuadruple (BandedAligner.java:39-48) built from the four batched kernels."""
    rng = random.Random(9)
    pairs = []
    for _ in range(60):
        a = rand_seq(rng, rng.choice([60, 120]))
        b = mutate(rng, a, max_events=3, n_prob=0.0)
        if rng.random() < 0.5:
            b = revcomp(b)
        pairs.append((a, b))
    al = B.BandedAligner(21, B.SEMANTICS_JAVA)
    got = al.alignQuadruple(pairs, 8, False)
    for (q, r), g in zip(pairs, got):
        a = banded_align(0, q, r, 0, 0, 8, False, 21, 1)[0]
        b = banded_align(2, q, r, len(q) - 1, len(r) - 1, 8, False, 21, 1)[0]
        me2 = min(8, max(a, b))
        if me2 == 0:
            exp = 0
        else:
            c = banded_align(1, q, r, len(q) - 1, 0, me2, False, 21, 1)[0]
            d = banded_align(3, q, r, 0, len(r) - 1, me2, False, 21, 1)[0]
            exp = min(max(a, b), max(c, d))
        assert g == exp


def _quad_oracle(q, r, maxEdits, exact, width, variant):
    a = banded_align(0, q, r, 0, 0, maxEdits, exact, width, variant)[0]
    b = banded_align(2, q, r, len(q) - 1, len(r) - 1, maxEdits, exact, width, variant)[0]
    me2 = min(maxEdits, max(a, b))
    if me2 == 0:
        return 0
    c = banded_align(1, q, r, len(q) - 1, 0, me2, exact, width, variant)[0]
    d = banded_align(3, q, r, 0, len(r) - 1, me2, exact, width, variant)[0]
    return min(max(a, b), max(c, d))


def test_progressive_and_double_match_reference_orchestration():
    """alignQuadrupleProgressive (BandedAligner.java:24-37) and alignDouble (:50-55) behind the C ABI, both semantics."""
    rng = random.Random(11)
    pairs = []
    for _ in range(80):
        a = rand_seq(rng, rng.choice([40, 90, 200]))
        b = mutate(rng, a, max_events=rng.choice([0, 2, 6, 14]), n_prob=0.0)
        if rng.random() < 0.5:
            b = revcomp(b)
        pairs.append((a, b))
    for sem in (B.SEMANTICS_JNI_C, B.SEMANTICS_JAVA):
        width = 41
        al = B.BandedAligner(width, sem)
        mw = al.maxWidth
        for minE, maxE, exact in ((1, 20, False), (2, 9, True), (10, 10, False)):
            got = al.alignQuadrupleProgressive(pairs, minE, maxE, exact)
            for (q, r), g in zip(pairs, got):
                mx = min(maxE, max(len(q), len(r)))
                i, me, exp = min(minE, mx), -1, mx
                while me < mx:
                    me = min(i, mx)
                    if me * 2 > mx:
                        me = mx
                    e = _quad_oracle(q, r, me, exact, mw, sem)
                    if e < me:
                        exp = e
                        break
                    i *= 4
                assert g == exp, (sem, minE, maxE, exact, q, r)
        got = al.alignDouble(pairs, 12, False)
        for (q, r), g in zip(pairs, got):
            a = banded_align(0, q, r, 0, 0, 12, False, mw, sem)[0]
            exp = 0 if a == 0 else min(a, banded_align(1, q, r, len(q) - 1, 0, a, False, mw, sem)[0])
            assert g == exp
        al.close()
