"""GPU parity for SURVEY D11 (the MultiStateAligner9PacBio parameter set): the HIP path through the C ABI against the
restatement compiled with the PacBio constants (oracle/liboracle_pacbio.so).  Parity is pinned by restatement only: the
reference holds no known answers for this class and there is no JVM to produce any."""
import random

import numpy as np
import pytest

from bbmap_amd import msa as M
from oracle.oracle import OracleMSA
from tests.problems import rand_seq
from tests.test_msa_gpu import oracle_align

pytestmark = pytest.mark.gpu


def pacbio_problems(seed, n, lo=60, hi=700):
    """Reads with PacBio-like errors (indel-rich) against a padded window of their origin."""
    rng = random.Random(seed)
    genome = rand_seq(rng, 6000)
    out = []
    for _ in range(n):
        L = rng.randint(lo, hi)
        s = rng.randint(50, len(genome) - L - 200)
        rd = bytearray(genome[s:s + L])
        i = 0
        err = rng.choice([0.0, 0.03, 0.12, 0.2])
        while i < len(rd):
            x = rng.random()
            if x < err * 0.45:
                rd.insert(i, ord(rng.choice("ACGT"))); i += 2
            elif x < err * 0.8 and len(rd) > 30:
                del rd[i]
            elif x < err:
                rd[i] = ord(rng.choice("ACGTN")); i += 1
            else:
                i += 1
        rd = bytes(rd[:hi + 40])
        pad = rng.choice([4, 8, 30])
        a = max(0, s - pad)
        b = min(len(genome) - 1, s + L + pad + rng.randint(0, 25))
        maxq = 90 + 100 * (len(rd) - 1)
        ms = rng.choice([0, int(maxq * 0.3), int(maxq * 0.56), int(maxq * 0.8), maxq - 50])
        out.append((rd, genome, a, b, ms))
    return out


@pytest.fixture(autouse=True, params=["sequential", "pipelined"])
def strip_form(request, monkeypatch):
    """Every test runs under both forms of the strip kernel: one wavefront per job (what a full batch gets) and the strips of a job
    pipelined over several wavefronts (what a launch with few jobs gets; BBMSA_STRIP_PIPE_JOBS is the job-count threshold, read when
    the context is created)."""
    monkeypatch.setenv("BBMSA_STRIP_PIPE_JOBS", "0" if request.param == "sequential" else "512")
    return request.param


def check(problems, flags, maxRows=800, maxColumns=1100):
    al = M.MultiStateAligner9PacBio(maxRows, maxColumns)
    got = al.align(problems, flags)
    om = OracleMSA(maxRows, maxColumns, scheme="9pacbio")
    n_null = 0
    for k, (p, g) in enumerate(zip(problems, got)):
        exp = oracle_align(om, p[0], p[1], p[2], p[3], p[4], flags)
        ctx = "job %d rows=%d cols=%d ms=%d" % (k, len(p[0]), p[3] - p[2] + 1, p[4])
        if exp["result"] is not None:
            assert g["result"] == exp["result"], ctx
        else:
            assert g["status"] == M.ST_NULL, ctx
            n_null += 1
        assert g["status"] == exp["status"], ctx
        assert g["iterations"] == exp["iterations"], ctx
        assert g["fill_kind"] == exp["fill_kind"], ctx
        assert g["score"] == exp["score"], ctx
        assert g["match"] == exp["match"], ctx
    al.ctx.close()
    return n_null


def test_pacbio_perfect_read_scores_90_plus_100_per_base():
    rng = random.Random(3)
    g = rand_seq(rng, 900)
    rd = g[200:500]
    al = M.MultiStateAligner9PacBio(400, 500)
    r = al.align([(rd, g, 190, 520, 20000)], M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)[0]
    assert r["result"] == [300, 310, 0, 90 + 100 * 299, 0] and r["iterations"] == 49168
    assert r["match"] == b"m" * 300
    r = al.align([(rd, g, 190, 520, 0)], M.FILL_UNLIMITED_RAW)[0]
    assert r["result"][:4] == [300, 310, 0, 90 + 100 * 299] and r["iterations"] == 300 * 331
    al.ctx.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_pacbio_fill_score_traceback(seed):
    probs = pacbio_problems(seed, 120)
    n_null = check(probs, M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK)
    assert 0 < n_null < len(probs)


def test_pacbio_raw_modes():
    probs = pacbio_problems(9, 80, lo=40, hi=400)
    check(probs, M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)
    check(probs, M.FILL_UNLIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)


def test_pacbio_long_read():
    """One 3 kb read: beyond anything the 11ts kernels take (640 rows)."""
    rng = random.Random(5)
    g = rand_seq(rng, 5000)
    rd = bytearray(g[700:3700])
    for pos in range(100, 2900, 97):
        del rd[pos]
    rd = bytes(rd)
    check([(rd, g, 680, 3760, int((90 + 100 * (len(rd) - 1)) * 0.5))], M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK,
          maxRows=3100, maxColumns=3300)


def test_pacbio_gapped_reference_jobs():
    """fillAndScoreLimited(..., gaps) with the PacBio scheme: makeGref on the device, fill on the gapped reference in the
    one-job-per-thread kernel, coordinates translated back (MultiStateAligner9PacBio.java:110-124, 1403-1470)."""
    rng = random.Random(41)
    ref = rand_seq(rng, 7000)
    probs = []
    for i in range(14):
        L = rng.choice([300, 450, 600])
        st = rng.randrange(200, 2500)
        cut = rng.randrange(80, L - 80)
        dl = rng.choice([300, 500, 900, 1500])
        rd = bytearray(ref[st:st + cut] + ref[st + cut + dl: st + dl + L])
        for pos in range(20 + i, len(rd) - 5, 53):            # PacBio-like: a deletion every ~50 bases
            del rd[pos]
        stop = st + dl + L - 1
        gaps = [st, st + cut - 1 + rng.randint(0, 3), st + cut + dl - rng.randint(0, 3), stop]
        ms = int(rng.choice([0.3, 0.45]) * (90 + 100 * (len(rd) - 1)))
        probs.append((bytes(rd), ref, st - 8, stop + 8, ms, gaps))
    probs.append((probs[0][0], ref, probs[0][2], probs[0][3], probs[0][4], None))
    al = M.MultiStateAligner9PacBio(maxRows=640, maxColumns=2400)
    got = al.alignGapped(probs)
    om = OracleMSA(640, 2400, scheme="9pacbio")
    nonnull = 0
    for p, g in zip(probs, got):
        sv, mx = om.fillAndScoreLimited(p[0], p[1], p[2], p[3], p[4], p[5])
        assert g["status"] != M.ST_BAD_SHAPE
        assert g["score"] == sv, (p[2:], g, sv)
        if sv is not None:
            nonnull += 1
            tb = om.traceback(p[0], p[1], max(0, p[2]), min(len(p[1]) - 1, p[3]), mx[0], mx[1], mx[2], gapped=p[5] is not None)
            assert g["match"] == tb
    al.ctx.close()
    assert nonnull > 6


def _pacbio_read(rng, genome, s, L, err):
    rd = bytearray(genome[s:s + L])
    i = 0
    while i < len(rd):
        x = rng.random()
        if x < err * 0.45:
            rd.insert(i, ord(rng.choice("ACGT"))); i += 2
        elif x < err * 0.8 and len(rd) > 30:
            del rd[i]
        elif x < err:
            rd[i] = ord(rng.choice("ACGT")); i += 1
        else:
            i += 1
    return bytes(rd)


def test_pacbio_full_size_reads_cross_many_strips():
    """mapPacBio's own limits (ALIGN_ROWS 6020, ALIGN_COLUMNS 7600, current/align2/BBMapThreadPacBio.java:27-28): reads of up to
    6,019 bases with 13-17 % PacBio errors against windows of up to 7,600 columns -- twelve 512-row strips, the boundary row
    handed through HBM, the traceback walking back across every strip border -- limited, Java-gated and unlimited fills."""
    rng = random.Random(17)
    genome = rand_seq(rng, 9000)
    probs = []
    for L, err, pad in ((6019, 0.15, 40), (5200, 0.13, 700), (4097, 0.17, 8), (513, 0.1, 30), (512, 0.0, 4), (1025, 0.15, 300)):
        s = rng.randint(800, 1200)
        rd = _pacbio_read(rng, genome, s, L, err)[:6019]
        a, b = s - pad, min(len(genome) - 1, s + L + pad)
        if b - a + 1 > 7600:
            b = a + 7599
        maxq = 90 + 100 * (len(rd) - 1)
        probs.append((rd, genome, a, b, int(maxq * 0.3)))
    check(probs, M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK, maxRows=6019, maxColumns=7600)
    check(probs[:2], M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, maxRows=6019, maxColumns=7600)
    check(probs[2:4], M.FILL_UNLIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, maxRows=6019, maxColumns=7600)


def test_pacbio_windows_narrower_than_the_read():
    """A read with more inserted than deleted bases spans fewer reference bases than it has rows: cells then need deletions AND
    insertions to finish, and each plane applies its own priority (jni-less class: MultiStateAligner9PacBio.fillLimited).  The strip
    kernel takes these windows itself (the one-thread kernel needs seconds for a 6,000-base piece)."""
    rng = random.Random(29)
    genome = rand_seq(rng, 9000)
    probs = []
    for i, (L, short) in enumerate([(1400, 3), (1400, 40), (1400, 300), (900, 12), (2600, 90), (700, 699 - 60)]):
        rd = _pacbio_read(rng, genome, 1000, L, 0.12)
        a = 1000 - 8
        b = a + len(rd) - 1 - short                                    # columns = rows - short
        maxq = 90 + 100 * (len(rd) - 1)
        probs.append((rd, genome, a, b, int(maxq * (0.15 if i % 2 else 0.3))))
    check(probs, M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK, maxRows=2700, maxColumns=3000)
    check(probs, M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, maxRows=2700, maxColumns=3000)
    check(probs[:4], M.FILL_UNLIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, maxRows=2700, maxColumns=3000)


def test_pacbio_fill_dies_inside_a_strip_and_a_narrow_window():
    rng = random.Random(23)
    genome = rand_seq(rng, 6000)
    junk = rand_seq(rng, 1500)                       # unrelated read: the limited fill runs out of good cells early
    rd = _pacbio_read(rng, genome, 900, 1400, 0.1)
    maxq = 90 + 100 * 1399
    probs = [(junk, genome, 100, 1800, int(0.5 * (90 + 100 * 1499))),
             (rd, genome, 890, 2320, int(maxq * 0.4)),
             (rd, genome, 890, 2200, int(maxq * 0.2)),          # window narrower than the read by more than two columns
             (rd[:700], genome, 890, 1630, int(0.99 * (90 + 100 * 699)))]
    check(probs, M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, maxRows=1600, maxColumns=2000)


def test_pipelined_form_survives_hand_shake_timeouts(monkeypatch, strip_form):
    """The pipelined form assumes the wavefronts of a slot are co-resident; when one is not, its partners time out.  With a spin limit of
    a few polls nearly every hand-shake times out: slots die, their jobs are claimed one by one and handed to the one-thread kernel, no
    job is lost or answered twice, and every result equals the sequential kernel's (= the oracle's)."""
    if strip_form != "pipelined":
        pytest.skip("the hand-shakes exist in the pipelined form only")
    monkeypatch.setenv("BBMSA_PIPE_SPIN_LIMIT", "3")
    probs = pacbio_problems(21, 40, lo=900, hi=1500)           # two or three strips per read
    n_null = check(probs, M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK, maxRows=1600, maxColumns=1800)
    assert n_null < len(probs)
    monkeypatch.setenv("BBMSA_PIPE_SPIN_LIMIT", "2000")        # some hand-shakes make it, some do not: slots die in the middle of a job
    check(pacbio_problems(22, 40, lo=900, hi=1500), M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK, maxRows=1600, maxColumns=1800)
