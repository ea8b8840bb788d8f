"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import random

import numpy as np
import pytest

from bbmap_amd import msa as M
from oracle.oracle import OracleMSA
from tests.problems import mixed_problems, survey_problem_stream, max_quality, rand_seq

pytestmark = pytest.mark.gpu


def oracle_align(om, read, ref, a, b, ms, flags):
    """What the reference would produce for one job with these flags (oracle restatement)."""
    mode = flags & 7
    out = {"score": None, "match": None, "status": 0}
    if flags & M.CLAMP_WINDOW:
        a = max(0, a)
        b = min(len(ref) - 1, b)
    it_l0, it_u0 = om.iterationsLimited, om.iterationsUnlimited
    if mode == M.FILL_LIMITED_RAW:
        res, _ = om.fill_limited_raw(read, ref, a, b, ms)
        om.s.rows, om.s.columns = len(read), b - a + 1
        null = res[4] == 1
    elif mode == M.FILL_UNLIMITED_RAW:
        res, _ = om.fill_unlimited_raw(read, ref, a, b)
        res = res + [0]
        om.s.rows, om.s.columns = len(read), b - a + 1
        null = False
    else:
        r4 = om.fillLimited(read, ref, a, b, ms)
        null = r4 is None
        res = None if null else r4 + [0]
        if null:
            out["status"] = 1
    out["iterations"] = (om.iterationsLimited - it_l0) + (om.iterationsUnlimited - it_u0)
    out["fill_kind"] = 1 if om.iterationsUnlimited != it_u0 else 0
    out["result"] = res
    if not null:
        if flags & M.DO_SCORE:
            out["score"] = om.score(read, ref, a, b, res[0], res[1], res[2])
        if flags & M.DO_TRACEBACK:
            out["match"] = om.traceback(read, ref, a, b, res[0], res[1], res[2])
    return out


def check_batch(problems, flags, maxRows=601, maxColumns=3000, bandwidth=0, bandwidthRatio=0.0, **kw):
    al = M.MultiStateAligner11ts(maxRows, maxColumns, bandwidth, bandwidthRatio, **kw)
    got = al.align(problems, flags)
    om = OracleMSA(maxRows, maxColumns, bandwidth, bandwidthRatio)
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


ALL = M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK


def test_known_answers_from_survey():
    ref = b"NNNNACGTTGCAAGCTTAGGCTTACGGATCCGATTACAGGCATTAGCCGTAAGCTTGCAATGCNNNN"
    read = b"GCTTAGGCTTACGGATCGATTACAGGCATTAGCC"
    al = M.MultiStateAligner11ts()
    g = al.align([(read, ref, 0, 66, 0)], M.FILL_UNLIMITED_RAW)[0]
    assert g["result"][:4] == [34, 48, 0, 2868] and g["iterations"] == 2278
    random.seed(7)
    gen = "".join(random.choice("ACGT") for _ in range(400)).encode()
    rd = bytearray(gen[100:250])
    rd[40] = ord("A") if rd[40] != ord("A") else ord("C")
    del rd[90:92]
    g = al.align([(bytes(rd), gen, 96, 253, 8151)], M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)[0]
    assert g["result"] == [148, 154, 0, 13978, 0] and g["iterations"] == 12688
    assert g["score"] == [13978, 100, 249, 148, 154, 0]
    assert g["match"] == b"m" * 40 + b"S" + b"m" * 49 + b"DD" + b"m" * 58


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mixed_problems_fill_score_traceback(seed):
    probs = mixed_problems(seed, 300)
    n_null = check_batch(probs, ALL)
    assert 0 < n_null < len(probs)


def test_raw_limited_and_unlimited_modes():
    probs = mixed_problems(17, 200)
    check_batch(probs, M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)
    check_batch(probs, M.FILL_UNLIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)


def test_survey_stream_including_banded():
    """The survey's 400-problem stream; one third of it is banded (bw=40, bwr=0.18)."""
    groups = {}
    for rd, G, a, b, ms, bw, bwr in survey_problem_stream():
        groups.setdefault((bw, bwr), []).append((rd, G, a, b, ms))
    assert len(groups) == 2
    for (bw, bwr), probs in groups.items():
        check_batch(probs, M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK, bandwidth=bw, bandwidthRatio=bwr)


@pytest.mark.parametrize("lanes,maxRows", [(16, 160), (32, 160), (64, 160), (32, 320), (64, 601), (16, 48), (64, 64)])
def test_every_lane_geometry(lanes, maxRows):
    lens = tuple(x for x in (40, 60, 100, 150, 250, 300, 500, 600) if x <= maxRows)[-3:]
    probs = mixed_problems(100 + lanes + maxRows, 120, read_lens=lens, ref_len=1600)
    probs = [p for p in probs if len(p[0]) <= maxRows]
    check_batch(probs, ALL, maxRows=maxRows, maxColumns=1024, lanes_per_job=lanes)


def test_generic_kernel_handles_wide_windows():
    # fast_cols=128 forces every wider window through the per-thread generic kernel
    probs = mixed_problems(23, 80)
    check_batch(probs, ALL, fast_cols=128)


def test_edge_cases():
    rng = random.Random(4)
    ref = rand_seq(rng, 500)
    probs = []
    probs.append((ref[100:101], ref, 100, 100, 0))                      # 1 x 1
    probs.append((ref[100:103], ref, 98, 106, 0))                        # tiny
    probs.append((b"N" * 50, ref, 100, 160, 100))                        # all-N read
    probs.append((ref[200:350], b"N" * 500, 190, 360, 100))              # all-N reference
    probs.append((ref[200:350], ref, -5, 170, 5000))                     # window clipped at the left end
    probs.append((ref[400:500], ref, 390, 520, 3000))                    # window clipped at the right end
    gapped = ref[:250] + b"-" * 3 + ref[250:]
    probs.append((ref[150:350], gapped, 140, 365, 8000))                 # gap symbols in the reference
    probs.append((ref[150:300].lower(), ref, 146, 303, 2000))            # lower-case read never matches
    probs.append((ref[150:300], ref, 146, 303, max_quality(150)))        # perfect, minScore = max
    probs.append((ref[150:300], ref, 146, 303, max_quality(150) + 121))  # unreachable minScore
    check_batch(probs, ALL)
    check_batch(probs[:4] + probs[6:], M.FILL_LIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)
    check_batch(probs[:4] + probs[6:], M.FILL_UNLIMITED_RAW | M.DO_SCORE | M.DO_TRACEBACK)


def test_bad_shape_reported_not_computed():
    rng = random.Random(8)
    ref = rand_seq(rng, 800)
    ctx = M.MSAContext(maxRows=100, maxColumns=200)
    jobs, reads, refs = M.pack_problems([(ref[100:250], ref, 96, 253, 100), (ref[100:150], ref, 96, 400, 100),
                                         (ref[100:150], ref, 96, 160, 100)], M.FILL_LIMITED)
    res, _ = ctx.align_batch(jobs, reads, refs)
    assert res["status"].tolist() == [M.ST_BAD_SHAPE, M.ST_BAD_SHAPE, M.ST_OK]


def test_full_size_batch_properties():
    """BASELINE-sized shapes (150-bp reads, window = read + 2*4 pad): size-independent properties."""
    rng = random.Random(77)
    ref = rand_seq(rng, 200000)
    probs = []
    for _ in range(20000):
        st = rng.randrange(100, len(ref) - 300)
        probs.append((ref[st:st + 150], ref, st - 4, st + 153, int(0.56 * max_quality(150))))
    al = M.MultiStateAligner11ts(maxRows=160, maxColumns=256)
    got = al.align(probs, ALL)
    for p, g in zip(probs, got):
        assert g["score"][0] == max_quality(150)                # a perfect read scores maxQuality
        assert g["score"][1] == p[2] + 4 and g["score"][2] == p[2] + 4 + 149
        assert g["match"] == b"m" * 150
    # idempotence: a second pass gives identical records
    got2 = al.align(probs, ALL)
    assert got == got2


def test_gapped_reference_jobs():
    """fillAndScoreLimited(..., gaps) + traceback(gapped=True): makeGref on the device, fill on the gapped reference,
    coordinates translated back (MultiStateAligner11tsJNI.java:116-128, :362-372, :499-531, :668-801)."""
    import random
    from oracle.oracle import OracleMSA
    rng = random.Random(77)
    ref = bytes(rng.choice(b"ACGT") for _ in range(6000))
    probs = []
    for i in range(60):
        L = rng.choice([100, 150, 150, 200])
        st = rng.randrange(200, 2000)
        cut = rng.randrange(30, L - 30)
        dl = rng.choice([300, 400, 700, 1500, 2600])          # long deletion: the read skips `dl` reference bases
        rd = bytearray(ref[st:st + cut] + ref[st + cut + dl: st + dl + L])
        for _ in range(rng.randint(0, 3)):
            rd[rng.randrange(L)] = rng.choice(b"ACGT")
        if i % 7 == 3:
            del rd[10:12]                                      # plus a short deletion near the start
        stop = st + dl + L - 1
        # gap array as BBIndex.makeGapArray produces it: {start, end of first block, start of second block, stop}
        gaps = [st, st + cut - 1 + rng.randint(0, 3), st + cut + dl - rng.randint(0, 3), stop]
        if i % 5 == 4:                                         # three blocks
            mid = st + cut + dl + 20
            if stop - mid > 400:
                gaps = gaps[:3] + [mid, stop - 20, stop]
                gaps[4] = max(gaps[4], gaps[3] + 300)
                if gaps[4] >= stop:
                    gaps = gaps[:3] + [stop]
        ms = int(rng.choice([0.3, 0.5, 0.56]) * (70 + 100 * (len(rd) - 1)))
        probs.append((bytes(rd), ref, st - 4, stop + 4, ms, gaps))
    probs.append((probs[0][0], ref, probs[0][2], probs[0][3], probs[0][4], None))     # an ungapped job in the same batch
    al = M.MultiStateAligner11ts(maxRows=224, maxColumns=1600)
    got = al.alignGapped(probs)
    om = OracleMSA(224, 1600)
    nonnull = 0
    for p, g in zip(probs, got):
        sv, mx = om.fillAndScoreLimited(p[0], p[1], p[2], p[3], p[4], p[5])
        assert g["status"] != M.ST_BAD_SHAPE
        assert g["score"] == sv, (p[2:], g, sv)
        if sv is not None:
            nonnull += 1
            tb = om.traceback(p[0], p[1], max(0, p[2]), min(len(p[1]) - 1, p[3]), mx[0], mx[1], mx[2], gapped=p[5] is not None)
            assert g["match"] == tb
    assert nonnull > 30


def _legacy_cases(rng, ref, n):
    for i in range(n):
        L = rng.choice([60, 100, 150])
        st = rng.randrange(50, 2500)
        rd = bytearray(ref[st:st + L + 10])
        if i % 3 == 1:
            rd[rng.randrange(L)] = ord("N")
            rd[rng.randrange(L)] = rng.choice(b"ACGT")
        if i % 3 == 2:
            del rd[L // 2:L // 2 + rng.randint(1, 6)]
        rd = bytes(rd[:L])
        a, b = st - 4, st + L + 8
        if i % 11 == 7:
            b = a + L - 6                                   # narrower than the read: handed to the one-thread kernel
        limited = i % 4 != 3
        yield rd, a, b, limited, int(0.5 * (70 + 100 * (L - 1)))


@pytest.mark.parametrize("band", [(0, 0.0), (40, 0.18)])
def test_legacy_packed_matrix_feeds_the_java_walkers(band):
    """bbmsa_fill_submit / _collect: the planes land in the Java layout.  Three checks per fill: (1) every cell the oracle's
    restatement of the native fill WROTE holds the same int on our side (score, time bits and the subfloor of visited-but-bad cells);
    (2) score2 / traceback2 (the oracle's restatement of the Java walkers, which read `packed`) run on OUR matrix and give what they
    give on the oracle's own; (3) vertLimit / horizLimit come back as the native code leaves them."""
    import ctypes as C
    import random
    import numpy as np
    from oracle.oracle import OracleMSA
    rng = random.Random(91)
    maxRows, maxCols = 160, 300
    ctx = M.MSAContext(maxRows=maxRows, maxColumns=maxCols, bandwidth=band[0], bandwidthRatio=band[1], legacy=True)
    ref = bytes(rng.choice(b"ACGT") for _ in range(3000))
    n_int = 3 * (maxRows + 1) * (maxCols + 1)
    MARK = 0x5a5a5a5a
    done = 0
    for rd, a, b, limited, ms in _legacy_cases(rng, ref, 48):
        om = OracleMSA(maxRows, maxCols, bandwidth=band[0], bandwidthRatio=band[1])
        view = np.ctypeslib.as_array(om.s.packed, shape=(3, maxRows + 1, maxCols + 1))
        pristine = view.copy()                              # row 0 / column 0 as the constructor leaves them
        view[:, 1:, 1:] = MARK
        if limited:
            exp, exp_it = om.fill_limited_raw(rd, ref, a, b, ms)
        else:
            exp, exp_it = om.fill_unlimited_raw(rd, ref, a, b)
            exp = exp + [0]
        packed = pristine.copy().reshape(-1)
        got, it, vl, hl = ctx.fill_packed(rd, ref, a, b, ms, limited, packed, limits=True)
        assert got[:4] == exp[:4] and (not limited or got[4] == exp[4]) and it == exp_it
        rows, cols = len(rd), b - a + 1
        ours = packed.reshape(3, maxRows + 1, maxCols + 1)
        o, g = view[:, 1:rows + 1, 1:cols + 1], ours[:, 1:rows + 1, 1:cols + 1]
        wrote = o != MARK
        assert wrote.sum() > rows                           # (the oracle did fill something)
        # "not a score": subfloor (pruned, below the limit, or a row-end sentinel) and the BADoff the native fill spreads over the last
        # row first (:398-403).  Nothing reads the time bits of such a cell, and the native code itself strips them wherever a
        # sentinel lands on a computed cell; we keep subfloor there.  Every other cell -- every real score and its time -- is exact.
        maxGain = (rows - 1) * 100 + 70
        subfloor = ((ms << 11) - (maxGain << 11) - 5 * (100 << 11)) if limited else -2 * (maxGain << 11)
        badoff = (-(1 << 20) + 2000) << 11
        dead = wrote & (((o & ~2047) == subfloor) | (o == badoff))
        live = wrote & ~dead
        assert live.sum() > rows
        assert (g[live] == o[live]).all()
        assert (((g[dead] & ~2047) == subfloor) | (g[dead] == badoff)).all()
        assert (ours[:, 0, :] == pristine[:, 0, :]).all() and (ours[:, :, 0] == pristine[:, :, 0]).all()
        assert (ours[:, rows + 1:, :] == pristine[:, rows + 1:, :]).all() and (ours[:, 1:, cols + 1:] == pristine[:, 1:, cols + 1:]).all()
        if limited:
            assert vl.tolist() == np.ctypeslib.as_array(om.s.vertLimit, shape=(maxRows + 1,))[:rows + 1].tolist()
            assert hl.tolist() == np.ctypeslib.as_array(om.s.horizLimit, shape=(maxCols + 1,))[:cols + 1].tolist()
        if limited and exp[4] == 1:
            continue
        want_score = om.score(rd, ref, a, b, exp[0], exp[1], exp[2])
        want_tb = om.traceback(rd, ref, a, b, exp[0], exp[1], exp[2])
        # same walkers, our matrix: overwrite the oracle's packed with the planes the GPU produced
        C.memmove(om.s.packed, packed.ctypes.data, n_int * 4)
        assert om.score(rd, ref, a, b, got[0], got[1], got[2]) == want_score
        assert om.traceback(rd, ref, a, b, got[0], got[1], got[2]) == want_tb
        done += 1
    st = ctx.legacy_stats()
    assert st["calls"] == 48 and st["launches"] == 48 and st["handed_on"] >= 4
    assert done > 24
    ctx.close()


def test_legacy_calls_from_many_threads_share_launches():
    """The per-call entry is thread-safe on ONE context and combines calls that arrive together: 16 threads x 24 fills give what the
    same fills gave alone, in far fewer launches than calls."""
    import random
    import threading
    import numpy as np
    rng = random.Random(17)
    maxRows, maxCols = 160, 300
    ctx = M.MSAContext(maxRows=maxRows, maxColumns=maxCols, legacy=True)
    ref = bytes(rng.choice(b"ACGT") for _ in range(3000))
    cases = list(_legacy_cases(rng, ref, 16 * 24))
    n_int = 3 * (maxRows + 1) * (maxCols + 1)

    def run(idx, out):
        packed = np.zeros(n_int, np.int32)
        for i in idx:
            rd, a, b, limited, ms = cases[i]
            packed[:] = 0
            got, it = ctx.fill_packed(rd, ref, a, b, ms, limited, packed)
            out[i] = (got[:4 + limited], it, hash(packed.tobytes()))
    solo = {}
    run(range(len(cases)), solo)
    before = ctx.legacy_stats()
    together = {}
    th = [threading.Thread(target=run, args=(range(t, len(cases), 16), together)) for t in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    after = ctx.legacy_stats()
    assert together == solo
    assert after["calls"] - before["calls"] == len(cases)
    assert after["launches"] - before["launches"] < len(cases)          # some calls shared a launch (ctypes releases the GIL in the call)
    ctx.close()


def test_wide_pass_takes_windows_beyond_the_first_column_buffer():
    """Windows wider than the first pass's LDS buffer (fast_cols) run in the wide pass (64 lanes per job, buffer as wide as
    maxColumns), not in the one-thread-per-job generic kernel; results are the same bits."""
    import random
    from oracle.oracle import OracleMSA
    rng = random.Random(123)
    ref = bytes(rng.choice(b"ACGT") for _ in range(6000))
    probs = []
    for i in range(40):
        L = rng.choice([100, 150])
        st = rng.randrange(100, 2000)
        rd = bytearray(ref[st:st + L])
        for _ in range(rng.randint(0, 4)):
            rd[rng.randrange(L)] = rng.choice(b"ACGT")
        width = rng.choice([300, 700, 1500, 2600])                  # 300 fits the first pass, the others do not
        probs.append((bytes(rd), ref, st - rng.randrange(0, 50), st + width, int(0.4 * (70 + 100 * (L - 1)))))
    al = M.MultiStateAligner11ts(maxRows=160, maxColumns=3000, fast_cols=512)
    got = al.align(probs)
    om = OracleMSA(160, 3000)
    for p, g in zip(probs, got):
        sv, mx = om.fillAndScoreLimited(*p)
        assert g["score"] == sv
        if sv is not None:
            assert g["match"] == om.traceback(p[0], p[1], max(0, p[2]), p[3], mx[0], mx[1], mx[2])
    counts = al.ctx.last_counts()
    assert counts["generic"] == 0, counts


def test_no_iterations_flag_changes_nothing():
    """BBMSA_NO_ITERATIONS is accepted and has no effect (the tighter-bound shortcut it used to allow was withdrawn: the
    reference's pruning is not admissible, tests/test_oracle_final.py::test_a_tighter_min_score_can_change_a_fill).  Checked
    where many alignments tie (low-complexity sequence: indels inside homopolymers and dinucleotide repeats can be placed in
    several equally good ways): every field, the visited-cell counter included, and the split over the kernels stay the same."""
    import random
    from oracle.oracle import OracleMSA
    rng = random.Random(2024)

    def lowcomplex(n):
        out = bytearray()
        while len(out) < n:
            kind = rng.random()
            if kind < 0.35:
                out += bytes([rng.choice(b"ACGT")]) * rng.randint(3, 12)
            elif kind < 0.6:
                out += bytes(rng.choice(b"ACGT") for _ in range(2)) * rng.randint(2, 8)
            else:
                out += bytes(rng.choice(b"ACGT") for _ in range(rng.randint(4, 30)))
        return bytes(out[:n])

    refs = [lowcomplex(4000), bytes(rng.choice(b"ACGT") for _ in range(4000))]
    probs = []
    for i in range(6000):
        ref = refs[i % 2]
        L = rng.choice([100, 150, 150])
        st = rng.randrange(50, 3500)
        rd = bytearray(ref[st:st + L + 60])
        span = L
        ev = rng.random()
        if ev < 0.45:
            p = rng.randrange(10, L - 10); d = rng.choice([1, 1, 2, 3, 5, 8, 12, 20])
            del rd[p:p + d]; span = L + d
        elif ev < 0.8:
            p = rng.randrange(10, L - 10); d = rng.choice([1, 1, 2, 3, 5, 9])
            rd[p:p] = bytes(rng.choice(b"ACGT") for _ in range(d)); span = L - d
        for _ in range(rng.choice([0, 0, 1, 2, 3])):
            rd[rng.randrange(L)] = rng.choice(b"ACGTN")
        rd = bytes(rd[:L])
        ms = int(rng.choice([0.4, 0.56, 0.56, 0.7]) * (70 + 100 * (L - 1)))
        probs.append((rd, ref, st - 4, st + span + 3, ms))
    al = M.MultiStateAligner11ts(maxRows=160, maxColumns=320)
    flags = M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK
    exact = al.align(probs, flags)
    split_exact = al.ctx.last_counts()
    relaxed = al.align(probs, flags | M.NO_ITERATIONS)
    split_relaxed = al.ctx.last_counts()
    assert split_relaxed == split_exact, (split_exact, split_relaxed)
    for e, r in zip(exact, relaxed):
        assert (r["status"], r["result"], r["score"], r["match"], r["iterations"]) == (e["status"], e["result"], e["score"], e["match"], e["iterations"])
    om = OracleMSA(160, 320)                                             # and the exact run is the oracle's, on a sample
    for p, e in list(zip(probs, exact))[:400]:
        sv, mx = om.fillAndScoreLimited(*p)
        assert e["score"] == sv
