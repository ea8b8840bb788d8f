"""libbbtoolsjni.so, the JNI shim: it compiles (against jni/jni_min.h, there is no JDK here), exports every symbol the reference's
library exports, its Java_* layer behaves under a mock JNIEnv (jni/mock_jni_test.cpp: same answers as the plain layer, and no
JNI call or GPU wait inside a critical region), and BBMerge's three host natives equal the oracle's restatement."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bbmap_amd")

REFERENCE_SYMBOLS = [        # jni/align2_MultiStateAligner11tsJNI.h:164-174, jni/align2_BandedAlignerJNI.h:17-41, jni/jgi_BBMergeOverlapper.h:21-43
    "Java_align2_MultiStateAligner11tsJNI_fillUnlimitedJNI", "Java_align2_MultiStateAligner11tsJNI_fillLimitedXJNI",
    "Java_align2_BandedAlignerJNI_alignForwardJNI", "Java_align2_BandedAlignerJNI_alignForwardRCJNI",
    "Java_align2_BandedAlignerJNI_alignReverseJNI", "Java_align2_BandedAlignerJNI_alignReverseRCJNI",
    "Java_jgi_BBMergeOverlapper_mateByOverlapJNI", "Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI_1WithQualities",
    "Java_jgi_BBMergeOverlapper_mateByOverlapRatioJNI",
    "Java_jgi_BBMergeOverlapper_mateByOverlapJNI_WithQualities",     # the name the reference's C file defines (jni/BBMergeOverlapper.c:389)
]


@pytest.fixture(scope="module")
def shim():
    from bbmap_amd import build as hip_build
    hip_build.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "jni"), "-s"])
    return os.path.join(PKG, "libbbtoolsjni.so")


def test_exports_every_symbol_of_the_reference_library(shim):
    out = subprocess.check_output(["nm", "-D", "--defined-only", shim], text=True)
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = [s for s in REFERENCE_SYMBOLS if s not in have]
    assert not missing, missing


def test_mock_jnienv_bbmerge_natives(shim):
    out = subprocess.run([os.path.join(PKG, "mock_jni_test"), "merge"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout


def test_bbmerge_natives_equal_the_oracle(shim):
    from oracle.oracle import lib as orc_lib
    O = orc_lib()
    S = C.CDLL(shim)
    rng = np.random.default_rng(5)
    i8, f32, i32 = C.POINTER(C.c_int8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    for fn in (S.bbmerge_mate_by_overlap, O.orc_bbmerge_mate_by_overlap):
        fn.argtypes = [i8, C.c_int, i8, C.c_int, i8, i8, f32, f32, i32] + [C.c_int] * 7
    for fn in (S.bbmerge_mate_by_overlap_ratio, O.orc_bbmerge_mate_by_overlap_ratio):
        fn.argtypes = [i8, C.c_int, i8, C.c_int, i32, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5
    for fn in (S.bbmerge_mate_by_overlap_ratio_with_qualities, O.orc_bbmerge_mate_by_overlap_ratio_q):
        fn.argtypes = [i8, C.c_int, i8, C.c_int, i8, i8, f32, f32, i32, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 3
    acgt = np.frombuffer(b"ACGT", np.int8)
    hits = 0
    for trial in range(400):
        alen, blen = int(rng.integers(50, 160)), int(rng.integers(50, 160))
        ov = int(rng.integers(10, min(alen, blen)))
        a = acgt[rng.integers(0, 4, alen)].copy()
        b = acgt[rng.integers(0, 4, blen)].copy()
        if trial % 5:
            b[:ov] = a[alen - ov:]                                    # b continues a's tail: a real overlap
            for _ in range(int(rng.integers(0, 4))):
                b[int(rng.integers(0, ov))] = acgt[int(rng.integers(0, 4))]
        if trial % 7 == 0:
            b[int(rng.integers(0, blen))] = ord("N")
        aq = rng.integers(2, 41, alen).astype(np.int8)
        bq = rng.integers(2, 41, blen).astype(np.int8)
        cap = max(alen, blen) + 1
        p = lambda x, t: x.ctypes.data_as(t)

        def run(f_overlap, f_ratio, f_ratio_q):
            res = []
            ap, bp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
            rv = np.zeros(5, np.int32)
            res.append((f_ratio(p(a, i8), alen, p(b, i8), blen, p(rv, i32), 8, 12, 35, 35, 0.075, 2.0, 0.55, 0.65, 0.95), rv.tolist()))
            res.append((f_ratio_q(p(a, i8), alen, p(b, i8), blen, p(aq, i8), p(bq, i8), p(ap, f32), p(bp, f32), p(rv, i32), 8, 12, 35, 35,
                                  0.075, 2.0, 0.55), rv.tolist(), ap[:alen].tolist()))
            res.append((f_overlap(p(a, i8), alen, p(b, i8), blen, p(aq, i8), p(bq, i8), p(ap, f32), p(bp, f32), p(rv, i32), 8, 14, 35, 2, 3, 3,
                                  10), rv.tolist()))
            return res
        got = run(S.bbmerge_mate_by_overlap, S.bbmerge_mate_by_overlap_ratio, S.bbmerge_mate_by_overlap_ratio_with_qualities)
        exp = run(O.orc_bbmerge_mate_by_overlap, O.orc_bbmerge_mate_by_overlap_ratio, O.orc_bbmerge_mate_by_overlap_ratio_q)
        assert got == exp, trial
        hits += got[0][0] > 0
    assert hits > 100                                                 # the planted overlaps are found: insert sizes, not just -1


def test_bbmerge_known_answers_derived_by_hand(shim):
    """Known answers worked out BY HAND from the reference's C text (jni/BBMergeOverlapper.c), not from any implementation: the pin
    for the three BBMerge natives and for the oracle's restatement of them (the reference holds no fixture for this path).

    Reads: a = AAAACGGCAC, b = CGGCAC TTTT: b's first six bases are a's last six, so the true insert is 10 + 10 - 6 = 14; the tail
    of b (T) occurs nowhere in a, and a's shifted self-comparisons match in at most one column.

    mateByOverlapRatio(a, b, minOverlap0 4, minOverlap 4, minInsert0 10, minInsert 10, maxRatio .4, margin 2, offset .5, gIncr 1, bIncr 1)
      findBestRatio (:116-161): bestRatio .4001, halfmax .2, inserts 16..10, overlapLength 20 - insert, badlimit bestRatio * length.
        16: GCAC/CGGC   G.C bad 1, C.G bad 2 > 1.6004 -> skipped      15: GGCAC/CGGCA  bad 1, good 1, bad 2, bad 3 > 2.0005 -> skipped
        14: CGGCAC/CGGCAC good 6 bad 0; ratio (0 + .5) / 6 = .08333 < bestRatio; good >= 4 and ratio < halfmax -> return .08333
      x < maxRatio, so maxRatio = .08333; altBadlimit = .08333 * 2 * 10 + 1 = 2.667; margin2 = 2.5 / 10.
      main loop (:332-382), badlimit = min(2.667, min(bestRatio, maxRatio) * 2 * length):
        16: limit .667, first column G.C bad 1 -> skipped; 15: limit .833, bad 1 -> skipped
        14: limit 1.0, good 6 bad 0 (6 > 4 but not < 4: no early exit); ratio .08333 < 1 * 2; ambig = (.1667 >= 1) or (6 < 4) = false;
            bestInsert 14, bestBad 0, bestRatio .08333
        13..10: limits 1.167, 1.333, 1.5, 1.667; columns A.C bad 1, then (C.G | A.G) bad 2 -> skipped each time
      end: bestRatio > maxRatio is false (the same float expression), so insert 14, rvector[2] = 0, rvector[4] = 0.
    With b[2] = T (b = CGTCACTTTT) the true overlap has one mismatch:
      findBestRatio: 16, 15 skipped as above (the changed column only adds bad); 14: good 5 bad 1 <= 2.4006, ratio 1.5 / 6 = .25,
        not < halfmax: continue with bestRatio .25; 13: limit 1.75, A.C, C.G -> 2, skipped; 12: limit 2.0, A.C, A.G (2 <= 2), C.T 3,
        skipped; 11: 2.25 and 10: 2.5: A.C, A.G, A.T -> 3, skipped.  Returns .25 = the new maxRatio; altBadlimit 6, margin2 .25.
      main loop: 16: limit 2.0: G.C, C.G, A.T -> 3 skipped; 15: limit 2.5: bad 1, good, C.T 2, A.C 3 skipped; 14: limit 3: good 5 bad 1,
        ratio .25 < 2, ambig = (.5 >= 1) or (5 < 4) = false, best = (14, bad 1, .25); 13: limit 3.5: A.C C.G G.T G.C -> 4 skipped;
        12: limit 4: A.C A.G C.T G.C (4 <= 4) G.A 5 skipped; 11: limit 4.5: 3 bad, C.C good, G.A 4, G.C 5 skipped; 10: limit 5: six
        mismatching columns, skipped.  End: .25 > .25 false -> insert 14, rvector[2] = 1, rvector[4] = 0.
    With b = TTTTTTTTTT nothing overlaps: every insert of findBestRatio meets two mismatches within its limit (1.6 .. 4.0) -- it returns
      .4001 >= maxRatio, and mateByOverlapRatio returns -1 with rvector[2] = min(alen, blen) = 10, rvector[4] = 0 (:312-316).
    mateByOverlapRatio_WithQualities with every quality 40: probCorrect[40] = 1, so every column weighs 1 and the three traces are
      the same (its own exit test is x > maxRatio, :238); aprob / bprob come back as 1.0.
    mateByOverlap(a, b, q 40, minOverlap0 4, minOverlap 4, minInsert0 10, margin 1, maxMismatches0 2, maxMismatches 2, minq 10) (:23-113):
      bestBad 2, maxOverlap 20 - 10 = 10, minprob probCorrect[10] = .9 < 1 so every column counts; overlaps 4..9 compare a's tail with
      b's head, badlim = bestBad + 1:   4: GCAC/CGGC bad 3 good 1, 6 < 1 false;   5: GGCAC/CGGCA good 1 bad 4, false;
      6: good 6 bad 0: 0 < 6, 6 > 4, 0 < 2 -> winner, bestBad - bad = 2 is not < margin so not ambiguous; 7, 8, 9: badlim 1, two
      mismatches at once.  End: bestBad 0 > 2 - 1 false -> returns 10 + 10 - 6 = 14, rvector[2] = 0, rvector[4] = 0."""
    from oracle.oracle import lib as orc_lib
    O = orc_lib()
    S = C.CDLL(shim)
    i8, f32, i32 = C.POINTER(C.c_int8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    arr = lambda s: np.frombuffer(s, np.int8).copy()
    p = lambda x, t: x.ctypes.data_as(t)
    a = arr(b"AAAACGGCAC")
    q40 = np.full(10, 40, np.int8)
    cases = [(b"CGGCACTTTT", 14, 0), (b"CGTCACTTTT", 14, 1), (b"TTTTTTTTTT", -1, 10)]
    for L, names in ((S, ("bbmerge_mate_by_overlap_ratio", "bbmerge_mate_by_overlap_ratio_with_qualities", "bbmerge_mate_by_overlap")),
                     (O, ("orc_bbmerge_mate_by_overlap_ratio", "orc_bbmerge_mate_by_overlap_ratio_q", "orc_bbmerge_mate_by_overlap"))):
        f_ratio, f_ratio_q, f_overlap = (getattr(L, n) for n in names)
        f_ratio.argtypes = [i8, C.c_int, i8, C.c_int, i32, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 5
        f_ratio_q.argtypes = [i8, C.c_int, i8, C.c_int, i8, i8, f32, f32, i32, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 3
        f_overlap.argtypes = [i8, C.c_int, i8, C.c_int, i8, i8, f32, f32, i32] + [C.c_int] * 7
        for bseq, insert, bad in cases:
            b = arr(bseq)
            rv = np.array([7, 7, 7, 7, 7], np.int32)
            assert f_ratio(p(a, i8), 10, p(b, i8), 10, p(rv, i32), 4, 4, 10, 10, 0.4, 2.0, 0.5, 1.0, 1.0) == insert, (names[0], bseq)
            assert rv.tolist() == [7, 7, bad, 7, 0]
            rv[:] = 7
            ap, bp = np.zeros(11, np.float32), np.zeros(11, np.float32)
            assert f_ratio_q(p(a, i8), 10, p(b, i8), 10, p(q40, i8), p(q40, i8), p(ap, f32), p(bp, f32), p(rv, i32), 4, 4, 10, 10, 0.4, 2.0, 0.5) == insert
            assert rv.tolist() == [7, 7, bad, 7, 0] and (ap[:10] == 1.0).all() and (bp[:10] == 1.0).all()
        rv = np.array([7, 7, 7, 7, 7], np.int32)
        ap, bp = np.zeros(11, np.float32), np.zeros(11, np.float32)
        b = arr(b"CGGCACTTTT")
        assert f_overlap(p(a, i8), 10, p(b, i8), 10, p(q40, i8), p(q40, i8), p(ap, f32), p(bp, f32), p(rv, i32), 4, 4, 10, 1, 2, 2, 10) == 14
        assert rv.tolist() == [7, 7, 0, 7, 0]


GLUE_SYMBOLS = [            # natives of jni/java/align2/*.java, implemented in jni/hip_glue.c
    "Java_align2_MultiStateAligner11tsHIP_create", "Java_align2_MultiStateAligner11tsHIP_destroy",
    "Java_align2_MultiStateAligner11tsHIP_alignBatch", "Java_align2_MultiStateAligner11tsHIP_alignGappedBatch",
    "Java_align2_BBIndexHIP_build", "Java_align2_BBIndexHIP_destroy", "Java_align2_BBIndexHIP_setMaxReadLen", "Java_align2_BBIndexHIP_findBatch",
    "Java_align2_BBMapHIP_create", "Java_align2_BBMapHIP_destroy", "Java_align2_BBMapHIP_mapBatch", "Java_align2_BBMapHIP_getFinal", "Java_align2_BBMapHIP_lastError",
]


def test_glue_library_exports_every_native_the_java_classes_declare(shim):
    """jni/java/align2/*.java declare `private static native` methods; each must resolve to a symbol of libbbmap_amd_jni.so with the
    JNI-mangled name, and the library must declare nothing the classes do not use."""
    import re
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(PKG, "libbbmap_amd_jni.so")], text=True)
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip() and ln.split()[-1].startswith("Java_")}
    declared = set()
    jdir = os.path.join(ROOT, "jni", "java", "align2")
    for f in sorted(os.listdir(jdir)):
        src = open(os.path.join(jdir, f)).read()
        cls = f[:-5]
        assert re.search(r"^package align2;", src, re.M) and ("final class " + cls) in src
        assert 'System.loadLibrary("bbmap_amd_jni")' in src
        for m in re.finditer(r"private static native \S+ (\w+)\(", src):
            declared.add("Java_align2_%s_%s" % (cls, m.group(1)))
    assert declared == have == set(GLUE_SYMBOLS), (declared ^ have)


@pytest.mark.gpu
def test_mock_jnienv_glue_natives(shim):
    """alignBatch / findBatch / mapBatch through the JNI layer (direct buffers) equal the C ABI, and reads map to their origins"""
    out = subprocess.run([os.path.join(PKG, "mock_jni_test"), "glue"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    print(out.stdout)


@pytest.mark.gpu
def test_mock_jnienv_fills_and_banded(shim):
    out = subprocess.run([os.path.join(PKG, "mock_jni_test"), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout


@pytest.mark.gpu
def test_mock_jnienv_32_mapping_threads_share_launches(shim):
    """32 threads call the two fill symbols concurrently: every result, plane and limit vector equals the solo run,
    no JNI call inside a critical region, and calls are combined into far fewer launches; prints the calls/s of 1 and 32 threads."""
    out = subprocess.run([os.path.join(PKG, "mock_jni_test"), "threads", "32"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    print(out.stdout)


@pytest.mark.gpu
def test_mock_jnienv_threads_with_a_tiny_batch_capacity(shim):
    """The same with room for only 3 calls per device batch (BBMSA_LEGACY_BATCH): callers that find the open batch full wait for the
    next one, or take the full one to the device themselves -- every call still gets its own result."""
    env = dict(os.environ, BBMSA_LEGACY_BATCH="3")
    out = subprocess.run([os.path.join(PKG, "mock_jni_test"), "threads", "12"], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr + out.stdout
