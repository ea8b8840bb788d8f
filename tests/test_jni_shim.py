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
