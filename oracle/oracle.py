"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under bbmap_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)


def build(force=False, pacbio=False):
    so = os.path.join(_HERE, "liboracle_pacbio.so" if pacbio else "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _declare_msa(L):
    L.orc_msa_new.restype = C.c_void_p
    L.orc_msa_new.argtypes = [C.c_int, C.c_int]
    L.orc_msa_free.argtypes = [C.c_void_p]
    for name in ("orc_points_ins_array", "orc_pointsoff_ins_array", "orc_points_ins_array_c",
                 "orc_pointsoff_ins_array_c", "orc_points_sub_array", "orc_pointsoff_sub_array"):
        getattr(L, name).restype = c_i32p
    L.orc_base_to_number.restype = c_i8p
    L.orc_calc_del_score_offset.restype = C.c_int32
    L.orc_calc_ins_score_offset.restype = C.c_int32
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _declare_msa(C.CDLL(build()))
    return _LIB


_LIB_PB = None


def lib_pacbio():
    """The same restatements compiled with mapPacBio's classes' constants (-DORC_PACBIO): MultiStateAligner9PacBio, BBIndexPacBio,
    BBMapThreadPacBio / BBMapPacBio.setDefaults."""
    global _LIB_PB
    if _LIB_PB is None:
        _LIB_PB = _declare_msa(C.CDLL(build(pacbio=True)))
    return _LIB_PB


def _u8(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    if a.size == 0:
        a = np.zeros(1, np.uint8)
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


class MSAStruct(C.Structure):
    _fields_ = [("maxRows", C.c_int), ("maxColumns", C.c_int),
                ("packed", c_i32p), ("vertLimit", c_i32p), ("horizLimit", c_i32p),
                ("grefbuffer", c_u8p),
                ("greflimit", C.c_int), ("greflimit2", C.c_int), ("grefRefOrigin", C.c_int),
                ("iterationsLimited", C.c_int64), ("iterationsUnlimited", C.c_int64),
                ("rows", C.c_int), ("columns", C.c_int),
                ("bandwidth", C.c_int), ("bandwidthRatio", C.c_float)]


class OracleMSA:
    """Mirror of align2.MSA's interface over the C restatement."""

    def __init__(self, maxRows=601, maxColumns=3000, bandwidth=0, bandwidthRatio=0.0, scheme="11ts"):
        assert scheme in ("11ts", "9pacbio")
        self.L = lib() if scheme == "11ts" else lib_pacbio()
        self.h = self.L.orc_msa_new(maxRows, maxColumns)
        if not self.h:
            raise MemoryError("orc_msa_new failed")
        self.s = MSAStruct.from_address(self.h)
        self.s.bandwidth = bandwidth
        self.s.bandwidthRatio = bandwidthRatio
        self.maxRows, self.maxColumns = maxRows, maxColumns

    def __del__(self):
        try:
            if self.h:
                self.L.orc_msa_free(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass

    # -- raw C entry points ------------------------------------------------------------
    def fill_unlimited_raw(self, read, ref, a, b):
        r, f = _u8(read), _u8(ref)
        res = np.zeros(4, np.int32)
        it0 = self.s.iterationsUnlimited
        self.L.orc_fill_unlimited_raw(C.c_void_p(self.h), _p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                      a, b, _p(res, c_i32p))
        return res.tolist(), self.s.iterationsUnlimited - it0

    def fill_limited_raw(self, read, ref, a, b, minScore):
        r, f = _u8(read), _u8(ref)
        res = np.zeros(5, np.int32)
        it0 = self.s.iterationsLimited
        self.L.orc_fill_limited_raw(C.c_void_p(self.h), _p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                    a, b, minScore, _p(res, c_i32p))
        return res.tolist(), self.s.iterationsLimited - it0

    # -- Java-level wrappers -----------------------------------------------------------
    def fillLimited(self, read, ref, a, b, minScore, gaps=None):
        r, f = _u8(read), _u8(ref)
        out = np.zeros(4, np.int32)
        g = None if gaps is None else np.asarray(gaps, np.int32).copy()
        ok = self.L.orc_fill_limited(C.c_void_p(self.h), _p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                     a, b, minScore,
                                     None if g is None else _p(g, c_i32p), 0 if g is None else len(g),
                                     _p(out, c_i32p))
        return out.tolist() if ok else None

    def fillUnlimited(self, read, ref, a, b, gaps=None):
        r, f = _u8(read), _u8(ref)
        out = np.zeros(4, np.int32)
        g = None if gaps is None else np.asarray(gaps, np.int32).copy()
        self.L.orc_fill_unlimited(C.c_void_p(self.h), _p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                  a, b, None if g is None else _p(g, c_i32p), 0 if g is None else len(g),
                                  _p(out, c_i32p))
        return out.tolist()

    def traceback(self, read, ref, a, b, row, col, state, gapped=False):
        r, f = _u8(read), _u8(ref)
        cap = row + col + 8 + 128 * 64
        out = np.zeros(cap, np.uint8)
        n = self.L.orc_traceback(C.c_void_p(self.h), _p(r, c_u8p), _p(f, c_u8p), a, b, row, col, state,
                                 1 if gapped else 0, _p(out, c_u8p), cap)
        if n < 0:
            raise RuntimeError("traceback overflow")
        return out[:n].tobytes()

    def score(self, read, ref, a, b, maxRow, maxCol, maxState, gapped=False):
        r, f = _u8(read), _u8(ref)
        out = np.zeros(8, np.int32)
        n = self.L.orc_score(C.c_void_p(self.h), _p(r, c_u8p), _p(f, c_u8p), a, b, maxRow, maxCol, maxState,
                             1 if gapped else 0, _p(out, c_i32p))
        return out[:n].tolist()

    def fillAndScoreLimited(self, read, ref, refStartLoc, refEndLoc, minScore, gaps=None):
        """Returns (score_vec or None, max4 or None)."""
        r, f = _u8(read), _u8(ref)
        out = np.zeros(8, np.int32)
        mx = np.zeros(4, np.int32)
        g = None if gaps is None else np.asarray(gaps, np.int32).copy()
        n = self.L.orc_fill_and_score_limited(C.c_void_p(self.h), _p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                              refStartLoc, refEndLoc, minScore,
                                              None if g is None else _p(g, c_i32p), 0 if g is None else len(g),
                                              _p(out, c_i32p), _p(mx, c_i32p))
        if n == 0:
            return None, None
        return out[:n].tolist(), mx.tolist()

    @property
    def iterationsLimited(self):
        return self.s.iterationsLimited

    @property
    def iterationsUnlimited(self):
        return self.s.iterationsUnlimited

    @property
    def columns(self):
        return self.s.columns


def score_no_indels(read, ref, refStart, baseScores=None):
    L = lib()
    r, f = _u8(read), _u8(ref)
    bs = None if baseScores is None else np.asarray(baseScores, np.int8).copy()
    return L.orc_score_no_indels(_p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                 None if bs is None else _p(bs, c_i8p), refStart)


def score_no_indels_match(read, ref, refStart, baseScores=None):
    L = lib()
    r, f = _u8(read), _u8(ref)
    bs = None if baseScores is None else np.asarray(baseScores, np.int8).copy()
    m = np.zeros(max(1, len(read)), np.uint8)
    s = L.orc_score_no_indels_match(_p(r, c_u8p), len(read), _p(f, c_u8p), len(ref),
                                    None if bs is None else _p(bs, c_i8p), refStart, _p(m, c_u8p))
    return s, m[:len(read)].tobytes()


def calc_affine_score(locArray, baseScores, minContig=0):
    L = lib()
    la = np.asarray(locArray, np.int32).copy()
    bs = np.asarray(baseScores, np.int8).copy()
    return L.orc_calc_affine_score(_p(la, c_i32p), len(la), _p(bs, c_i8p), minContig)


def banded_align(direction, query, ref, qstart, rstart, maxEdits, exact, maxWidth, variant=0):
    """direction: 0 forward, 1 forwardRC, 2 reverse, 3 reverseRC; variant 0 = JNI C, 1 = Java concrete.
    Returns (edits, [lastQueryLoc, lastRefLoc, lastRow, lastEdits, lastOffset])."""
    L = lib()
    q, r = _u8(query), _u8(ref)
    out = np.zeros(5, np.int32)
    e = L.orc_banded_align(direction, _p(q, c_u8p), len(query), _p(r, c_u8p), len(ref), qstart, rstart,
                           maxEdits, 1 if exact else 0, maxWidth, variant, _p(out, c_i32p))
    return e, out.tolist()


# ---------------------------------------------------------------------------------------------- index probe
class IndexParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "k", "chromBits", "minChrom", "maxChrom", "maxIndel", "maxIndel2", "minApproxHitsToKeep", "kfilter",
        "maxUsableLength", "maxUsableLength2", "maxHitsReduction2", "maximumMaxHitsReduction", "hitReductionDiv",
        "quitAfterTwoPerfects", "prescanQscore", "trimByGreedy", "slow",
        "maxAverageListToSearch", "maxAverageListToSearch2", "maxShortestListToSearch")] + [("_pad", C.c_int32 * 0), ("pointsPerSite", C.c_int64)]


class IndexStruct(C.Structure):
    _fields_ = [("p", IndexParams), ("nblocks", C.c_int32), ("starts", C.POINTER(c_i32p)), ("sites", C.POINTER(c_i32p)),
                ("numSites", C.POINTER(C.c_int64)), ("counts", c_i32p), ("lengthHistogram", C.c_int32 * 1001),
                ("nchroms", C.c_int32), ("chromArr", C.POINTER(c_u8p)), ("chromArrLen", c_i32p), ("chromLengths", c_i32p)]


class SiteStruct(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("chrom", "strand", "start", "stop", "hits", "score", "perfect", "semiperfect", "ngaps")] + \
               [("gaps", C.c_int32 * 16)]


def small_genome_tuning(ix_struct, defined_bases):
    """BBMap.loadIndex's genome-size adjustments (current/align2/BBMap.java:367-381), applied before analyzeIndex."""
    p = ix_struct.p
    if defined_bases < 300000000:
        p.maxHitsReduction2 += 1
        p.maximumMaxHitsReduction += 1
        if defined_bases < 30000000:
            p.maximumMaxHitsReduction += 1
            p.hitReductionDiv = max(p.hitReductionDiv - 1, 3)


def _lib_for(profile):
    assert profile in ("bbmap", "pacbio")
    return lib_pacbio() if profile == "pacbio" else lib()


def fraction_to_exclude(defined_bases, profile="bbmap"):
    # FRACTION_GENOME_TO_EXCLUDE: BBIndex.java:3214 (0.03), BBIndexPacBio.java:2509 (0.005); scaled as BBMap.java:367-381 /
    # BBMapPacBio.java:351-365 do
    f = np.float32(0.005 if profile == "pacbio" else 0.03)
    if defined_bases < 30000000:
        return float(f * np.float32(0.5))
    if defined_bases < 100000000:
        return float(f * np.float32(0.6))
    if defined_bases < 300000000:
        return float(f * np.float32(0.75))
    return float(f)


class OracleIndex:
    """Index + probe of the CPU oracle.  chroms: list of bytes-like (chromosome 1..n), already padded with N."""

    def __init__(self, chroms, k=None, chromBits=None, profile="bbmap"):
        self.L = _lib_for(profile)
        self.profile = profile
        if k is None:
            k = 12 if profile == "pacbio" else 13                           # BBMapPacBio.java:51 / BBMap.java:48
        self.chroms = [np.frombuffer(bytes(c), np.uint8).copy() for c in chroms]
        n = len(self.chroms)
        maxlen = max(len(c) for c in self.chroms)
        if chromBits is None:
            chromBits = min(16, (32 - int(maxlen).bit_length()) - 1)        # AUTO_CHROMBITS, BBMap.java:317-321
        self.k, self.chromBits = k, chromBits
        arr = (c_u8p * (n + 1))()
        lens = (C.c_int32 * (n + 1))()
        for i, c in enumerate(self.chroms):
            arr[i + 1] = c.ctypes.data_as(c_u8p)
            lens[i + 1] = len(c)
        defined = sum(int(np.isin(c, np.frombuffer(b"ACGT", np.uint8)).sum()) for c in self.chroms)
        self.defined_bases = defined
        self.L.orc_index_build.restype = C.c_void_p
        self.L.orc_index_build.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(c_u8p), c_i32p, C.c_float]
        self.h = self.L.orc_index_build(k, chromBits, n, arr, lens, fraction_to_exclude(defined, profile))
        self.s = IndexStruct.from_address(self.h)
        small_genome_tuning(self.s, defined)
        self._keep = (arr, lens)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_index_free.argtypes = [C.c_void_p]
                self.L.orc_index_free(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass

    def block_arrays(self, b=0):
        ks = 1 << (2 * self.k)
        starts = np.ctypeslib.as_array(self.s.starts[b], shape=(ks + 1,))
        ns = int(self.s.numSites[b])
        sites = np.ctypeslib.as_array(self.s.sites[b], shape=(max(ns, 1),))[:ns]
        return starts, sites

    def counts(self):
        return np.ctypeslib.as_array(self.s.counts, shape=(1 << (2 * self.k),))

    def find(self, basesP, basesM, baseScoresP, keyScoresP, offsets, cap=64, want_stats=False):
        bp, bm = _u8(basesP), _u8(basesM)
        bs = np.asarray(baseScoresP, np.int8).copy()
        ks = np.asarray(keyScoresP, np.int32).copy()
        of = np.asarray(offsets, np.int32).copy()
        out = (SiteStruct * cap)()
        st = np.zeros(4, np.int64)
        self.L.orc_index_find.argtypes = [C.c_void_p, c_u8p, c_u8p, C.c_int, c_i8p, c_i32p, c_i32p, C.c_int,
                                          C.POINTER(SiteStruct), C.c_int, C.POINTER(C.c_int64)]
        n = self.L.orc_index_find(C.c_void_p(self.h), _p(bp, c_u8p), _p(bm, c_u8p), len(basesP), _p(bs, c_i8p),
                                  _p(ks, c_i32p), _p(of, c_i32p), len(of), out, cap,
                                  st.ctypes.data_as(C.POINTER(C.c_int64)))
        if n < 0:
            raise RuntimeError("site list overflow")
        res = [dict(chrom=o.chrom, strand=o.strand, start=o.start, stop=o.stop, hits=o.hits, score=o.score,
                    perfect=o.perfect, semiperfect=o.semiperfect, gaps=list(o.gaps[:o.ngaps])) for o in out[:n]]
        return (res, st.tolist()) if want_stats else res


class OracleIndexView(OracleIndex):
    """The oracle's probe over index arrays that already exist (a device-built index exported to the host; the GPU tests show
    that build equal to the oracle's own, array by array).  Keeps bench.py's CPU leg and parity sample affordable on a
    3 Gbp reference, where the oracle's single-threaded index build would take longer than the whole benchmark."""

    def __init__(self, chroms, k, chromBits, params, blocks, profile="bbmap"):
        """blocks: list of (starts, sites, counts, lengthHistogram) per block (counts / histogram taken from block 0)."""
        self.L = _lib_for(profile)
        self.profile = profile
        self.chroms = [np.ascontiguousarray(c, np.uint8) for c in chroms]
        self.k, self.chromBits = k, chromBits
        n, nb = len(self.chroms), len(blocks)
        self._blocks = [(np.ascontiguousarray(b[0], np.int32), np.ascontiguousarray(b[1], np.int32)) for b in blocks]
        self._counts = np.ascontiguousarray(blocks[0][2], np.int32)
        self._hist = np.ascontiguousarray(blocks[0][3], np.int32)
        p = IndexParams()
        for name, _ in IndexParams._fields_:
            if name in params:
                setattr(p, name, int(params[name]))
        starts = (c_i32p * nb)(*[b[0].ctypes.data_as(c_i32p) for b in self._blocks])
        sites = (c_i32p * nb)(*[b[1].ctypes.data_as(c_i32p) for b in self._blocks])
        num = (C.c_int64 * nb)(*[len(b[1]) for b in self._blocks])
        arr = (c_u8p * (n + 1))()
        lens = (C.c_int32 * (n + 1))()
        for i, c in enumerate(self.chroms):
            arr[i + 1] = c.ctypes.data_as(c_u8p)
            lens[i + 1] = len(c)
        self.L.orc_index_from_arrays.restype = C.c_void_p
        self.L.orc_index_from_arrays.argtypes = [C.POINTER(IndexParams), C.c_int, C.POINTER(c_i32p), C.POINTER(c_i32p), C.POINTER(C.c_int64),
                                                 c_i32p, c_i32p, C.c_int, C.POINTER(c_u8p), c_i32p]
        self.h = self.L.orc_index_from_arrays(C.byref(p), nb, starts, sites, num, self._counts.ctypes.data_as(c_i32p),
                                              self._hist.ctypes.data_as(c_i32p), n, arr, lens)
        self.s = IndexStruct.from_address(self.h)
        self._keep = (starts, sites, num, arr, lens, p)

    def __del__(self):
        try:
            if self.h:
                self.L.orc_index_free_view.argtypes = [C.c_void_p]
                self.L.orc_index_free_view(C.c_void_p(self.h))
                self.h = None
        except Exception:
            pass


def make_offsets(readlen, k, density=1.9, min_keys=2):
    L = lib()
    out = np.zeros(256, np.int32)
    L.orc_make_offsets.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, c_i32p, C.c_int]
    n = L.orc_make_offsets(readlen, k, density, min_keys, _p(out, c_i32p), 256)
    return out[:n].tolist()


def quick_rescue(bases, ref, minIndex, loc, searchDist, searchRight, idealStart, maxAllowedMismatches,
                 pointsMatch=70, pointsMatch2=100, useAffine=True, baseHitScore=100):
    """AbstractMapThread.quickRescue; returns None or dict(start, stop, score, mismatches, perfect, semiperfect, contig)."""
    b, f = _u8(bases), _u8(ref)
    out = np.zeros(8, np.int32)
    lib().orc_quick_rescue(_p(b, c_u8p), len(bases), _p(f, c_u8p), len(ref), minIndex, loc, searchDist,
                           1 if searchRight else 0, idealStart, maxAllowedMismatches, pointsMatch, pointsMatch2,
                           1 if useAffine else 0, baseHitScore, _p(out, c_i32p))
    if not out[0]:
        return None
    return dict(start=int(out[1]), stop=int(out[2]), score=int(out[3]), mismatches=int(out[4]), perfect=int(out[5]),
                semiperfect=int(out[6]), contig=int(out[7]))


def set_perfect(bases, ref, start, stop):
    """SiteScore.setPerfect(bases) for a site [start, stop] of `ref`; returns (perfect, semiperfect)."""
    b, f = _u8(bases), _u8(ref)
    out = np.zeros(2, np.int32)
    lib().orc_set_perfect(_p(b, c_u8p), len(bases), _p(f, c_u8p), len(ref), start, stop, _p(out, c_i32p))
    return int(out[0]), int(out[1])


# ---------------------------------------------------------------------------------------------- mapper control flow
MSITE_DTYPE = np.dtype([("chrom", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("stop", "<i4"), ("hits", "<i4"),
                        ("quickScore", "<i4"), ("score", "<i4"), ("slowScore", "<i4"), ("pairedScore", "<i4"),
                        ("perfect", "<i4"), ("semiperfect", "<i4"), ("rescued", "<i4"), ("ngaps", "<i4"),
                        ("gaps", "<i4", (16,)), ("match_job", "<i4"), ("reserved", "<i4", (2,))])
MJOB_DTYPE = np.dtype([("read", "<i4"), ("seq", "<i4"), ("kind", "<i4"), ("strand", "<i4"), ("chrom", "<i4"),
                       ("refStartLoc", "<i4"), ("refEndLoc", "<i4"), ("minScore", "<i4"), ("ngaps", "<i4"),
                       ("score_len", "<i4"), ("score", "<i4", (8,)), ("match_len", "<i4"), ("pad_", "<i4"),
                       ("iterations", "<i8")])
assert MSITE_DTYPE.itemsize == 128 and MJOB_DTYPE.itemsize == 88


class MapParams(C.Structure):
    _fields_ = [("minRatio", C.c_float)] + [(n, C.c_int32) for n in (
        "slowAlignPadding", "slowRescuePadding", "extraPadding", "tipSearchDist", "maxPairDist", "averagePairDist",
        "maxRescueDist", "maxRescueMismatches", "maxTrimSitesToRetain", "trimList", "doRescue", "alignColumns",
        "clearzone3", "msaMaxRows", "msaMaxColumns", "finalStage")]


# orc_final: stream.Read's mapping fields after processRead / processReadPair (the final alignment stage, final_stage.inc)
FINAL_DTYPE = np.dtype([(n, "<i4") for n in ("mapped", "chrom", "strand", "start", "stop", "mapScore", "paired", "ambiguous",
                                             "perfect", "rescued", "match_len", "nsites")])
assert FINAL_DTYPE.itemsize == 48


def map_default_params(profile="bbmap", **kw):
    p = MapParams()
    _lib_for(profile).orc_map_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


READ_DTYPE = np.dtype([("bases_off", "<i8"), ("keys_off", "<i8"), ("len", "<i4"), ("nkeys", "<i4")])     # orc_read = bbidx_read


def map_batch(oi, reads1, reads2, L, offsets, key_scores, params=None, cap=64, want_log=True, threads=1, match_stride=0):
    """The mapper control flow (BBMapThread.processRead / processReadPair; params.finalStage = 0 stops after the rescue stage) on
    the CPU oracle, for the uniform case: every read L bases, one set of key offsets / scores.  reads1/reads2: uint8 arrays of
    n*L bases (reads2 None = single-ended).  Returns a dict: sites1/nsites1 (and 2), final1 (and 2) + fmatch1 (and 2), the job
    log (one record per fill, with its traceback string), stats, seconds."""
    r1 = np.ascontiguousarray(reads1, np.uint8).reshape(-1, L)
    n = r1.shape[0]
    paired = reads2 is not None
    if paired:
        r2 = np.ascontiguousarray(reads2, np.uint8).reshape(-1, L)
        bases = np.stack([r1, r2], axis=1).reshape(-1)
    else:
        bases = r1.reshape(-1)
    nr = 2 * n if paired else n
    recs = np.zeros(nr, READ_DTYPE)
    recs["bases_off"] = np.arange(nr, dtype=np.int64) * L
    recs["keys_off"] = 0
    recs["len"] = L
    recs["nkeys"] = len(offsets)
    keyinfo = np.concatenate([np.asarray(offsets, np.int32), np.asarray(key_scores, np.int32)])
    o = map_reads(oi, recs, bases, keyinfo, None, paired, params, cap, want_log, threads, match_stride, jobs_per_read=12)
    out = dict(log=o["log"], match=o["match"], stats=o["stats"], seconds=o["seconds"])
    if paired:
        out.update(sites1=o["sites"][0::2], nsites1=o["nsites"][0::2], sites2=o["sites"][1::2], nsites2=o["nsites"][1::2],
                   final1=o["final"][0::2], final2=o["final"][1::2], fmatch1=o["fmatch"][0::2], fmatch2=o["fmatch"][1::2])
    else:
        out.update(sites1=o["sites"], nsites1=o["nsites"], sites2=None, nsites2=None, final1=o["final"], final2=None,
                   fmatch1=o["fmatch"], fmatch2=None)
    return out


def map_reads(oi, recs, bases, keyinfo, base_scores=None, paired=False, params=None, cap=64, want_log=True, threads=1,
              match_stride=0, jobs_per_read=8):
    """The general form of map_batch: per-read records (READ_DTYPE: where a read's bases / base scores and its keys are, its
    length and key count; keyinfo holds offsets[nkeys] then keyScores[nkeys] per read), as bbmap_map_batch_device takes them.
    Returns a dict: sites (n x cap), nsites, final (FINAL_DTYPE per read: what BBMap prints; zeros when params.finalStage = 0),
    fmatch (n x fstride: the reads' match strings), log, match, stats, seconds."""
    L_ = oi.L
    p = params or map_default_params(getattr(oi, "profile", "bbmap"))
    rc = np.ascontiguousarray(recs, READ_DTYPE)
    n = rc.size
    b = np.ascontiguousarray(bases, np.uint8)
    ki = np.ascontiguousarray(keyinfo, np.int32)
    bs = None if base_scores is None else np.ascontiguousarray(base_scores, np.int8)
    sites = np.zeros((n, cap), MSITE_DTYPE)
    ns = np.zeros(n, np.int32)
    logcap = (n * jobs_per_read + 64) if want_log else 0
    log = np.zeros(max(1, logcap), MJOB_DTYPE)
    stride = match_stride or (p.msaMaxRows + p.msaMaxColumns + 64 if p.msaMaxRows > 1000 else p.msaMaxRows + 700)
    match = np.zeros((max(1, logcap), stride), np.uint8) if want_log else None
    nlog = C.c_int64(0)
    stats = np.zeros(4, np.int64)
    fin = np.zeros(n, FINAL_DTYPE)
    fstride = int(stride)
    fmatch = np.zeros((n, fstride), np.uint8)
    L_.orc_map_reads_final.restype = C.c_double
    L_.orc_map_reads_final.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int]
    t = L_.orc_map_reads_final(C.c_void_p(oi.h), C.byref(p), rc.ctypes.data, n, 1 if paired else 0, b.ctypes.data,
                               None if bs is None else bs.ctypes.data, ki.ctypes.data, cap, sites.ctypes.data, ns.ctypes.data,
                               log.ctypes.data if want_log else None, logcap, C.addressof(nlog),
                               match.ctypes.data if want_log else None, stride, threads, stats.ctypes.data,
                               fin.ctypes.data, fmatch.ctypes.data, fstride)
    if t < 0:
        raise ValueError("orc_map_reads: bad argument (read longer than the MSA's rows, or mates of different length)")
    if want_log and nlog.value > logcap:
        raise RuntimeError("job log overflow")
    return dict(sites=sites, nsites=ns, final=fin, fmatch=fmatch, log=log[:nlog.value] if want_log else None,
                match=match[:nlog.value] if want_log else None, stats=stats.tolist(), seconds=t)


def final_reads(oi, recs, bases, sites, nsites, paired=False, params=None, match_stride=4200, jobs_per_read=16):
    """The final alignment stage alone over given site lists (orc_final_reads): sites MSITE_DTYPE[n, cap], nsites int32[n].
    Returns a dict like map_reads': sites / nsites after the stage, final, fmatch, log, match."""
    L_ = oi.L
    p = params or map_default_params(getattr(oi, "profile", "bbmap"))
    rc = np.ascontiguousarray(recs, READ_DTYPE)
    n = rc.size
    b = np.ascontiguousarray(bases, np.uint8)
    st = np.ascontiguousarray(sites, MSITE_DTYPE).copy()
    ns = np.ascontiguousarray(nsites, np.int32).copy()
    cap = st.shape[1]
    logcap = n * jobs_per_read + 64
    log = np.zeros(logcap, MJOB_DTYPE)
    match = np.zeros((logcap, match_stride), np.uint8)
    nlog = C.c_int64(0)
    fin = np.zeros(n, FINAL_DTYPE)
    fmatch = np.zeros((n, match_stride), np.uint8)
    L_.orc_final_reads.restype = C.c_int
    L_.orc_final_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rcode = L_.orc_final_reads(C.c_void_p(oi.h), C.byref(p), rc.ctypes.data, n, 1 if paired else 0, b.ctypes.data, cap, st.ctypes.data,
                               ns.ctypes.data, log.ctypes.data, logcap, C.addressof(nlog), match.ctypes.data, match_stride,
                               fin.ctypes.data, fmatch.ctypes.data, match_stride)
    if rcode != 0:
        raise ValueError("orc_final_reads: bad argument")
    if nlog.value > logcap:
        raise RuntimeError("job log overflow")
    return dict(sites=st, nsites=ns, final=fin, fmatch=fmatch, log=log[:nlog.value], match=match[:nlog.value])


def final_branch_counts(oi, reset=True):
    """Which of the final stage's rare paths ran since the last reset (see final_stage.inc: g_branch); single-threaded runs only."""
    out = np.zeros(16, np.int64)
    oi.L.orc_final_branch_counts.argtypes = [C.c_void_p, C.c_int]
    oi.L.orc_final_branch_counts.restype = None
    oi.L.orc_final_branch_counts(out.ctypes.data, 1 if reset else 0)
    names = ("clip_tip_indels", "fix_xy", "to_local", "to_local_clipped", "realign_recursion", "second_realign", "resort_loop", "do_while_repeat",
             "later_site_matched", "duplicate_best_removed")
    return {n: int(out[i]) for i, n in enumerate(names)}
