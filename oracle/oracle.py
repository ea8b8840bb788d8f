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


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_msa_new.restype = C.c_void_p
        L.orc_msa_new.argtypes = [C.c_int, C.c_int]
        L.orc_msa_free.argtypes = [C.c_void_p]
        for name in ("orc_points_ins_array", "orc_pointsoff_ins_array", "orc_points_ins_array_c",
                     "orc_pointsoff_ins_array_c", "orc_points_sub_array", "orc_pointsoff_sub_array"):
            getattr(L, name).restype = c_i32p
        L.orc_base_to_number.restype = c_i8p
        L.orc_calc_del_score_offset.restype = C.c_int32
        L.orc_calc_ins_score_offset.restype = C.c_int32
        _LIB = L
    return _LIB


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

    def __init__(self, maxRows=601, maxColumns=3000, bandwidth=0, bandwidthRatio=0.0):
        self.L = lib()
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
