"""Python binding of the batched BandedAligner C ABI (tests / bench plumbing).

Mirrors align2.BandedAligner (current/align2/BandedAligner.java): the four directional aligners are
batched natively; alignQuadruple / alignQuadrupleProgressive / alignDouble are the reference's
host-side orchestration (BandedAligner.java:24-55) over them.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import bbband_config

FORWARD, FORWARD_RC, REVERSE, REVERSE_RC = 0, 1, 2, 3
EXACT = 1 << 2
SEMANTICS_JNI_C, SEMANTICS_JAVA = 0, 1

JOB_DTYPE = np.dtype([("query_off", "<i8"), ("ref_off", "<i8"), ("query_len", "<i4"), ("ref_len", "<i4"),
                      ("qstart", "<i4"), ("rstart", "<i4"), ("maxEdits", "<i4"), ("flags", "<i4")])
PAIR_DTYPE = np.dtype([("query_off", "<i8"), ("ref_off", "<i8"), ("query_len", "<i4"), ("ref_len", "<i4")])
RESULT_DTYPE = np.dtype([("edits", "<i4"), ("lastQueryLoc", "<i4"), ("lastRefLoc", "<i4"), ("lastRow", "<i4"),
                         ("lastEdits", "<i4"), ("lastOffset", "<i4"), ("status", "<i4"), ("reserved", "<i4")])


class BandedAligner:
    def __init__(self, width, semantics=SEMANTICS_JAVA, device=0):
        self.L = _lib.load()
        cfg = bbband_config()
        cfg.device, cfg.width, cfg.semantics, cfg.reserved = device, width, semantics, 0
        h = C.c_void_p()
        _lib.check(self.L.bbband_create(C.byref(cfg), C.byref(h)), "bbband_create")
        self.h = h
        self.maxWidth = max(width, 3) | 1

    def close(self):
        if getattr(self, "h", None):
            self.L.bbband_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align_batch(self, problems):
        """problems: list of (direction, query, ref, qstart, rstart, maxEdits, exact).  Returns RESULT_DTYPE array."""
        blob = bytearray()
        cache = {}

        def put(b):
            k = id(b)
            if k not in cache:
                cache[k] = len(blob)
                blob.extend(bytes(b))
            return cache[k]
        jobs = np.zeros(len(problems), JOB_DTYPE)
        for n, (d, q, r, qs, rs, me, ex) in enumerate(problems):
            jobs[n] = (put(q), put(r), len(q), len(r), qs, rs, me, d | (EXACT if ex else 0))
        seqs = np.frombuffer(bytes(blob) or b"\0", np.uint8)
        res = np.zeros(len(problems), RESULT_DTYPE)
        rc = self.L.bbband_align_batch(self.h, len(jobs), jobs.ctypes.data, seqs.ctypes.data, seqs.size, res.ctypes.data)
        _lib.check(rc, "bbband_align_batch")
        return res

    # BandedAligner.java:24-55: the orchestration runs in the library (bbband_align_*_batch), batched over pairs
    def _pairs(self, pairs):
        blob = bytearray()
        recs = np.zeros(len(pairs), PAIR_DTYPE)
        for n, (q, r) in enumerate(pairs):
            recs[n] = (len(blob), len(blob) + len(q), len(q), len(r))
            blob += bytes(q) + bytes(r)
        return recs, np.frombuffer(bytes(blob) or b"\0", np.uint8)

    def alignQuadruple(self, pairs, maxEdits, exact):
        recs, seqs = self._pairs(pairs)
        out = np.zeros(len(pairs), np.int32)
        self.L.bbband_align_quadruple_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        _lib.check(self.L.bbband_align_quadruple_batch(self.h, len(pairs), recs.ctypes.data, seqs.ctypes.data, seqs.size, maxEdits,
                                                       1 if exact else 0, out.ctypes.data), "bbband_align_quadruple_batch")
        return out.tolist()

    def alignQuadrupleProgressive(self, pairs, minEdits, maxEdits, exact):
        recs, seqs = self._pairs(pairs)
        out = np.zeros(len(pairs), np.int32)
        self.L.bbband_align_quadruple_progressive_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                                    C.c_int32, C.c_int32, C.c_void_p]
        _lib.check(self.L.bbband_align_quadruple_progressive_batch(self.h, len(pairs), recs.ctypes.data, seqs.ctypes.data, seqs.size,
                                                                   minEdits, maxEdits, 1 if exact else 0, out.ctypes.data),
                   "bbband_align_quadruple_progressive_batch")
        return out.tolist()

    def alignDouble(self, pairs, maxEdits, exact):
        recs, seqs = self._pairs(pairs)
        out = np.zeros(len(pairs), np.int32)
        self.L.bbband_align_double_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        _lib.check(self.L.bbband_align_double_batch(self.h, len(pairs), recs.ctypes.data, seqs.ctypes.data, seqs.size, maxEdits,
                                                    1 if exact else 0, out.ctypes.data), "bbband_align_double_batch")
        return out.tolist()
