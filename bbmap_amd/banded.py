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

    # BandedAligner.java:39-48, batched over pairs
    def alignQuadruple(self, pairs, maxEdits, exact):
        fw = self.align_batch([(FORWARD, q, r, 0, 0, maxEdits, exact) for q, r in pairs])
        rv = self.align_batch([(REVERSE, q, r, len(q) - 1, len(r) - 1, maxEdits, exact) for q, r in pairs])
        out = [0] * len(pairs)
        todo, me2s = [], []
        for k, (q, r) in enumerate(pairs):
            a, b = int(fw[k]["edits"]), int(rv[k]["edits"])
            me2 = min(maxEdits, max(a, b))
            out[k] = max(a, b)
            if me2 != 0:
                todo.append(k)
                me2s.append(me2)
        if todo:
            c = self.align_batch([(FORWARD_RC, pairs[k][0], pairs[k][1], len(pairs[k][0]) - 1, 0, m, exact)
                                  for k, m in zip(todo, me2s)])
            d = self.align_batch([(REVERSE_RC, pairs[k][0], pairs[k][1], 0, len(pairs[k][1]) - 1, m, exact)
                                  for k, m in zip(todo, me2s)])
            for n, k in enumerate(todo):
                out[k] = min(out[k], max(int(c[n]["edits"]), int(d[n]["edits"])))
        return out
