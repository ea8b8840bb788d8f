"""Python binding of the batched MultiStateAligner11ts C ABI (tests / bench plumbing).

Mirrors the call shapes of align2.MSA (current/align2/MSA.java:70-144) but batched: every method
takes lists of problems and issues ONE bbmsa_align_batch call.  All compute happens in the HIP
library; nothing here touches the oracle.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import bbmsa_config, bbmsa_job, bbmsa_result

FILL_LIMITED_RAW = 0
FILL_UNLIMITED_RAW = 1
FILL_LIMITED = 2
CLAMP_WINDOW = 1 << 3
DO_SCORE = 1 << 4
DO_TRACEBACK = 1 << 5
NO_ITERATIONS = 1 << 6
FILL_AND_SCORE_LIMITED = FILL_LIMITED | CLAMP_WINDOW | DO_SCORE

ST_OK, ST_NULL, ST_BAD_SHAPE = 0, 1, 2

JOB_DTYPE = np.dtype([("read_off", "<i8"), ("ref_off", "<i8"), ("read_len", "<i4"), ("ref_len", "<i4"),
                      ("refStartLoc", "<i4"), ("refEndLoc", "<i4"), ("minScore", "<i4"), ("flags", "<i4")])
RESULT_DTYPE = np.dtype([("result", "<i4", (5,)), ("status", "<i4"), ("iterations", "<i8"),
                         ("score", "<i4", (8,)), ("score_len", "<i4"), ("match_len", "<i4"),
                         ("fill_kind", "<i4"), ("columns", "<i4")])
GAPS_DTYPE = np.dtype([("ngaps", "<i4"), ("gaps", "<i4", (16,))])
assert JOB_DTYPE.itemsize == C.sizeof(bbmsa_job) and RESULT_DTYPE.itemsize == C.sizeof(bbmsa_result)
assert GAPS_DTYPE.itemsize == 68


class bbmsa_ticket(C.Structure):
    _fields_ = [("batch", C.c_int32), ("slot", C.c_int32), ("gen", C.c_int64), ("rows", C.c_int32), ("columns", C.c_int32)]


SCHEME_11TS, SCHEME_9PACBIO = 0, 1      # BBMSA_SCHEME_* (include/bbmap_amd.h)
LEGACY_ONLY = 0x100                     # BBMSA_LEGACY_ONLY: a context for the per-call fills (fill_packed) only


class MSAContext:
    """Owns a bbmsa_ctx (one per device and per (maxRows, maxColumns, band) setting)."""

    def __init__(self, maxRows=601, maxColumns=3000, bandwidth=0, bandwidthRatio=0.0, device=0,
                 lanes_per_job=0, fast_cols=0, scheme=SCHEME_11TS, legacy=False):
        self.L = _lib.load()
        cfg = bbmsa_config()
        cfg.device, cfg.maxRows, cfg.maxColumns = device, maxRows, maxColumns
        cfg.bandwidth, cfg.bandwidthRatio = bandwidth, bandwidthRatio
        cfg.reserved[0], cfg.reserved[1], cfg.reserved[2] = lanes_per_job, fast_cols, scheme | (LEGACY_ONLY if legacy else 0)
        h = C.c_void_p()
        _lib.check(self.L.bbmsa_create(C.byref(cfg), C.byref(h)), "bbmsa_create")
        self.h = h
        self.maxRows, self.maxColumns = maxRows, maxColumns

    def close(self):
        if getattr(self, "h", None):
            self.L.bbmsa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host buffers -------------------------------------------------------------------
    def align_batch(self, jobs, reads, refs, match_stride=0):
        """jobs: structured array (JOB_DTYPE); reads/refs: uint8 arrays.  Returns (results, match)."""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        refs = np.ascontiguousarray(refs, dtype=np.uint8)
        n = len(jobs)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        match = np.zeros((n, match_stride), np.uint8) if match_stride > 0 else None
        rc = self.L.bbmsa_align_batch(self.h, n, jobs.ctypes.data, reads.ctypes.data, reads.size,
                                      refs.ctypes.data, refs.size, res.ctypes.data,
                                      match.ctypes.data if match is not None else None, match_stride)
        _lib.check(rc, "bbmsa_align_batch")
        return res, match

    def align_gapped_batch(self, jobs, gaps, reads, refs, match_stride=0):
        """Like align_batch, with one GAPS_DTYPE record per job (ngaps == 0: an ordinary job)."""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        gaps = np.ascontiguousarray(gaps, dtype=GAPS_DTYPE)
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        refs = np.ascontiguousarray(refs, dtype=np.uint8)
        n = len(jobs)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        match = np.zeros((n, match_stride), np.uint8) if match_stride > 0 else None
        rc = self.L.bbmsa_align_gapped_batch(self.h, n, jobs.ctypes.data, gaps.ctypes.data, reads.ctypes.data, reads.size,
                                             refs.ctypes.data, refs.size, res.ctypes.data,
                                             match.ctypes.data if match is not None else None, match_stride)
        _lib.check(rc, "bbmsa_align_gapped_batch")
        return res, match

    def fill_packed(self, read, ref, a, b, minScore, limited, packed, limits=False):
        """The legacy per-call fill (context made with legacy=True): writes the planes into `packed` (int32 array of
        3*(maxRows+1)*(maxColumns+1)).  Returns (result[5], iterations), and with limits=True also (vertLimit, horizLimit).
        Thread-safe: concurrent calls on one context are combined into one launch."""
        r = np.frombuffer(bytes(read), np.uint8)
        f = np.frombuffer(bytes(ref), np.uint8)
        res = np.zeros(5, np.int32)
        it = C.c_int64(0)
        t = bbmsa_ticket()
        rc = self.L.bbmsa_fill_submit(self.h, r.ctypes.data, len(r), f.ctypes.data, len(f), a, b, minScore,
                                      FILL_LIMITED_RAW if limited else FILL_UNLIMITED_RAW, res.ctypes.data, C.byref(it), C.byref(t))
        _lib.check(rc, "bbmsa_fill_submit")
        vl = np.zeros(len(r) + 1, np.int32)
        hl = np.zeros(b - a + 2, np.int32)
        rc = self.L.bbmsa_fill_collect(self.h, C.byref(t), packed.ctypes.data, vl.ctypes.data, hl.ctypes.data)
        _lib.check(rc, "bbmsa_fill_collect")
        if limits:
            return res.tolist(), it.value, vl, hl
        return res.tolist(), it.value

    def legacy_stats(self):
        st = (C.c_int64 * 6)()
        _lib.check(self.L.bbmsa_legacy_stats(self.h, st), "bbmsa_legacy_stats")
        return dict(calls=st[0], launches=st[1], handed_on=st[2], wave_ms=st[3] / 1e6, handed_ms=st[4] / 1e6, wait_collect_ms=st[5] / 1e6)

    # -- device buffers (torch tensors or raw pointers) --------------------------------------
    def align_batch_device(self, n_jobs, jobs_ptr, reads_ptr, refs_ptr, results_ptr, match_ptr=0,
                           match_stride=0, stream=0):
        rc = self.L.bbmsa_align_batch_device(self.h, C.c_void_p(stream), n_jobs, C.c_void_p(jobs_ptr),
                                             C.c_void_p(reads_ptr), C.c_void_p(refs_ptr),
                                             C.c_void_p(results_ptr),
                                             C.c_void_p(match_ptr) if match_ptr else None, match_stride)
        _lib.check(rc, "bbmsa_align_batch_device")

    def last_counts(self):
        """{narrow: finished by the one-job-per-lane kernel, narrow_left: its candidates handed on, wave: jobs of the
        wavefront kernel, generic: jobs of the generic kernel} for the last launch sequence."""
        c = (C.c_int64 * 4)()
        _lib.check(self.L.bbmsa_last_counts(self.h, c), "bbmsa_last_counts")
        return {"narrow": c[0], "narrow_left": c[1], "wave": c[2], "generic": c[3]}

    def last_kernel_ms3(self):
        """(narrow-window kernel, wavefront kernel, generic kernel) milliseconds of the last launch sequence."""
        m = (C.c_float * 3)()
        _lib.check(self.L.bbmsa_last_kernel_ms3(self.h, m), "bbmsa_last_kernel_ms3")
        return m[0], m[1], m[2]

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        _lib.check(self.L.bbmsa_last_kernel_ms(self.h, C.byref(a), C.byref(b)), "bbmsa_last_kernel_ms")
        return a.value, b.value


def pack_problems(problems, flags):
    """problems: iterable of (read_bytes, ref_bytes, refStartLoc, refEndLoc, minScore).
    Packs them into (jobs, reads_blob, refs_blob); identical ref objects are stored once."""
    reads, refs, jobs = bytearray(), bytearray(), []
    ref_cache = {}
    for k, (read, ref, a, b, ms) in enumerate(problems):
        ro = len(reads)
        reads += bytes(read)
        key = id(ref)
        if key not in ref_cache:
            ref_cache[key] = len(refs)
            refs += bytes(ref)
        fo = ref_cache[key]
        fl = flags[k] if isinstance(flags, (list, tuple, np.ndarray)) else flags
        jobs.append((ro, fo, len(read), len(ref), a, b, ms, fl))
    jobs = np.array(jobs, dtype=JOB_DTYPE) if jobs else np.zeros(0, JOB_DTYPE)
    return jobs, np.frombuffer(bytes(reads) or b"\0", np.uint8), np.frombuffer(bytes(refs) or b"\0", np.uint8)


class MultiStateAligner11ts:
    """Batched mirror of align2.MultiStateAligner11tsJNI's public methods."""

    def __init__(self, maxRows=601, maxColumns=3000, bandwidth=0, bandwidthRatio=0.0, device=0, **kw):
        self.ctx = MSAContext(maxRows, maxColumns, bandwidth, bandwidthRatio, device, **kw)
        self.maxRows, self.maxColumns = maxRows, maxColumns
        self.iterationsLimited = 0
        self.iterationsUnlimited = 0

    def _run(self, problems, flags, want_match=False):
        problems = list(problems)
        jobs, reads, refs = pack_problems(problems, flags)
        stride = 0
        if want_match and problems:
            # rows + columns symbols at most, and every '-' gap symbol expands to 128 'D'
            stride = max(len(p[0]) + (p[3] - p[2] + 1) + 8 +
                         127 * bytes(p[1][max(0, p[2]):max(0, p[3] + 1)]).count(b"-") for p in problems)
            stride = (stride + 15) & ~15
        res, match = self.ctx.align_batch(jobs, reads, refs, stride)
        for r in res:
            if r["status"] == ST_BAD_SHAPE:
                raise ValueError("alignment exceeds maxRows/maxColumns of this aligner")
            if r["fill_kind"] == 0:
                self.iterationsLimited += int(r["iterations"])
            else:
                self.iterationsUnlimited += int(r["iterations"])
        return res, match

    # MultiStateAligner11tsJNI.java:116-164 (gaps==null)
    def fillLimited(self, problems):
        res, _ = self._run(problems, FILL_LIMITED)
        return [None if r["status"] == ST_NULL else r["result"][:4].tolist() for r in res]

    # :166-192
    def fillUnlimited(self, problems):
        res, _ = self._run([(p[0], p[1], p[2], p[3], 0) for p in problems], FILL_UNLIMITED_RAW)
        return [r["result"][:4].tolist() for r in res]

    # MSA.java:103-134 (gaps==null); returns score vectors (or None)
    def fillAndScoreLimited(self, problems):
        res, _ = self._run(problems, FILL_AND_SCORE_LIMITED)
        return [None if r["score_len"] == 0 else r["score"][:r["score_len"]].tolist() for r in res]

    # fillLimited + score + traceback in one launch
    def align(self, problems, flags=FILL_AND_SCORE_LIMITED | DO_TRACEBACK):
        res, match = self._run(problems, flags, want_match=True)
        out = []
        for k, r in enumerate(res):
            ms = None
            if r["match_len"] > 0:
                ms = match[k, :r["match_len"]].tobytes()
            out.append({"result": r["result"].tolist(), "status": int(r["status"]),
                        "iterations": int(r["iterations"]),
                        "score": None if r["score_len"] == 0 else r["score"][:r["score_len"]].tolist(),
                        "match": ms, "fill_kind": int(r["fill_kind"]), "columns": int(r["columns"])})
        return out

    # MSA.fillAndScoreLimited(read, ref, start, stop, minScore, gaps) + traceback(..., gapped) in one launch;
    # problems: (read, ref, refStartLoc, refEndLoc, minScore, gaps or None)
    def alignGapped(self, problems, traceback=True):
        problems = list(problems)
        flags = FILL_AND_SCORE_LIMITED | (DO_TRACEBACK if traceback else 0)
        jobs, reads, refs = pack_problems([p[:5] for p in problems], flags)
        gaps = np.zeros(len(problems), GAPS_DTYPE)
        for k, p in enumerate(problems):
            if p[5] is not None:
                gaps[k]["ngaps"] = len(p[5])
                gaps[k]["gaps"][:len(p[5])] = p[5]
        stride = ((self.maxRows + self.maxColumns + 2 + 128 * 24 + 15) // 16) * 16
        res, match = self.ctx.align_gapped_batch(jobs, gaps, reads, refs, stride)
        out = []
        for k, r in enumerate(res):
            out.append({"status": int(r["status"]), "result": r["result"].tolist(),
                        "score": None if r["score_len"] == 0 else r["score"][:r["score_len"]].tolist(),
                        "match": match[k, :r["match_len"]].tobytes() if r["match_len"] > 0 else None})
        return out


class MultiStateAligner9PacBio(MultiStateAligner11ts):
    """align2.MultiStateAligner9PacBio (current/align2/MultiStateAligner9PacBio.java): the same batched interface with the
    PacBio parameter set (9 time bits, its own point values and barriers); reads up to 6019 bases."""

    def __init__(self, maxRows=6019, maxColumns=7600, bandwidth=0, bandwidthRatio=0.0, device=0, **kw):
        super().__init__(maxRows, maxColumns, bandwidth, bandwidthRatio, device, scheme=SCHEME_9PACBIO, **kw)
