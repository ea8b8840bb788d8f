"""Binding of bbkeys_*: the probe's per-read inputs (key offsets, key scores, base scores), made on the host exactly where the
reference makes them -- AbstractMapThread.quickMap up to its findAdvanced call (current/align2/AbstractMapThread.java:642-728)."""
import ctypes as C

import numpy as np

from . import _lib
from .index import READ_DTYPE

PROFILE_BBMAP, PROFILE_PACBIO = 0, 1


class bbkeys_config(C.Structure):
    _fields_ = [("k", C.c_int32), ("keyDensity", C.c_float), ("maxKeyDensity", C.c_float), ("minKeyDensity", C.c_float),
                ("maxDesiredKeys", C.c_int32), ("minApproxHitsToKeep", C.c_int32), ("semiperfectMode", C.c_int32), ("reserved", C.c_int32)]


def default_config(profile=PROFILE_BBMAP, **kw):
    L = _lib.load()
    cfg = bbkeys_config()
    L.bbkeys_default_config.argtypes = [C.c_int32, C.POINTER(bbkeys_config)]
    L.bbkeys_default_config.restype = C.c_int
    _lib.check(L.bbkeys_default_config(profile, C.byref(cfg)), "bbkeys_default_config")
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def make_keys(bases, quality=None, cfg=None):
    """One read: returns (offsets, keyScores, baseScores); offsets is empty when quickMap would not probe the read.
    quality: numeric phred values (bytes / uint8 array) or None."""
    L = _lib.load()
    cfg = cfg or default_config()
    b = np.frombuffer(bytes(bases), np.uint8).copy() if not isinstance(bases, np.ndarray) else np.ascontiguousarray(bases, np.uint8)
    n = len(b)
    q = None if quality is None else np.ascontiguousarray(np.frombuffer(bytes(quality), np.uint8) if not isinstance(quality, np.ndarray) else quality, np.uint8)
    cap = max(1, n)
    offs, ks, bs = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(max(1, n), np.int8)
    L.bbkeys_make.argtypes = [C.POINTER(bbkeys_config), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.bbkeys_make.restype = C.c_int
    m = L.bbkeys_make(C.byref(cfg), b.ctypes.data, None if q is None else q.ctypes.data, n, offs.ctypes.data, ks.ctypes.data, cap, bs.ctypes.data)
    if m < 0:
        _lib.check(m, "bbkeys_make")
    return offs[:m].tolist(), ks[:m].tolist(), bs[:n]


def make_batch(reads, qualities=None, cfg=None):
    """reads: list of uint8 arrays / bytes (any lengths); qualities: None or a list of the same shapes (numeric phred).
    Returns (recs [READ_DTYPE], bases blob, baseScores blob, keyinfo) laid out as bbmap_map_batch_device / bbidx_find_batch take them."""
    L = _lib.load()
    cfg = cfg or default_config()
    arrs = [np.frombuffer(bytes(r), np.uint8) if not isinstance(r, np.ndarray) else np.ascontiguousarray(r, np.uint8) for r in reads]
    lens = np.array([len(a) for a in arrs], np.int32)
    offs = np.zeros(len(arrs), np.int64)
    if len(arrs) > 1:
        offs[1:] = np.cumsum(lens[:-1].astype(np.int64))
    blob = np.concatenate(arrs) if arrs else np.zeros(1, np.uint8)
    qblob = None
    if qualities is not None:
        qblob = np.concatenate([np.frombuffer(bytes(q), np.uint8) if not isinstance(q, np.ndarray) else np.ascontiguousarray(q, np.uint8) for q in qualities])
        assert len(qblob) == len(blob)
    recs = np.zeros(len(arrs), READ_DTYPE)
    cap = int(2 * lens.astype(np.int64).sum()) + 2
    keyinfo = np.zeros(cap, np.int32)
    bs = np.zeros(max(1, len(blob)), np.int8)
    used = C.c_int64(0)
    L.bbkeys_make_batch.argtypes = [C.POINTER(bbkeys_config), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]
    L.bbkeys_make_batch.restype = C.c_int
    _lib.check(L.bbkeys_make_batch(C.byref(cfg), len(arrs), offs.ctypes.data, lens.ctypes.data, blob.ctypes.data,
                                   None if qblob is None else qblob.ctypes.data, recs.ctypes.data, keyinfo.ctypes.data, cap, bs.ctypes.data,
                                   C.byref(used)), "bbkeys_make_batch")
    return recs, blob, bs, keyinfo[:max(1, used.value)].copy()
