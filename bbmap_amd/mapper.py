"""Python binding of the device-resident mapper (bbmap_* in include/bbmap_amd.h): probe -> pairing / trimming -> ungapped
scores -> scoreSlow in rounds -> rescue, everything in HBM.  Used by bench.py and the tests; needs torch for device buffers."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import msa as M
from .index import READ_DTYPE

MSITE_DTYPE = np.dtype([("chrom", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("stop", "<i4"), ("hits", "<i4"),
                        ("quickScore", "<i4"), ("score", "<i4"), ("slowScore", "<i4"), ("pairedScore", "<i4"),
                        ("perfect", "<i4"), ("semiperfect", "<i4"), ("rescued", "<i4"), ("ngaps", "<i4"),
                        ("gaps", "<i4", (16,)), ("match_job", "<i4"), ("reserved", "<i4", (2,))])
JOBINFO_DTYPE = np.dtype([("read", "<i4"), ("seq", "<i4"), ("kind", "<i4"), ("site", "<i4")])
# bbmap_final: what BBMap prints for a read (the final alignment stage)
FINAL_DTYPE = np.dtype([("mapped", "<i4"), ("chrom", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("stop", "<i4"), ("mapScore", "<i4"),
                        ("paired", "<i4"), ("ambiguous", "<i4"), ("perfect", "<i4"), ("rescued", "<i4"), ("match_len", "<i4"),
                        ("nsites", "<i4"), ("match_off", "<i8"), ("reserved", "<i4", (2,))])
assert MSITE_DTYPE.itemsize == 128 and FINAL_DTYPE.itemsize == 64
GAPPED_BIT = 1 << 30


class bbmap_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("paired", C.c_int32), ("max_reads", C.c_int32), ("max_read_len", C.c_int32),
                ("max_sites", C.c_int32), ("minRatio", C.c_float)] + [(n, C.c_int32) for n in (
                    "slowAlignPadding", "slowRescuePadding", "extraPadding", "tipSearchDist", "maxPairDist", "averagePairDist",
                    "maxRescueDist", "maxRescueMismatches", "maxTrimSitesToRetain", "trimList", "doRescue", "alignColumns",
                    "clearzone3", "msaMaxColumns", "fastCols", "jobsPerRead", "finalStage")] + [("reserved", C.c_int32 * 4)]


class bbmap_output(C.Structure):
    _fields_ = [("sites", C.c_void_p), ("nsites", C.c_void_p), ("cap", C.c_int32), ("match_stride", C.c_int32),
                ("gmatch_stride", C.c_int32), ("reserved", C.c_int32), ("n_jobs", C.c_int64), ("n_gapped_jobs", C.c_int64),
                ("jobs", C.c_void_p), ("results", C.c_void_p), ("jobinfo", C.c_void_p), ("match", C.c_void_p),
                ("gjobs", C.c_void_p), ("gresults", C.c_void_p), ("gjobinfo", C.c_void_p), ("gmatch", C.c_void_p),
                ("ggaps", C.c_void_p), ("final", C.c_void_p), ("final_match", C.c_void_p), ("final_match_bytes", C.c_int64),
                ("n_final_fills", C.c_int64)]


class bbmap_stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("reads", "reads_overflowed", "reads_without_site", "fills", "gapped_fills", "refills",
                                         "rescue_scans", "rescue_fills", "rounds", "fills_dropped")] + \
               [(n, C.c_float) for n in ("ms_probe", "ms_begin", "ms_score", "ms_slow", "ms_finish", "ms_rescue", "ms_total",
                                         "ms_dp_narrow", "ms_dp_wave", "ms_dp_generic", "ms_dp_gapped", "ms_quick_rescue")] + \
               [("probe_stats", C.c_int64 * 5), ("reads_reprobed", C.c_int64), ("ms_overflow", C.c_float), ("log_growths", C.c_float),
                ("ms_dp_wave_max", C.c_float), ("ms_final", C.c_float), ("final_fills", C.c_int64), ("final_rounds", C.c_int64),
                ("final_local", C.c_int64)]


class bbmap_overflow_output(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("read_ids", C.c_void_p), ("out", bbmap_output)]


NSITES_OVERFLOW, NSITES_MATE_OVERFLOW, NSITES_IN_TIER = -1, -2, -3


def _bind(L):
    L.bbmap_default_config.argtypes = [C.POINTER(bbmap_config)]
    L.bbmap_default_config_profile.argtypes = [C.c_int32, C.POINTER(bbmap_config)]
    L.bbmap_default_config_profile.restype = C.c_int
    L.bbmap_create.argtypes = [C.c_void_p, C.POINTER(bbmap_config), C.POINTER(C.c_void_p)]
    L.bbmap_destroy.argtypes = [C.c_void_p]
    L.bbmap_destroy.restype = None
    L.bbmap_map_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.bbmap_get_output.argtypes = [C.c_void_p, C.POINTER(bbmap_output)]
    L.bbmap_last_stats.argtypes = [C.c_void_p, C.POINTER(bbmap_stats)]
    L.bbmap_get_overflow_output.argtypes = [C.c_void_p, C.POINTER(bbmap_overflow_output)]
    L.bbmap_pack_sites_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.bbmap_pack_sites_device.restype = C.c_int
    L.bbmap_map_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.bbmap_map_batch.restype = C.c_int
    L.bbmap_get_final.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.bbmap_get_final.restype = C.c_int
    L.bbmap_set_average_pair_dist.argtypes = [C.c_void_p, C.c_int32]
    L.bbmap_set_average_pair_dist.restype = C.c_int
    for f in ("bbmap_default_config", "bbmap_create", "bbmap_map_batch_device", "bbmap_get_output", "bbmap_last_stats",
              "bbmap_get_overflow_output"):
        getattr(L, f).restype = C.c_int


def _copy(ptr, nbytes, dev=None):
    """device memory -> numpy bytes (bbmap_copy_to_host)"""
    out = np.empty(max(nbytes, 0), np.uint8)
    if nbytes > 0:
        L = _lib.load()
        L.bbmap_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bbmap_copy_to_host.restype = C.c_int
        _lib.check(L.bbmap_copy_to_host(out.ctypes.data, C.c_void_p(ptr), nbytes), "bbmap_copy_to_host")
    return out


class Mapper:
    """One bbmap_ctx over a DeviceIndex.  Reads of one fixed length, all with the same key offsets / key scores."""

    def __init__(self, di, n_reads, read_len, offsets, key_scores, paired=False, device=0, max_sites=32, **cfg_kw):
        self.L = _lib.load()
        _bind(self.L)
        self.di, self.n, self.read_len, self.paired = di, n_reads, read_len, paired
        self.dev = torch.device("cuda", device)
        di.set_max_read_len(read_len)
        cfg = bbmap_config()
        _lib.check(self.L.bbmap_default_config(C.byref(cfg)), "bbmap_default_config")
        cfg.device, cfg.paired, cfg.max_reads, cfg.max_read_len, cfg.max_sites = device, int(paired), n_reads, read_len, max_sites
        for k, v in cfg_kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.L.bbmap_create(di.h, C.byref(cfg), C.byref(h)), "bbmap_create")
        self.h = h
        self.total_bytes = n_reads * read_len
        self.bases = torch.zeros(2 * self.total_bytes, dtype=torch.uint8, device=self.dev)       # plus strands, then reverse complements
        self.base_scores = torch.zeros(self.total_bytes, dtype=torch.int8, device=self.dev)
        recs = np.zeros(n_reads, READ_DTYPE)
        recs["bases_off"] = np.arange(n_reads, dtype=np.int64) * read_len
        recs["keys_off"] = 0
        recs["len"] = read_len
        recs["nkeys"] = len(offsets)
        self.reads = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(self.dev)
        self.keyinfo = torch.tensor(list(offsets) + list(key_scores), dtype=torch.int32, device=self.dev)

    @classmethod
    def from_records(cls, di, recs, bases, base_scores, keyinfo, paired=False, device=0, max_sites=32, profile=0, **cfg_kw):
        """The general form: reads of any lengths with their own keys, as bbkeys_make_batch (bbmap_amd.keys.make_batch) lays them
        out -- recs (READ_DTYPE), the bases blob, base scores at the same offsets, keyinfo.  profile: 0 = bbmap.sh's classes,
        1 = mapPacBio.sh's (BBIDX_PROFILE_*; must be the index's)."""
        self = cls.__new__(cls)
        self.L = _lib.load()
        _bind(self.L)
        recs = np.ascontiguousarray(recs, READ_DTYPE)
        n = len(recs)
        self.di, self.n, self.paired = di, n, paired
        self.read_len = int(recs["len"].max()) if n else 0
        self.dev = torch.device("cuda", device)
        if profile == 0:
            di.set_max_read_len(max(1, self.read_len))
        cfg = bbmap_config()
        _lib.check(self.L.bbmap_default_config_profile(profile, C.byref(cfg)), "bbmap_default_config_profile")
        cfg.device, cfg.paired, cfg.max_reads, cfg.max_read_len, cfg.max_sites = device, int(paired), n, max(1, self.read_len), max_sites
        for k, v in cfg_kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.L.bbmap_create(di.h, C.byref(cfg), C.byref(h)), "bbmap_create")
        self.h = h
        blob = np.ascontiguousarray(bases, np.uint8)
        self.total_bytes = int(blob.size)
        self.bases = torch.zeros(2 * self.total_bytes, dtype=torch.uint8, device=self.dev)
        self.bases[: self.total_bytes].copy_(torch.from_numpy(blob))
        self.base_scores = torch.from_numpy(np.ascontiguousarray(base_scores, np.int8)).to(self.dev)
        assert self.base_scores.numel() >= self.total_bytes
        self.reads = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(self.dev)
        self.keyinfo = torch.from_numpy(np.ascontiguousarray(keyinfo, np.int32)).to(self.dev)
        return self

    def close(self):
        if getattr(self, "h", None):
            self.L.bbmap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_reads(self, reads_u8):
        """reads_u8: n_reads x read_len bases; paired mode: mates interleaved (read 2p, 2p+1)."""
        assert reads_u8.size == self.total_bytes
        self.bases[: self.total_bytes].copy_(torch.from_numpy(np.ascontiguousarray(reads_u8).reshape(-1)))

    def step(self, bases=None):
        """Maps the resident batch; returns when it is done (bbmap_map_batch_device waits for its stream).  bases: another
        device buffer of 2 * n_reads * read_len bytes holding the batch's plus strands in its first half (a host that streams
        batches uploads the next one while this one is mapped)."""
        stream = torch.cuda.current_stream().cuda_stream
        b = self.bases if bases is None else bases
        assert b.numel() == 2 * self.total_bytes and b.dtype == torch.uint8
        _lib.check(self.L.bbmap_map_batch_device(self.h, C.c_void_p(stream), self.n, self.reads.data_ptr(), b.data_ptr(),
                                                 self.total_bytes, self.base_scores.data_ptr(), self.keyinfo.data_ptr()),
                   "bbmap_map_batch_device")

    def map_batch_host(self, recs, bases, base_scores, keyinfo, sites_cap):
        """bbmap_map_batch: host buffers in, packed site lists out (what the JNI glue calls).  Returns (nsites int32[n],
        offsets int64[n + 1], records MSITE_DTYPE[min(total, sites_cap)], total)."""
        recs = np.ascontiguousarray(recs, READ_DTYPE)
        bases = np.ascontiguousarray(bases, np.uint8)
        base_scores = np.ascontiguousarray(base_scores, np.int8)
        keyinfo = np.ascontiguousarray(keyinfo, np.int32)
        n = len(recs)
        ns = np.zeros(n, np.int32)
        offs = np.zeros(n + 1, np.int64)
        sites = np.zeros(max(1, sites_cap), MSITE_DTYPE)
        total = C.c_int64(0)
        _lib.check(self.L.bbmap_map_batch(self.h, n, recs.ctypes.data, bases.ctypes.data, bases.size, base_scores.ctypes.data,
                                          keyinfo.ctypes.data, keyinfo.size, ns.ctypes.data, offs.ctypes.data, sites.ctypes.data,
                                          sites_cap, C.byref(total)), "bbmap_map_batch")
        return ns, offs, sites[:min(total.value, sites_cap)], total.value

    def final(self, with_match=True):
        """bbmap_get_final: (records FINAL_DTYPE[n], match blob uint8[]) of the last step, overflow tier included; read r's match
        string is blob[match_off : match_off + match_len]."""
        fin = np.zeros(self.n, FINAL_DTYPE)
        nb = C.c_int64(0)
        _lib.check(self.L.bbmap_get_final(self.h, self.n, fin.ctypes.data, None, 0, C.byref(nb)), "bbmap_get_final")
        blob = np.zeros(max(1, nb.value), np.uint8)
        if with_match and nb.value:
            _lib.check(self.L.bbmap_get_final(self.h, self.n, fin.ctypes.data, blob.ctypes.data, blob.size, C.byref(nb)), "bbmap_get_final")
        return fin, blob

    def final_only(self, sites, nsites):
        """bbmap_final_batch_device: the final alignment stage alone over the given site lists (MSITE_DTYPE[n, cap], int32[n]); the
        reverse complements must be in place (a step() wrote them)."""
        st = np.ascontiguousarray(sites, MSITE_DTYPE)
        assert st.shape == (self.n, self.cfg.max_sites)
        d_sites = torch.from_numpy(st.view(np.uint8).reshape(-1).copy()).to(self.dev)
        d_ns = torch.from_numpy(np.ascontiguousarray(nsites, np.int32)).to(self.dev)
        stream = torch.cuda.current_stream().cuda_stream
        self.L.bbmap_final_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        self.L.bbmap_final_batch_device.restype = C.c_int
        _lib.check(self.L.bbmap_final_batch_device(self.h, C.c_void_p(stream), self.n, self.reads.data_ptr(), self.bases.data_ptr(),
                                                   self.total_bytes, d_sites.data_ptr(), d_ns.data_ptr()), "bbmap_final_batch_device")

    def set_average_pair_dist(self, v):
        _lib.check(self.L.bbmap_set_average_pair_dist(self.h, int(v)), "bbmap_set_average_pair_dist")

    def pack_sites(self, counts, offsets, packed):
        """The last step's site lists without their empty slots (bbmap_pack_sites_device), enqueued on the current stream:
        counts int32[n+1], offsets int64[n+1], packed uint8[cap_records * 128] -- device tensors of the caller's."""
        stream = torch.cuda.current_stream().cuda_stream
        assert counts.numel() == self.n + 1 and offsets.numel() == self.n + 1 and packed.numel() % 128 == 0
        _lib.check(self.L.bbmap_pack_sites_device(self.h, C.c_void_p(stream), self.n, counts.data_ptr(), offsets.data_ptr(),
                                                  packed.data_ptr(), packed.numel() // 128), "bbmap_pack_sites_device")

    def output_pointers(self):
        """(bbmap_output of the last step) -- device pointers and counts, for callers that move the logs themselves."""
        o = bbmap_output()
        _lib.check(self.L.bbmap_get_output(self.h, C.byref(o)), "bbmap_get_output")
        return o

    def stats(self):
        st = bbmap_stats()
        _lib.check(self.L.bbmap_last_stats(self.h, C.byref(st)), "bbmap_last_stats")
        d = {n: getattr(st, n) for n, _ in bbmap_stats._fields_ if n not in ("probe_stats",)}
        d["probe_stats"] = list(st.probe_stats)
        return d

    def _fetch_output(self, o, n, with_match, rows=None):
        """Host copies of one bbmap_output over n reads (rows: only the site lists of reads [0, rows))."""
        cap = o.cap
        out = dict(cap=cap)
        ns = n if rows is None else min(rows, n)
        out["sites"] = _copy(o.sites, ns * cap * 128, self.dev).view(MSITE_DTYPE).reshape(ns, cap)
        out["nsites"] = _copy(o.nsites, n * 4, self.dev).view(np.int32)
        nj, ng = int(o.n_jobs), int(o.n_gapped_jobs)
        out["jobs"] = _copy(o.jobs, nj * 40, self.dev).view(M.JOB_DTYPE)
        out["results"] = _copy(o.results, nj * 80, self.dev).view(M.RESULT_DTYPE)
        out["jobinfo"] = _copy(o.jobinfo, nj * 16, self.dev).view(JOBINFO_DTYPE)
        out["gjobs"] = _copy(o.gjobs, ng * 40, self.dev).view(M.JOB_DTYPE)
        out["gresults"] = _copy(o.gresults, ng * 80, self.dev).view(M.RESULT_DTYPE)
        out["gjobinfo"] = _copy(o.gjobinfo, ng * 16, self.dev).view(JOBINFO_DTYPE)
        out["ggaps"] = _copy(o.ggaps, ng * 68, self.dev).view(M.GAPS_DTYPE)
        out["match_stride"], out["gmatch_stride"] = o.match_stride, o.gmatch_stride
        if with_match:
            out["match"] = _copy(o.match, nj * o.match_stride, self.dev).reshape(nj, o.match_stride)
            out["gmatch"] = _copy(o.gmatch, ng * o.gmatch_stride, self.dev).reshape(ng, o.gmatch_stride)
        return out

    def fetch(self, with_match=True, rows=None):
        """Host copies of a step's results: sites (n x cap, MSITE_DTYPE), nsites, and the two fill logs; out["overflow"] holds
        the same for the reads the overflow tier mapped (nsites == NSITES_IN_TIER in the main list), plus their read_ids."""
        o = bbmap_output()
        _lib.check(self.L.bbmap_get_output(self.h, C.byref(o)), "bbmap_get_output")
        out = self._fetch_output(o, self.n, with_match, rows)
        ov = bbmap_overflow_output()
        _lib.check(self.L.bbmap_get_overflow_output(self.h, C.byref(ov)), "bbmap_get_overflow_output")
        if ov.n_reads > 0:
            t = self._fetch_output(ov.out, int(ov.n_reads), with_match)
            t["read_ids"] = _copy(ov.read_ids, int(ov.n_reads) * 4, self.dev).view(np.int32)
            out["overflow"] = t
        if self.cfg.finalStage and rows is None:
            out["final"], out["final_match"] = self.final(with_match)
        return out
