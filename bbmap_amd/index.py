"""Index construction for the probe (host plumbing, numpy) + Python binding of the bbidx_* C ABI.

The reference builds its k-mer index on the host (align2.IndexMaker4, current/align2/IndexMaker4.java:303-421:
count -> prefix sum -> fill) and then derives COUNTS / the length histogram / MAX_USABLE_LENGTH in
BBIndex.analyzeIndex (current/align2/BBIndex.java:101-191).  No JVM exists in this pipeline, so the
harness builds the same arrays here, vectorised; the arrays are then uploaded once and stay in HBM.
A GPU build (count -> scan -> scatter) is SURVEY section 8(f) row N1 and comes later.
"""
import ctypes as C

import numpy as np

from . import _lib

SMALL_GENOME_LIST = 20


def rc_keys(keys, k):
    """AminoAcid.reverseComplementBinaryFast for an int array."""
    out = np.zeros_like(keys)
    x = keys.copy()
    for _ in range(k):
        out = (out << 2) | ((~x) & 3)
        x >>= 2
    return out


def _rc_keys_torch(keys, k):
    import torch
    out = torch.zeros_like(keys)
    x = keys.clone()
    for _ in range(k):
        out = (out << 2) | ((~x) & 3)
        x >>= 2
    return out


class HostIndex:
    """The arrays BBIndex reads: per-block CSR (starts, sites), COUNTS, the length histogram, the tunables.

    backend="numpy" builds on the host; backend="torch" runs the same count -> scan -> scatter on the GPU
    (every step is a bandwidth-bound pass over the genome or the key space) and copies the arrays back."""

    def __init__(self, chroms, k=13, chromBits=None, backend="auto", device=0):
        self.k = k
        self.chroms = [np.ascontiguousarray(np.frombuffer(bytes(c), np.uint8)) if not isinstance(c, np.ndarray)
                       else np.ascontiguousarray(c, dtype=np.uint8) for c in chroms]
        n = len(self.chroms)
        maxlen = max(len(c) for c in self.chroms)
        if chromBits is None:                     # RefToIndex.AUTO_CHROMBITS, BBMap.java:317-321
            chromBits = min(16, (32 - int(maxlen).bit_length()) - 1)
        self.chromBits = chromBits
        self.nchroms = n
        self.nblocks = (n >> chromBits) + 1
        if backend == "auto":
            backend = "numpy"
            if k >= 12:
                try:
                    import torch
                    if torch.cuda.is_available():
                        backend = "torch"
                except Exception:
                    pass
        if backend == "torch":
            self._build_torch(device)
        else:
            self._build_numpy()
        self.length_histogram = self._length_histogram(self.counts)
        self._set_params()

    # -- shared pieces -----------------------------------------------------------------------------
    def _block_chroms(self, b):
        cpb = 1 << self.chromBits
        return range(max(1, b * cpb), min(self.nchroms, b * cpb + cpb - 1) + 1)

    def _build_numpy(self):
        k, n, chromBits = self.k, self.nchroms, self.chromBits
        keyspace = 1 << (2 * k)
        cpb = 1 << chromBits
        shift = 31 - chromBits
        lut = np.full(256, -1, np.int64)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = i
        banmask = (1 << (2 * k - 4)) - 1
        self.starts, self.sites = [], []
        counts = np.zeros(keyspace, np.int32)
        clump_keys = []
        self.defined_bases = 0
        for b in range(self.nblocks):
            keys_all, sites_all = [], []
            for chrom in self._block_chroms(b):
                arr = self.chroms[chrom - 1]
                code = lut[arr]
                self.defined_bases += int((code >= 0).sum())
                L = len(arr)
                npos = L - k          # IndexMaker4: a < max, max = maxIndex - KEYLEN + 1 = L - k
                if npos <= 0:
                    continue
                bad = (code < 0).astype(np.int32)
                csum = np.concatenate(([0], np.cumsum(bad)))
                valid = (csum[k:k + npos] - csum[:npos]) == 0
                key = np.zeros(npos, np.int64)
                c2 = np.where(code < 0, 0, code)
                for j in range(k):
                    key = (key << 2) | c2[j:j + npos]
                valid &= (key >> 4) != (key & banmask)          # periodic k-mers are banned (:327-339)
                pos = np.nonzero(valid)[0]
                keys_all.append(key[pos])
                sites_all.append((((chrom & (cpb - 1)) << shift) | pos).astype(np.int64))
            keys_cat = np.concatenate(keys_all) if keys_all else np.zeros(0, np.int64)
            sites_cat = np.concatenate(sites_all) if sites_all else np.zeros(0, np.int64)
            order = np.argsort(keys_cat, kind="stable")      # genome order inside each list, like the reference
            keys_sorted = keys_cat[order]
            sites = sites_cat[order].astype(np.int32)
            uniq, cnt = np.unique(keys_sorted, return_counts=True)
            c32 = np.zeros(keyspace + 1, np.int32)
            c32[uniq + 1] = cnt
            counts[uniq] += cnt.astype(np.int32)
            starts = np.cumsum(c32, dtype=np.int32)
            if len(sites) > 1:                                 # clumpy keys (BBIndex.java:125-143)
                dif = sites[1:].astype(np.int64) - sites[:-1].astype(np.int64)
                hit = (keys_sorted[1:] == keys_sorted[:-1]) & (dif > 0) & (dif <= 5)
                if hit.any():
                    kk = keys_sorted[1:][hit]
                    clump_keys.append(np.minimum(kk, rc_keys(kk, k)))
            self.starts.append(starts)
            self.sites.append(np.ascontiguousarray(sites))
        # COUNTS[key] = len(key) + len(rc(key)) (BBIndex.java:147-153): only keys that occur need touching
        nz = np.nonzero(counts)[0].astype(np.int64)
        rk = rc_keys(nz, k)
        own, other = counts[nz].astype(np.int64), counts[rk].astype(np.int64)
        comb = np.where(nz != rk, np.minimum(own + other, 2**31 - 1), own).astype(np.int32)
        counts[nz] = comb
        counts[rk] = comb
        if clump_keys:
            ck, cc = np.unique(np.concatenate(clump_keys), return_counts=True)
            ln = counts[ck].astype(np.int64)
            zero = ck[(ln > 2000) & (cc.astype(np.float32) > np.float32(0.75) * ln.astype(np.float32))]
            counts[zero] = 0
            counts[rc_keys(zero, k)] = 0
        self.counts = counts

    def _build_torch(self, device):
        import torch
        dev = torch.device("cuda", device)
        k, chromBits = self.k, self.chromBits
        keyspace = 1 << (2 * k)
        cpb = 1 << chromBits
        shift = 31 - chromBits
        lut = torch.full((256,), -1, dtype=torch.int64, device=dev)
        for i, ch in enumerate(b"ACGT"):
            lut[ch] = i
        banmask = (1 << (2 * k - 4)) - 1
        self.starts, self.sites = [], []
        counts = torch.zeros(keyspace, dtype=torch.int64, device=dev)
        clump = torch.zeros(0, dtype=torch.int64, device=dev)
        self.defined_bases = 0
        for b in range(self.nblocks):
            keys_all, sites_all = [], []
            for chrom in self._block_chroms(b):
                arr = torch.from_numpy(self.chroms[chrom - 1]).to(dev)
                code = lut[arr.long()]
                self.defined_bases += int((code >= 0).sum().item())
                L = arr.numel()
                npos = L - k
                if npos <= 0:
                    continue
                csum = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum((code < 0).long(), 0)))
                valid = (csum[k:k + npos] - csum[:npos]) == 0
                c2 = torch.where(code < 0, torch.zeros_like(code), code)
                key = torch.zeros(npos, dtype=torch.int64, device=dev)
                for j in range(k):
                    key = (key << 2) | c2[j:j + npos]
                valid &= (key >> 4) != (key & banmask)
                pos = torch.nonzero(valid).flatten()
                keys_all.append(key[pos])
                sites_all.append(((chrom & (cpb - 1)) << shift) | pos)
            keys_cat = torch.cat(keys_all) if keys_all else torch.zeros(0, dtype=torch.int64, device=dev)
            sites_cat = torch.cat(sites_all) if sites_all else torch.zeros(0, dtype=torch.int64, device=dev)
            keys_sorted, order = torch.sort(keys_cat, stable=True)
            sites = sites_cat[order].to(torch.int32)
            cnt = torch.bincount(keys_sorted, minlength=keyspace)
            starts = torch.zeros(keyspace + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt, 0, out=starts[1:])
            counts += cnt
            if sites.numel() > 1:
                dif = sites[1:].long() - sites[:-1].long()
                hit = (keys_sorted[1:] == keys_sorted[:-1]) & (dif > 0) & (dif <= 5)
                kk = keys_sorted[1:][hit]
                clump = torch.cat((clump, torch.minimum(kk, _rc_keys_torch(kk, k))))
            self.starts.append(starts.to(torch.int32).cpu().numpy())
            self.sites.append(np.ascontiguousarray(sites.cpu().numpy()))
            del cnt, starts, keys_sorted, order
        nz = torch.nonzero(counts).flatten()
        rk = _rc_keys_torch(nz, k)
        own, other = counts[nz], counts[rk]
        comb = torch.where(nz != rk, torch.clamp(own + other, max=2**31 - 1), own)
        counts[nz] = comb
        counts[rk] = comb
        if clump.numel():
            ck, cc = torch.unique(clump, return_counts=True)
            ln = counts[ck]
            zero = ck[(ln > 2000) & (cc.float() > 0.75 * ln.float())]
            counts[zero] = 0
            counts[_rc_keys_torch(zero, k)] = 0
        self.counts = counts.to(torch.int32).cpu().numpy()

    @staticmethod
    def _length_histogram(counts, buckets=1000):
        """Tools.makeLengthHistogram3/4 (current/align2/Tools.java:1797-1850), int32 wrap of counts[ptr]*ptr included."""
        mx = int(counts.max()) if len(counts) else 0
        cnt = np.bincount(counts, minlength=mx + 1).astype(np.int64)
        total = int((cnt * np.arange(mx + 1)).sum())
        hist = np.zeros(buckets + 1, np.int32)
        prod = ((cnt * np.arange(mx + 1)) & 0xFFFFFFFF)
        prod = np.where(prod >= 2**31, prod - 2**32, prod)
        csum = np.concatenate(([0], np.cumsum(prod)))        # csum[p] = sum after consuming ptr 0..p-1
        ptr = 0
        for i in range(buckets):
            limit = ((total * i) + buckets // 2) // buckets
            while ptr < mx + 1 and csum[ptr] < limit:
                ptr += 1
            hist[i] = max(0, ptr - 1)
        hist[buckets] = mx
        return hist

    def _set_params(self):
        """BBMap.loadIndex genome-size adjustments (BBMap.java:367-381) + analyzeIndex thresholds (BBIndex.java:176-190)."""
        f = np.float32(0.03)
        p = dict(k=self.k, chromBits=self.chromBits, minChrom=1, maxChrom=self.nchroms, maxIndel=16000, maxIndel2=32000,
                 minApproxHitsToKeep=1, kfilter=0, maxHitsReduction2=2, maximumMaxHitsReduction=3, hitReductionDiv=5,
                 quitAfterTwoPerfects=1, prescanQscore=1, trimByGreedy=1, slow=0)
        n = self.defined_bases
        if n < 300000000:
            p["maxHitsReduction2"] += 1
            p["maximumMaxHitsReduction"] += 1
            if n < 30000000:
                f = f * np.float32(0.5)
                p["maximumMaxHitsReduction"] += 1
                p["hitReductionDiv"] = max(p["hitReductionDiv"] - 1, 3)
            elif n < 100000000:
                f = f * np.float32(0.6)
            else:
                f = f * np.float32(0.75)
        fd = float(f)
        p["maxAverageListToSearch"] = int(1000 * (1 - 2.3 * fd))
        p["maxAverageListToSearch2"] = int(1000 * (1 - 1.4 * fd))
        p["maxShortestListToSearch"] = int(1000 * (1 - 2.8 * fd))
        h = self.length_histogram
        i1 = int((np.float32(1) - f) * np.float32(1000))
        i2 = int((np.float32(1) - f * np.float32(0.25)) * np.float32(1000))
        p["maxUsableLength"] = max(2 * SMALL_GENOME_LIST, int(h[i1]))
        p["maxUsableLength2"] = max(6 * SMALL_GENOME_LIST, int(h[i2]))
        q = np.float32(-50 * 4000.0) / np.float32(max(2 * SMALL_GENOME_LIST, int(h[p["maxAverageListToSearch"]])))
        pps = int(np.floor(np.float64(q)))
        p["pointsPerSite"] = pps if pps != 0 else -1
        self.params = p


# ------------------------------------------------------------------------------------------ C ABI binding
class bbidx_params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "k", "chromBits", "minChrom", "maxChrom", "maxIndel", "maxIndel2", "minApproxHitsToKeep", "kfilter",
        "maxUsableLength", "maxUsableLength2", "maxHitsReduction2", "maximumMaxHitsReduction", "hitReductionDiv",
        "quitAfterTwoPerfects", "prescanQscore", "trimByGreedy", "slow",
        "maxAverageListToSearch", "maxAverageListToSearch2", "maxShortestListToSearch")] + [("pointsPerSite", C.c_int64),
                                                                                             ("profile", C.c_int32), ("reserved", C.c_int32)]


PROFILE_BBMAP, PROFILE_PACBIO = 0, 1        # BBIDX_PROFILE_*


class bbidx_index_desc(C.Structure):
    _fields_ = [("params", bbidx_params), ("nblocks", C.c_int32), ("nchroms", C.c_int32),
                ("starts", C.POINTER(C.c_void_p)), ("sites", C.POINTER(C.c_void_p)), ("numSites", C.POINTER(C.c_int64)),
                ("counts", C.c_void_p), ("lengthHistogram", C.c_void_p),
                ("chromArr", C.POINTER(C.c_void_p)), ("chromArrLen", C.c_void_p), ("chromLengths", C.c_void_p)]


READ_DTYPE = np.dtype([("bases_off", "<i8"), ("keys_off", "<i8"), ("len", "<i4"), ("nkeys", "<i4")])
SITE_DTYPE = np.dtype([("chrom", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("stop", "<i4"), ("hits", "<i4"),
                       ("score", "<i4"), ("perfect", "<i4"), ("semiperfect", "<i4"), ("ngaps", "<i4"),
                       ("gaps", "<i4", (16,))])
assert C.sizeof(bbidx_params) == 96 and READ_DTYPE.itemsize == 24 and SITE_DTYPE.itemsize == 100


class BuiltIndexInfo:
    """What the rest of the Python plumbing needs to know about an index that was built on the device (bbidx_build):
    the chromosomes (host copies, for reference blobs and oracles), k, chromBits and the derived tunables."""

    def __init__(self, chroms, k, chromBits, params):
        self.chroms, self.k, self.chromBits, self.params = chroms, k, chromBits, params
        self.nchroms = len(chroms)
        self.nblocks = (self.nchroms >> chromBits) + 1


class DeviceIndex:
    """Uploads a HostIndex to the GPU (bbidx_create) -- or builds the index there (DeviceIndex.build) -- and runs
    batched probes."""

    @classmethod
    def build(cls, chroms, k=None, chromBits=None, device=0, profile=PROFILE_BBMAP):
        """IndexMaker4 + analyzeIndex on the device (bbidx_build_profile); returns a DeviceIndex whose .host is a BuiltIndexInfo.
        profile: PROFILE_BBMAP (BBIndex, k 13) or PROFILE_PACBIO (BBIndexPacBio, k 12)."""
        if k is None:
            k = 12 if profile == PROFILE_PACBIO else 13
        self = cls.__new__(cls)
        self.L = _lib.load()
        arrs = [np.ascontiguousarray(np.frombuffer(bytes(c), np.uint8)) if not isinstance(c, np.ndarray)
                else np.ascontiguousarray(c, dtype=np.uint8) for c in chroms]
        ptrs = (C.c_void_p * (len(arrs) + 1))(*([0] + [a.ctypes.data for a in arrs]))
        lens = np.array([0] + [len(a) for a in arrs], np.int32)
        h = C.c_void_p()
        self.L.bbidx_build_profile.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p,
                                               C.POINTER(C.c_void_p)]
        self.L.bbidx_build_profile.restype = C.c_int
        _lib.check(self.L.bbidx_build_profile(device, profile, k, -1 if chromBits is None else chromBits, len(arrs), ptrs,
                                              lens.ctypes.data, C.byref(h)), "bbidx_build_profile")
        self.h = h
        self._bind()
        p = bbidx_params()
        _lib.check(self.L.bbidx_get_params(self.h, C.byref(p)), "bbidx_get_params")
        params = {n: int(getattr(p, n)) for n, _ in bbidx_params._fields_}
        self.host = BuiltIndexInfo(arrs, k, params["chromBits"], params)
        return self

    def export_block(self, block=0):
        """(starts, sites, counts, lengthHistogram) of one block, copied back from the device."""
        nkeys = 1 << (2 * self.host.k)
        starts = np.zeros(nkeys + 1, np.int32)
        counts = np.zeros(nkeys, np.int32)
        hist = np.zeros(1001, np.int32)
        _lib.check(self.L.bbidx_export_block(self.h, block, starts.ctypes.data, None, 0, counts.ctypes.data, hist.ctypes.data),
                   "bbidx_export_block")
        sites = np.zeros(max(1, int(starts[nkeys])), np.int32)
        _lib.check(self.L.bbidx_export_block(self.h, block, None, sites.ctypes.data, len(sites), None, None), "bbidx_export_block")
        return starts, sites[: int(starts[nkeys])], counts, hist

    def _bind(self):
        L = self.L
        L.bbidx_destroy.argtypes = [C.c_void_p]
        L.bbidx_find_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.bbidx_find_batch.restype = C.c_int
        L.bbidx_find_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.bbidx_find_batch_device.restype = C.c_int
        L.bbidx_find_batch_device_rc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.bbidx_find_batch_device_rc.restype = C.c_int
        L.bbidx_set_kernel.argtypes = [C.c_void_p, C.c_int32]
        L.bbidx_set_kernel.restype = C.c_int
        L.bbidx_get_params.argtypes = [C.c_void_p, C.POINTER(bbidx_params)]
        L.bbidx_get_params.restype = C.c_int
        L.bbidx_export_block.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.bbidx_export_block.restype = C.c_int

    def __init__(self, host, device=0):
        self.L = _lib.load()
        self.host = host
        d = bbidx_index_desc()
        for k_, v in host.params.items():
            setattr(d.params, k_, int(v))
        d.nblocks, d.nchroms = host.nblocks, host.nchroms
        self._starts = (C.c_void_p * host.nblocks)(*[a.ctypes.data for a in host.starts])
        self._sites = (C.c_void_p * host.nblocks)(*[a.ctypes.data if len(a) else 0 for a in host.sites])
        self._num = (C.c_int64 * host.nblocks)(*[len(a) for a in host.sites])
        self._chr = (C.c_void_p * (host.nchroms + 1))(*([0] + [c.ctypes.data for c in host.chroms]))
        self._clen = np.array([0] + [len(c) for c in host.chroms], np.int32)
        d.starts, d.sites, d.numSites = self._starts, self._sites, self._num
        d.counts = host.counts.ctypes.data
        d.lengthHistogram = host.length_histogram.ctypes.data
        d.chromArr = self._chr
        d.chromArrLen = self._clen.ctypes.data
        d.chromLengths = self._clen.ctypes.data
        h = C.c_void_p()
        self.L.bbidx_create.argtypes = [C.c_int32, C.POINTER(bbidx_index_desc), C.POINTER(C.c_void_p)]
        self.L.bbidx_create.restype = C.c_int
        self.L.bbidx_destroy.argtypes = [C.c_void_p]
        self.L.bbidx_find_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        self.L.bbidx_find_batch.restype = C.c_int
        self.L.bbidx_find_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        self.L.bbidx_find_batch_device.restype = C.c_int
        _lib.check(self.L.bbidx_create(device, C.byref(d), C.byref(h)), "bbidx_create")
        self.h = h
        self._bind()

    def set_kernel(self, kind):
        """kind: "auto" (one read per wavefront, per-lane kernel for the reads that do not fit), "lane", or "long" (the
        long-read kernel: up to 6016 bases and 2047 keys per read; what a PROFILE_PACBIO index always runs)."""
        _lib.check(self.L.bbidx_set_kernel(self.h, {"auto": 0, "lane": 1, "long": 2}[kind]), "bbidx_set_kernel")

    def set_max_read_len(self, max_len):
        """Sizing hint for the wavefront kernel (reads of at most 160 bases: 8 waves per SIMD instead of 6)."""
        self.L.bbidx_set_max_read_len.argtypes = [C.c_void_p, C.c_int32]
        self.L.bbidx_set_max_read_len.restype = C.c_int
        _lib.check(self.L.bbidx_set_max_read_len(self.h, int(max_len)), "bbidx_set_max_read_len")

    def close(self):
        if getattr(self, "h", None):
            self.L.bbidx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find_batch(self, reads, max_sites=32):
        """reads: list of (basesP, baseScoresP, keyScoresP, offsets).  Returns list of lists of site dicts
        (None where the probe reported an overflow / unsupported read)."""
        bases, bscores, keyinfo = bytearray(), bytearray(), []
        recs = np.zeros(len(reads), READ_DTYPE)
        for i, (b, bs, ks, of) in enumerate(reads):
            recs[i] = (len(bases), len(keyinfo), len(b), len(of))
            bases += bytes(b)
            bscores += np.asarray(bs, np.int8).tobytes()
            keyinfo += list(of) + list(ks)
        bases_a = np.frombuffer(bytes(bases) or b"\0", np.uint8)
        bs_a = np.frombuffer(bytes(bscores) or b"\0", np.int8)
        ki = np.array(keyinfo or [0], np.int32)
        sites = np.zeros((len(reads), max_sites), SITE_DTYPE)
        ns = np.zeros(len(reads), np.int32)
        rc = self.L.bbidx_find_batch(self.h, len(reads), recs.ctypes.data, bases_a.ctypes.data, bs_a.ctypes.data,
                                     len(bases), ki.ctypes.data, len(keyinfo), sites.ctypes.data, max_sites, ns.ctypes.data)
        _lib.check(rc, "bbidx_find_batch")
        out = []
        for i in range(len(reads)):
            if ns[i] < 0:
                out.append(None)
                continue
            out.append([dict(chrom=int(s["chrom"]), strand=int(s["strand"]), start=int(s["start"]), stop=int(s["stop"]),
                             hits=int(s["hits"]), score=int(s["score"]), perfect=int(s["perfect"]),
                             semiperfect=int(s["semiperfect"]), gaps=s["gaps"][:s["ngaps"]].tolist())
                        for s in sites[i, :ns[i]]])
        return out
