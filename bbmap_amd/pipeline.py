"""Device-resident read pipeline: reverse complement -> index probe -> site filter -> slow-align DP.

Everything between the upload of a read batch and the download of its results stays in HBM; the only host
round trip per step is the 16-byte counter block that tells the host how many DP jobs the filter produced.
Used by bench.py and the pipeline tests.  Needs torch (device buffers, stream) and the HIP library.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import msa as M
from .index import READ_DTYPE, SITE_DTYPE, DeviceIndex


class MapPipeline:
    def __init__(self, host_index, n_reads, read_len, offsets, key_scores, device=0, max_sites=8,
                 max_columns=256, pad=4, min_ratio=0.56, no_iterations=False):
        self.L = _lib.load()
        self.dev = torch.device("cuda", device)
        # host_index: a HostIndex (arrays built on the host, uploaded by bbidx_create) or an already built DeviceIndex
        if isinstance(host_index, DeviceIndex):
            self.di = host_index
            host_index = self.di.host
        else:
            self.di = DeviceIndex(host_index, device)
        self.hi = host_index
        self.n, self.read_len, self.max_sites, self.pad, self.min_ratio = n_reads, read_len, max_sites, pad, min_ratio
        self.di.set_max_read_len(read_len)
        self.max_columns = max_columns
        self.no_iterations = no_iterations       # BBMSA_NO_ITERATIONS on every DP job (visited-cell counters not needed)
        max_rows = ((read_len + 31) // 32) * 32
        self.msa = M.MSAContext(maxRows=max_rows, maxColumns=max_columns, device=device)
        # reference blob = chromosomes back to back; chrom_off[c] = offset of chromosome c
        offs, total = [0], 0
        for c in host_index.chroms:
            offs.append(total)
            total += len(c)
        self.refs = torch.from_numpy(np.concatenate(host_index.chroms)).to(self.dev)
        self.chrom_off = torch.tensor(offs, dtype=torch.int64, device=self.dev)
        self.chrom_len = torch.tensor([0] + [len(c) for c in host_index.chroms], dtype=torch.int32, device=self.dev)
        # reads: plus strand in the first half of `bases`, reverse complements in the second half
        self.total_bytes = n_reads * read_len
        self.bases = torch.zeros(2 * self.total_bytes, dtype=torch.uint8, device=self.dev)
        self.base_scores = torch.zeros(self.total_bytes, dtype=torch.int8, device=self.dev)
        recs = np.zeros(n_reads, READ_DTYPE)
        recs["bases_off"] = np.arange(n_reads, dtype=np.int64) * read_len
        recs["keys_off"] = 0                       # every read uses the same offsets / key scores
        recs["len"] = read_len
        recs["nkeys"] = len(offsets)
        self.reads = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(self.dev)
        self.keyinfo = torch.tensor(list(offsets) + list(key_scores), dtype=torch.int32, device=self.dev)
        self.sites = torch.zeros(n_reads * max_sites * SITE_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.nsites = torch.zeros(n_reads, dtype=torch.int32, device=self.dev)
        cap = n_reads * max_sites
        self.job_cap = cap
        self.jobs = torch.zeros(cap * M.JOB_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.job_src = torch.zeros(cap, dtype=torch.int32, device=self.dev)
        self.counters = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.read_state = torch.zeros(n_reads, dtype=torch.int32, device=self.dev)
        self.ungapped_match = torch.zeros(n_reads * read_len, dtype=torch.uint8, device=self.dev)
        self.ungapped_len = torch.zeros(n_reads, dtype=torch.int32, device=self.dev)
        self.no_indel = torch.zeros(cap, dtype=torch.int32, device=self.dev)
        self.match_stride = ((max_rows + max_columns + 15) // 16) * 16
        self.results = torch.zeros(cap * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.match = torch.zeros(cap * self.match_stride, dtype=torch.uint8, device=self.dev)
        # second job list: sites whose index hit spans a long deletion (gap arrays) align against a gapped reference
        # (makeGref) that can be much wider than an ordinary window, so they get their own, wider MSA context
        self.gap_cap = max(1024, n_reads // 16)
        # BBMap's own maxColumns (BBMapThread.java:27-28); windows beyond the first pass's 1024-column LDS buffer (a handful
        # per million reads) are taken by the context's wide pass
        self.gap_columns = max(3000, max_columns)
        self.gap_fast_cols = max(int(__import__('os').environ.get('BBPIPE_GAP_FAST_COLS', '640')), max_columns)
        self.msa_gapped = M.MSAContext(maxRows=max_rows, maxColumns=self.gap_columns, device=device, fast_cols=self.gap_fast_cols,
                                       lanes_per_job=64)
        self.gjobs = torch.zeros(self.gap_cap * M.JOB_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.ggaps = torch.zeros(self.gap_cap * M.GAPS_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.gjob_src = torch.zeros(self.gap_cap, dtype=torch.int32, device=self.dev)
        self.gmatch_stride = ((max_rows + self.gap_columns + 2 + 128 * 8 + 15) // 16) * 16
        self.gresults = torch.zeros(self.gap_cap * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=self.dev)
        self.gmatch = torch.zeros(self.gap_cap * self.gmatch_stride, dtype=torch.uint8, device=self.dev)
        self.max_rows = max_rows
        self.last_counters = None
        self.last_ms = {}

    def load_reads(self, reads_u8):
        assert reads_u8.size == self.total_bytes
        self.bases[: self.total_bytes].copy_(torch.from_numpy(np.ascontiguousarray(reads_u8)))

    def step(self, sync=True):
        """One pass of the hot path over the resident batch: probe -> site filter -> DP (+ gapped-reference DP), all enqueued on
        the current stream with NO host round trip in between (the DP kernels read their job counts from the counters the
        site filter leaves on the device).  sync=True then waits and returns the number of DP jobs; sync=False returns None
        (call counts() later)."""
        L, n = self.L, self.n
        stream = torch.cuda.current_stream().cuda_stream
        plus = self.bases.data_ptr()
        # the probe writes every read's reverse complement (the DP jobs of minus-strand sites read it) while it has it in LDS
        _lib.check(L.bbidx_find_batch_device_rc(self.di.h, C.c_void_p(stream), n, self.reads.data_ptr(), plus,
                                                self.base_scores.data_ptr(), self.keyinfo.data_ptr(), self.sites.data_ptr(),
                                                self.max_sites, self.nsites.data_ptr(), plus + self.total_bytes),
                   "bbidx_find_batch_device_rc")
        _lib.check(L.bbpipe_select_jobs_device(C.c_void_p(stream), n, self.reads.data_ptr(), plus, self.total_bytes,
                                               self.nsites.data_ptr(), self.sites.data_ptr(), self.max_sites,
                                               self.chrom_off.data_ptr(), self.chrom_len.data_ptr(), self.refs.data_ptr(),
                                               self.pad, self.max_columns, self.min_ratio, self.jobs.data_ptr(),
                                               self.job_src.data_ptr(), self.counters.data_ptr(), self.no_indel.data_ptr(),
                                               self.gjobs.data_ptr(), self.ggaps.data_ptr(), self.gjob_src.data_ptr(), self.gap_cap,
                                               M.NO_ITERATIONS if self.no_iterations else 0, self.read_state.data_ptr(),
                                               # reads that need no DP still need their match string (one symbol per base at the
                                               # best site): the filter writes it on the way
                                               self.ungapped_match.data_ptr(), self.read_len, self.ungapped_len.data_ptr()),
                   "bbpipe_select_jobs_device")
        cptr = self.counters.data_ptr()
        _lib.check(L.bbmsa_align_batch_device_indirect(self.msa.h, C.c_void_p(stream), C.c_void_p(cptr), self.job_cap,
                                                       self.jobs.data_ptr(), plus, self.refs.data_ptr(), self.results.data_ptr(),
                                                       self.match.data_ptr(), self.match_stride),
                   "bbmsa_align_batch_device_indirect")
        _lib.check(L.bbmsa_align_gapped_batch_device_indirect(self.msa_gapped.h, C.c_void_p(stream), C.c_void_p(cptr + 8),
                                                              self.gap_cap, self.gjobs.data_ptr(), self.ggaps.data_ptr(), plus,
                                                              self.refs.data_ptr(), self.gresults.data_ptr(), self.gmatch.data_ptr(),
                                                              self.gmatch_stride), "bbmsa_align_gapped_batch_device_indirect")
        if not sync:
            self.last_counters = None
            return None
        return self.counts()[0]

    def counts(self):
        """Waits for the stream and returns the site filter's counters of the last step: [DP jobs, reads finished without DP,
        gapped-reference jobs, reads without a site]."""
        cnt = self.counters.cpu().numpy()
        if int(cnt[2]) > self.gap_cap:
            raise RuntimeError("more gapped sites (%d) than the pipeline's gapped-job capacity (%d)" % (int(cnt[2]), self.gap_cap))
        self.last_counters = cnt
        return int(cnt[0]), int(cnt[1]), int(cnt[2]), int(cnt[3])

    def probe_stats(self):
        st = (C.c_int64 * 5)()
        ms = C.c_float()
        self.L.bbidx_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        _lib.check(self.L.bbidx_last_stats(self.di.h, st, C.byref(ms)), "bbidx_last_stats")
        return list(st), ms.value

    def fetch(self, njobs):
        """Host copies of everything a step produced (for tests / parity sampling)."""
        sites = self.sites.cpu().numpy().view(SITE_DTYPE).reshape(self.n, self.max_sites)
        nsites = self.nsites.cpu().numpy()
        jobs = self.jobs[: njobs * M.JOB_DTYPE.itemsize].cpu().numpy().view(M.JOB_DTYPE)
        src = self.job_src[:njobs].cpu().numpy()
        res = self.results[: njobs * M.RESULT_DTYPE.itemsize].cpu().numpy().view(M.RESULT_DTYPE)
        match = self.match[: njobs * self.match_stride].cpu().numpy().reshape(njobs, self.match_stride)
        no_indel = self.no_indel.cpu().numpy().reshape(self.n, self.max_sites)
        ngap = int(self.last_counters[2])
        gjobs = self.gjobs[: ngap * M.JOB_DTYPE.itemsize].cpu().numpy().view(M.JOB_DTYPE)
        ggaps = self.ggaps[: ngap * M.GAPS_DTYPE.itemsize].cpu().numpy().view(M.GAPS_DTYPE)
        gsrc = self.gjob_src[:ngap].cpu().numpy()
        gres = self.gresults[: ngap * M.RESULT_DTYPE.itemsize].cpu().numpy().view(M.RESULT_DTYPE)
        gmatch = self.gmatch[: ngap * self.gmatch_stride].cpu().numpy().reshape(ngap, self.gmatch_stride)
        return dict(sites=sites, nsites=nsites, jobs=jobs, src=src, results=res, match=match, no_indel=no_indel,
                    read_state=self.read_state.cpu().numpy(), ungapped_len=self.ungapped_len.cpu().numpy(),
                    ungapped_match=self.ungapped_match.cpu().numpy().reshape(self.n, self.read_len),
                    gjobs=gjobs, ggaps=ggaps, gsrc=gsrc, gresults=gres, gmatch=gmatch)
