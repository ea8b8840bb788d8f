"""Python binding of bbpipe_quick_rescue_device (AbstractMapThread.quickRescue, batched) for tests and pipelines."""
import ctypes as C

import numpy as np
import torch

from . import _lib

JOB_DTYPE = np.dtype([("read_off", "<i8"), ("read_len", "<i4"), ("chrom", "<i4"), ("loc", "<i4"), ("searchDist", "<i4"),
                      ("idealStart", "<i4"), ("maxAllowedMismatches", "<i4"), ("flags", "<i4"), ("reserved", "<i4")])
RESULT_DTYPE = np.dtype([(n, "<i4") for n in ("found", "start", "stop", "score", "mismatches", "perfect", "semiperfect", "maxContig")])
assert JOB_DTYPE.itemsize == 40 and RESULT_DTYPE.itemsize == 32


def quick_rescue_batch(problems, chroms, min_index=None, points_match=70, points_match2=100, use_affine=True,
                       base_hit_score=100, device=0):
    """problems: list of (bases, chrom, loc, searchDist, searchRight, idealStart, maxAllowedMismatches);
    chroms: list of chromosome byte arrays (chromosome numbers start at 1).  Returns a list of dicts / None."""
    L = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.BBMapAmdError("quick_rescue_batch needs a GPU: there is no CPU path")
    dev = torch.device("cuda", device)
    reads = bytearray()
    jobs = np.zeros(len(problems), JOB_DTYPE)
    for i, (b, ch, loc, sd, right, ideal, mam) in enumerate(problems):
        jobs[i] = (len(reads), len(b), ch, loc, sd, ideal, mam, 1 if right else 0, 0)
        reads += bytes(b)
    offs, total = [0], 0
    for c in chroms:
        offs.append(total)
        total += len(c)
    refs = torch.from_numpy(np.concatenate([np.frombuffer(bytes(c), np.uint8) for c in chroms])).to(dev)
    t_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    t_len = torch.tensor([0] + [len(c) for c in chroms], dtype=torch.int32, device=dev)
    t_min = torch.tensor([0] + list(min_index or [0] * len(chroms)), dtype=torch.int32, device=dev)
    t_reads = torch.from_numpy(np.frombuffer(bytes(reads) or b"\0", np.uint8).copy()).to(dev)
    t_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
    t_res = torch.zeros(max(1, len(problems)) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(L.bbpipe_quick_rescue_device(C.c_void_p(stream), len(problems), t_jobs.data_ptr(), t_reads.data_ptr(),
                                            t_off.data_ptr(), t_len.data_ptr(), t_min.data_ptr(), refs.data_ptr(),
                                            t_res.data_ptr(), points_match, points_match2, 1 if use_affine else 0,
                                            base_hit_score), "bbpipe_quick_rescue_device")
    res = t_res.cpu().numpy().view(RESULT_DTYPE)[: len(problems)]
    out = []
    for r in res:
        if r["found"] != 1:
            out.append(None)
        else:
            out.append(dict(start=int(r["start"]), stop=int(r["stop"]), score=int(r["score"]), mismatches=int(r["mismatches"]),
                            perfect=int(r["perfect"]), semiperfect=int(r["semiperfect"]), contig=int(r["maxContig"])))
    return out
