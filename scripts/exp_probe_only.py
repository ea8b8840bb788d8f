"""Probe kernel alone (bbidx_find_batch_device) on a synthetic genome: python scripts/exp_probe_only.py [genome_bp] [n_reads] [max_sites]
Written against the C ABI only, so that it also runs in a checkout of an older commit (A/B of the kernel across the history)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bbmap_amd import _lib
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 4641652
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
max_sites = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L, k = 150, 13
ref = W.make_reference(bp, seed=1)
reads = W.make_reads_and_jobs(ref, n, read_len=L, seed=2)[0]
di = DeviceIndex.build([ref], k=k)
if hasattr(di, "set_max_read_len"):
    di.set_max_read_len(L)
dev = torch.device("cuda", 0)
offs = W.make_offsets(L, k, 1.9)
nk = len(offs)
recs = np.zeros(n, np.dtype([("bases_off", "<i8"), ("keys_off", "<i8"), ("len", "<i4"), ("nkeys", "<i4")]))
recs["bases_off"] = np.arange(n, dtype=np.int64) * L
recs["len"] = L
recs["nkeys"] = nk
d_reads = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(dev)
d_bases = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1)).to(dev)
d_bs = torch.zeros(n * L, dtype=torch.int8, device=dev)
d_keyinfo = torch.tensor(list(offs) + [100 * k] * nk, dtype=torch.int32, device=dev)
d_sites = torch.zeros(n * max_sites * 100, dtype=torch.uint8, device=dev)
d_nsites = torch.zeros(n, dtype=torch.int32, device=dev)
Lb = _lib.load()
f = Lb.bbidx_find_batch_device
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]
stream = torch.cuda.current_stream().cuda_stream
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    rc = f(di.h, C.c_void_p(stream), n, d_reads.data_ptr(), d_bases.data_ptr(), d_bs.data_ptr(), d_keyinfo.data_ptr(), d_sites.data_ptr(), max_sites, d_nsites.data_ptr())
    torch.cuda.synchronize()
    print("rc", rc, "probe %.3f ms" % (1e3 * (time.perf_counter() - t)), "sites", int(d_nsites.clamp(min=0).sum()), flush=True)
