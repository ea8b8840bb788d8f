"""PCIe-inclusive rate of the pipeline: upload the read batch, run one step, download everything a host would need
(site counts, site records, DP results and match strings)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W, msa as M
from bbmap_amd.index import DeviceIndex, SITE_DTYPE
from bbmap_amd.pipeline import MapPipeline
n = 1000000
ref = W.make_reference(W.ECOLI_K12_LEN, seed=1)
reads, _, _ = W.make_reads_and_jobs(ref, n, read_len=150, seed=2)
offsets = W.make_offsets(150, 13, 1.9)
di = DeviceIndex.build([ref], k=13)
pipe = MapPipeline(di, n, 150, offsets, [1300] * len(offsets), max_sites=8, max_columns=256)
pinned_reads = torch.from_numpy(np.ascontiguousarray(reads)).pin_memory()
h_nsites = torch.empty(n, dtype=torch.int32).pin_memory()
h_sites = torch.empty(n * 8 * SITE_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
cap = n * 8
h_res = torch.empty(cap * M.RESULT_DTYPE.itemsize // 8, dtype=torch.uint8).pin_memory()      # room for n jobs
h_match = torch.empty(cap * pipe.match_stride // 8, dtype=torch.uint8).pin_memory()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.bases[: pipe.total_bytes].copy_(pinned_reads, non_blocking=True)
    nj = pipe.step()
    h_nsites.copy_(pipe.nsites, non_blocking=True)
    h_sites.copy_(pipe.sites, non_blocking=True)
    h_res[: nj * M.RESULT_DTYPE.itemsize].copy_(pipe.results[: nj * M.RESULT_DTYPE.itemsize], non_blocking=True)
    h_match[: nj * pipe.match_stride].copy_(pipe.match[: nj * pipe.match_stride], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gb = (pinned_reads.numel() + h_nsites.numel() * 4 + h_sites.numel() + nj * (M.RESULT_DTYPE.itemsize + pipe.match_stride)) / 1e9
    print("iter %d: %.1f ms, %.2f M reads/s PCIe-inclusive, %.2f GB moved over PCIe" % (it, dt * 1e3, n / dt / 1e6, gb))
