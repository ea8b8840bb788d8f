"""One pipeline step, then statistics of the DP jobs it produced (shapes, outcomes, visited fraction)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W
from bbmap_amd.index import HostIndex
from bbmap_amd.pipeline import MapPipeline
n = 200000
ref = W.make_reference(W.ECOLI_K12_LEN, seed=1)
reads, _, _ = W.make_reads_and_jobs(ref, n, read_len=150, seed=2)
offsets = W.make_offsets(150, 13, 1.9)
hi = HostIndex([ref], k=13, backend="torch")
pipe = MapPipeline(hi, n, 150, offsets, [1300] * len(offsets), max_sites=8, max_columns=256)
pipe.load_reads(reads)
nj = pipe.step(); torch.cuda.synchronize()
out = pipe.fetch(nj)
res, jobs = out["results"], out["jobs"]
cols = res["columns"]; rows = jobs["read_len"]
print("jobs", nj, "per read", nj / n)
print("status counts", np.bincount(res["status"]))
print("fill_kind counts", np.bincount(res["fill_kind"]))
print("columns pct", np.percentile(cols, [0, 10, 50, 90, 99, 100]))
vis = res["iterations"] / (rows.astype(np.float64) * cols)
for st in (0, 1):
    m = res["status"] == st
    if m.any():
        print("status", st, "n", m.sum(), "visited frac mean", vis[m].mean(), "pct", np.percentile(vis[m], [10, 50, 90]))
# jobs per read histogram
src = out["src"] // 8
print("jobs per read with jobs", np.bincount(np.bincount(src)[np.bincount(src) > 0]))
print("kernel split", pipe.msa.last_counts(), "ms", pipe.msa.last_kernel_ms())
cnt = pipe.last_counters
print("counters", cnt)
if cnt[2]:
    g = out["gresults"]
    print("gapped: status", np.bincount(g["status"]), "fill_kind", np.bincount(g["fill_kind"]), "columns pct", np.percentile(g["columns"], [0, 50, 90, 99, 100]))
    print("gapped kernel split", pipe.msa_gapped.last_counts(), pipe.msa_gapped.last_kernel_ms())
    gg = out["ggaps"]
    print("ngaps", np.bincount(gg["ngaps"]))
