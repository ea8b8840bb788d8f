"""Row-extent statistics of the slow-align jobs of the bench workload (how wide is the visited band?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.pipeline import MapPipeline
n = 200000
ref = W.make_reference(4641652, seed=1)
reads = W.make_reads_and_jobs(ref, n, read_len=150, seed=2)[0]
offsets = W.make_offsets(150, 13, 1.9)
di = DeviceIndex.build([ref], k=13)
pipe = MapPipeline(di, n, 150, offsets, [1300] * len(offsets), max_sites=8, max_columns=256)
pipe.load_reads(reads)
nj = pipe.step()
f = pipe.fetch(nj)
res, jobs = f["results"], f["jobs"]
rows = jobs["read_len"].astype(np.int64)
cols = res["columns"].astype(np.int64)
it = res["iterations"]
maxq = 70 + (rows - 1) * 100 + 0
print("jobs", nj, "columns mean %.1f" % cols.mean(), "fill_kind1 frac %.4f" % (res["fill_kind"] == 1).mean())
w = it / rows
print("iters/rows percentiles (visited columns per row):", np.percentile(w, [5, 25, 50, 75, 90, 95, 99]).round(1))
print("frac of full matrix visited: %.3f" % (it.sum() / (rows * cols).sum()))
slack = (100 * rows + 70 - 100) - jobs["minScore"]
print("slack percentiles:", np.percentile(slack, [5, 25, 50, 75, 95]).round(0))
for lo, hi in [(0, 2000), (2000, 4000), (4000, 8000), (8000, 1 << 30)]:
    m = (slack >= lo) & (slack < hi)
    if m.any(): print("slack [%d,%d): %d jobs, visited cols/row median %.1f p90 %.1f" % (lo, hi, m.sum(), np.median(w[m]), np.percentile(w[m], 90)))
