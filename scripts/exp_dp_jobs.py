"""What the DP stage of one step looks like on a bench workload: slack (maxQuality - minScore), window width and visited share of
the fills in the plain log.  python scripts/exp_dp_jobs.py [workload] [n_reads]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench as B
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper

name = sys.argv[1] if len(sys.argv) > 1 else "hg38"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
lens, paired, _ = B.WORKLOADS[name]
chroms, _ = B.shared_reference(name, lens, 0.0 if name == "ecoli" else 0.1, 0, 1)
reads = B.make_batch(chroms, n, paired, 4)
di = DeviceIndex.build(chroms, k=13)
offs = W.make_offsets(150, 13, 1.9)
mp = Mapper(di, n, 150, offs, [1300] * len(offs), paired=paired, max_sites=32)
mp.load_reads(reads)
mp.step()
out = mp.fetch(with_match=False)
for tag, jobs, res, info in (("plain", out["jobs"], out["results"], out["jobinfo"]), ("second ctx", out["gjobs"], out["gresults"], out["gjobinfo"])):
    cols = jobs["refEndLoc"].astype(np.int64) - jobs["refStartLoc"] + 1
    rows = jobs["read_len"].astype(np.int64)
    slack = 70 + (rows - 1) * 100 - jobs["minScore"]
    vis = res["iterations"].astype(np.float64) / np.maximum(1, rows * cols)
    print("==", tag, "fills", len(jobs), "kinds", np.bincount(info["kind"], minlength=3).tolist())
    print(" cols pct 10/50/90/99/max", np.percentile(cols, [10, 50, 90, 99]).tolist(), int(cols.max()))
    print(" slack pct 10/25/50/75/90", np.percentile(slack, [10, 25, 50, 75, 90]).tolist())
    print(" visited share mean", float(vis.mean()), "pct 10/50/90", np.percentile(vis, [10, 50, 90]).tolist())
    for lo, hi in ((0, 1000), (1000, 2000), (2000, 3000), (3000, 4500), (4500, 7000), (7000, 99999)):
        m = (slack >= lo) & (slack < hi)
        if m.any():
            print("  slack [%d,%d): %5.1f %% of fills, visited share %.3f, cells %5.1f %% of all visited, ok %.2f" % (
                lo, hi, 100.0 * m.mean(), float(vis[m].mean()), 100.0 * res["iterations"][m].sum() / max(1, res["iterations"].sum()),
                float((res["score_len"][m] > 0).mean())))
