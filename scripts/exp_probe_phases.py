"""Debug build (-DBBIDX_PHASE_TIMERS): cycles the probe kernel spends per phase on the bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.pipeline import MapPipeline
n = 1000000
ref = W.make_reference(4641652, seed=1)
reads = W.make_reads_and_jobs(ref, n, read_len=150, seed=2)[0]
offsets = W.make_offsets(150, 13, 1.9)
di = DeviceIndex.build([ref], k=13)
pipe = MapPipeline(di, n, 150, offsets, [1300] * len(offsets), max_sites=8, max_columns=256)
pipe.load_reads(reads)
for _ in range(2): pipe.step()
st, ms = pipe.probe_stats()
tot = sum(st)
print("probe %.2f ms; phase cycle share: keys+lookup %.3f trim/setup %.3f prescan %.3f walk %.3f extend %.3f; cycles/read %.0f" % (
    ms, st[0] / tot, st[1] / tot, st[2] / tot, st[3] / tot, st[4] / tot, tot * 16 / n))
