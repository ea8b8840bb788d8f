"""Probe-kernel time by read class: all perfect vs the mutated mix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.pipeline import MapPipeline
n = 1000000
ref = W.make_reference(W.ECOLI_K12_LEN, seed=1)
offsets = W.make_offsets(150, 13, 1.9)
di = DeviceIndex.build([ref], k=13)
pipe = MapPipeline(di, n, 150, offsets, [1300] * len(offsets), max_sites=8, max_columns=256)
for pf in (1.0, 0.5, 0.0):
    reads, _, _ = W.make_reads_and_jobs(ref, n, read_len=150, seed=2, perfect_frac=pf)
    pipe.load_reads(reads)
    for _ in range(3):
        nj = pipe.step()
    st, ms = pipe.probe_stats()
    print("perfect_frac %.1f: probe %.2f ms, jobs %d, extend calls %d, list entries %d" % (pf, ms, nj, st[2], st[0] + st[1]))
