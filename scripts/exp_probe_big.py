"""Probe work split (prescan vs walk entries) on a large multi-scaffold reference."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.pipeline import MapPipeline
nsc, per, n = int(sys.argv[1]), int(sys.argv[2]), 200000
chroms = [W.make_reference(per, seed=1000 + i, repeat_frac=0.1, families=max(50, 2000 // nsc)) for i in range(nsc)]
reads = np.concatenate([W.make_reads_and_jobs(c, n // nsc + 1, read_len=150, seed=2 + 7 * i)[0] for i, c in enumerate(chroms)])[: n * 150]
offsets = W.make_offsets(150, 13, 1.9)
t = time.time(); di = DeviceIndex.build(chroms, k=13); torch.cuda.synchronize(); print("build %.2f s, blocks %d, params maxUsableLength %d" % (time.time() - t, di.host.nblocks, di.host.params["maxUsableLength"]))
pipe = MapPipeline(di, n, 150, offsets, [1300] * len(offsets), max_sites=16, max_columns=256)
pipe.load_reads(reads)
for _ in range(2):
    nj = pipe.step()
st, ms = pipe.probe_stats()
print("raw stats", st)
print("probe %.1f ms for %d reads; prescan entries %d, walk entries %d, extend calls %d; per read %.0f / %.0f" % (ms, n, st[0], st[1], st[2], st[0] / n, st[1] / n))
