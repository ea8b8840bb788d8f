"""Stage timings of the device mapper on a synthetic genome: python scripts/exp_mapper.py [ref_len] [n_reads] [paired 0/1] [scaffolds]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper

ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else W.ECOLI_K12_LEN
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
paired = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scaf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
max_sites = int(sys.argv[5]) if len(sys.argv) > 5 else 32
t = time.time()
chroms = [W.make_reference(ref_len // scaf, seed=1000 + i, repeat_frac=0.1, families=max(50, 2000 // scaf)) for i in range(scaf)]
print("reference %.1fs" % (time.time() - t), flush=True)
t = time.time()
if paired:
    parts = [W.make_pairs(c, n // 2 // scaf + 1, seed=3 + 7 * i)[0] for i, c in enumerate(chroms[:max(1, min(scaf, 4))])]
    reads = np.concatenate(parts)[: n * 150] if len(parts) > 1 else parts[0][: n * 150]
    if reads.size < n * 150:
        reads = np.resize(reads, n * 150)
else:
    reads = W.make_reads_and_jobs(chroms[0], n, seed=2)[0]
print("reads %.1fs" % (time.time() - t), flush=True)
t = time.time()
di = DeviceIndex.build(chroms, k=13)
print("index %.1fs" % (time.time() - t), flush=True)
offs = W.make_offsets(150, 13, 1.9)
mp = Mapper(di, n, 150, offs, [1300] * len(offs), paired=bool(paired), max_sites=max_sites)
mp.load_reads(reads)
for i in range(3):
    t = time.time()
    mp.step()
    wall = time.time() - t
    st = mp.stats()
    print(json.dumps({"wall_ms": 1e3 * wall, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}}), flush=True)
