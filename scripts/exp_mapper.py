"""Stage timings of the device mapper on one of bench.py's workloads:
python scripts/exp_mapper.py [hg38|chr21|ecoli] [n_reads] [max_sites] [repeat_frac (default: the workload's)]"""
import json
import sys
import time

sys.path.insert(0, ".")
import bench as B
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper

name = sys.argv[1] if len(sys.argv) > 1 else "hg38"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
max_sites = int(sys.argv[3]) if len(sys.argv) > 3 else 32
lens, paired, _ = B.WORKLOADS[name]
rf = float(sys.argv[4]) if len(sys.argv) > 4 else (0.0 if name == "ecoli" else 0.1)
chroms, _ = B.shared_reference(name, lens, rf, 0, 1)
if len(sys.argv) > 5:                      # hard_frac override (share of the mates riddled with substitutions)
    import functools
    W.make_pairs = functools.partial(W.make_pairs, hard_frac=float(sys.argv[5]))
reads = B.make_batch(chroms, n, paired, 4)
t = time.time()
di = DeviceIndex.build(chroms, k=13)
print("index %.1fs" % (time.time() - t), flush=True)
import numpy as np
from bbmap_amd import keys as K
offs, ks, _ = K.make_keys(np.frombuffer(b"ACGT" * 38, np.uint8)[:150])       # quickMap's placement for a read without qualities: 18 keys
import os
extra = {}
if os.environ.get("BBMAP_FASTCOLS"):
    extra["fastCols"] = int(os.environ["BBMAP_FASTCOLS"])
if os.environ.get("BBMAP_TIPSEARCH"):                       # 0 switches findTipDeletions off (what part of the score kernel is it?)
    extra["tipSearchDist"] = int(os.environ["BBMAP_TIPSEARCH"])
mp = Mapper(di, n, 150, offs, ks, paired=paired, max_sites=max_sites, **extra)
mp.load_reads(reads)
for i in range(3):
    t = time.time()
    mp.step()
    wall = time.time() - t
    st = mp.stats()
    print(json.dumps({"wall_ms": 1e3 * wall, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}}), flush=True)

import os
if os.environ.get("BBMAP_G2_HIST"):
    out = mp.fetch(with_match=False)
    g = out["gjobs"]
    w = (np.minimum(g["ref_len"] - 1, g["refEndLoc"]) - np.maximum(0, g["refStartLoc"]) + 1)
    ng = out["ggaps"]["ngaps"]
    print("gapped widths: n=%d with gap array %d; columns pct 10/50/90/99/max = %s; share <=384 %.3f <=512 %.3f <=640 %.3f" % (
        len(g), int((ng > 0).sum()), np.percentile(w, [10, 50, 90, 99, 100]).astype(int).tolist(), (w <= 384).mean(), (w <= 512).mean(), (w <= 640).mean()), flush=True)
