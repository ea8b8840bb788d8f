"""What a host with several mapping threads (BBMap has one per core, each with its own read lists) gets from ONE GPU: T Mapper
contexts on the same index, each with n / T reads, stepped concurrently from T Python threads (ctypes releases the GIL).
python scripts/exp_two_threads.py [hg38] [n_reads] [T ...]"""
import json, sys, threading, time
sys.path.insert(0, ".")
import numpy as np
import torch
import bench as B
from bbmap_amd import keys as K
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper

name = sys.argv[1] if len(sys.argv) > 1 else "hg38"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
Ts = [int(x) for x in sys.argv[3:]] or [1, 2]
lens, paired, _ = B.WORKLOADS[name]
chroms, _ = B.shared_reference(name, lens, 0.0 if name == "ecoli" else 0.1, 0, 1)
reads = np.ascontiguousarray(B.make_batch(chroms, n, paired, 4)).reshape(-1, 150)
di = DeviceIndex.build(chroms, k=13)
offs, ks, _ = K.make_keys(np.frombuffer(b"ACGT" * 38, np.uint8)[:150])
for T in Ts:
    per = (n // T) & ~1
    mps = []
    for t in range(T):
        mp = Mapper(di, per, 150, offs, ks, paired=paired, max_sites=32)
        mp.load_reads(reads[t * per:(t + 1) * per])
        mps.append(mp)
    streams = [torch.cuda.Stream() for _ in mps]             # (non-blocking streams: two contexts on the legacy default stream serialize)
    def run(mp, st=None):
        with torch.cuda.stream(streams[mps.index(mp)]):
            mp.step()
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(mp,)) for mp in mps]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        print("T=%d step %d: %.1f ms for %d reads = %.2f M reads/s" % (T, it, wall, per * T, per * T / wall / 1e3), flush=True)
    # the records must be those of the single context, whatever ran beside what
    fins = [mp.final() for mp in mps]
    recs = np.concatenate([f[0] for f in fins])
    strs = [bytes(b[int(r["match_off"]):int(r["match_off"]) + int(r["match_len"])]) for f, b in fins for r in f[:2000]]
    key = [recs[c] for c in ("mapped", "chrom", "strand", "start", "stop", "mapScore", "paired", "ambiguous", "perfect", "rescued", "match_len")]
    if T == Ts[0]:
        ref_key, ref_n = key, per * T
    else:
        m = min(ref_n, per * T)
        same = all(np.array_equal(a[:m], b[:m]) for a, b in zip(ref_key, key))
        print("T=%d: final records of the first %d reads equal the first run's: %s; mapped %d" % (T, m, same, int(recs["mapped"].sum())), flush=True)
    for mp in mps: mp.close()
