"""Final alignment stage, device against oracle, with counts (GPU box): python scripts/exp_final_check.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bbmap_amd import workload as W
from bbmap_amd.index import DeviceIndex
from bbmap_amd.mapper import Mapper
from oracle import oracle as O
from tests.mapper_check import compare

L, k = 150, 12
for paired in (False, True):
    ref = W.make_reference(300000, seed=5 + paired, pad=2000, repeat_frac=0.15)
    if paired:
        reads, _ = W.make_pairs(ref, 2000, read_len=L, seed=4, pad=2000, hard_frac=0.08)
    else:
        reads, _, _ = W.make_reads_and_jobs(ref, 3000, read_len=L, seed=9, pad=2000, long_del_frac=0.3, hard_frac=0.05)
    di = DeviceIndex.build([ref], k=k)
    offs = O.make_offsets(L, k, 1.9)
    ks = [100 * k] * len(offs)
    n = reads.size // L
    mp = Mapper(di, n, L, offs, ks, paired=paired, max_sites=32)
    mp.load_reads(reads)
    mp.step()
    out, st = mp.fetch(), mp.stats()
    oi = O.OracleIndex([ref], k=k)
    if paired:
        oi.s.p.quitAfterTwoPerfects = 0
        r = reads.reshape(-1, L)
        orc = O.map_batch(oi, r[0::2].copy(), r[1::2].copy(), L, offs, ks, cap=64, match_stride=4200)
    else:
        orc = O.map_batch(oi, reads, None, L, offs, ks, cap=64, match_stride=4200)
    bad = compare(out, orc, n, paired)
    f = out["final"]
    print("paired", paired, "bad", len(bad), "final fills", st["final_fills"], "rounds", st["final_rounds"], "local", st["final_local"], "ms_final %.2f" % st["ms_final"],
          "mapped", int(f["mapped"].sum()), "with match", int((f["match_len"] > 0).sum()), "oracle kinds", np.bincount(orc["log"]["kind"], minlength=7).tolist(),
          "pool bytes", len(out["final_match"]))
    for b in bad[:10]:
        print("  ", b)
    mp.close(); di.close()
