import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_mapper_gpu import _run, _repeat_workload
from tests.mapper_check import gpu_fills, oracle_fills
ref, reads, L, k = _repeat_workload(False)
out, orc, st, n = _run(ref, reads, L, k, paired=False, max_sites=4, cap=1024)
print(st)
t = out["overflow"]
ids = t["read_ids"]
gf = {(int(ids[i]), s): v for (i, s), v in gpu_fills(t).items()}
of = oracle_fills(orc)
cnt = 0
for key in sorted(gf):
    if key in of and gf[key]["match"] != of[key]["match"]:
        a, b = gf[key], of[key]
        i = a["index"]
        g = i >> 30 & 1
        j = i & ~(1 << 30)
        res = t["gresults"][j] if g else t["results"][j]
        job = t["gjobs"][j] if g else t["jobs"][j]
        oi = b["index"]
        print(key, "gapped" if g else "plain", "dev match_len", int(res["match_len"]), "status", int(res["status"]) if "status" in res.dtype.names else None,
              "oracle match_len", int(orc["log"]["match_len"][oi]), "ngaps", int(orc["log"]["ngaps"][oi]), "win", a["refStartLoc"], a["refEndLoc"], "score", a["score"], "flags", hex(int(job["flags"])))
        cnt += 1
        if cnt > 12: break
print("strides", t["match_stride"], t["gmatch_stride"], orc["match"].shape)
