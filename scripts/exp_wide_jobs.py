"""The wide pass alone: N limited fills of 150-base reads against windows of W columns on a context with 3000 columns
(python scripts/exp_wide_jobs.py N W ...): kernel milliseconds of the launch sequence (narrow, wavefront passes, generic)."""
import random, sys, time
sys.path.insert(0, ".")
import numpy as np
from bbmap_amd import msa as M

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = random.Random(7)
ref = bytes(rng.choice(b"ACGT") for _ in range(200000))
al = M.MultiStateAligner11ts(maxRows=160, maxColumns=3000)
for W in [int(x) for x in sys.argv[2:]] or [700, 1000, 2000, 3000]:
    probs = []
    for i in range(N):
        st = rng.randrange(100, 190000 - W)
        d = W - 170                                   # a read with one long deletion inside a window of W columns
        rd = bytearray(ref[st + 10:st + 85] + ref[st + 85 + d:st + 160 + d])
        for _ in range(3):
            rd[rng.randrange(150)] = rng.choice(b"ACGT")
        probs.append((bytes(rd), ref, st, st + W - 1, int(0.5 * (70 + 149 * 100))))
    flags = M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK
    al.align(probs[:64], flags)
    t = time.perf_counter()
    res = al.align(probs, flags)
    wall = 1e3 * (time.perf_counter() - t)
    print("W=%d N=%d: kernels (narrow, wavefront, generic) = %s ms, wall %.1f ms, counts %s, ok %d" % (
        W, N, ["%.2f" % x for x in al.ctx.last_kernel_ms3()], wall, al.ctx.last_counts(), sum(1 for r in res if r["score"] is not None)), flush=True)
