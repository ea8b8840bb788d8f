"""one fill at a time: the matrix-materialising launch against the batch kernel on the same single job (is it the stores or the clock?)"""
import random, sys, time
import numpy as np
sys.path.insert(0, ".")
from bbmap_amd import msa as M
rng = random.Random(3)
ref = bytes(rng.choice(b"ACGT") for _ in range(5000))
rd = bytearray(ref[1000:1150]); rd[70] = ord("A") if rd[70] != ord("A") else ord("C"); rd = bytes(rd)
a, b = 990, 1165
ms = int(0.56 * (70 + 100 * 149))
for (mr, mc) in ((601, 2000), (160, 300)):
    leg = M.MSAContext(maxRows=mr, maxColumns=mc, legacy=True)
    packed = np.zeros(3 * (mr + 1) * (mc + 1), np.int32)
    for _ in range(20):
        leg.fill_packed(rd, ref, a, b, ms, True, packed)
    s0 = leg.legacy_stats(); t0 = time.time()
    for _ in range(300):
        leg.fill_packed(rd, ref, a, b, ms, True, packed)
    dt = time.time() - t0; s1 = leg.legacy_stats()
    print("legacy %dx%d: %.0f us per call, leader wavefront pass %.0f us" % (mr, mc, dt / 300 * 1e6, (s1["wave_ms"] - s0["wave_ms"]) / 300 * 1e3))
    leg.close()
    for lanes in (64, 32, 16):
        if (mr + lanes - 1) // lanes > 10:
            continue
        bat = M.MultiStateAligner11ts(maxRows=mr, maxColumns=mc, lanes_per_job=lanes)
        prob = [(rd, ref, a, b, ms)]
        for _ in range(10):
            bat.align(prob, M.FILL_LIMITED_RAW)
        tot = 0.0
        for _ in range(100):
            bat.align(prob, M.FILL_LIMITED_RAW)
            tot += sum(bat.ctx.last_kernel_ms3())
        print("  batch kernel, same job alone, %d lanes per job: %.0f us of kernels per call" % (lanes, tot / 100 * 1e3))
        bat.ctx.close()
