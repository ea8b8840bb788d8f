"""CPU side of DESIGN section 9: the oracle's restatement of the native fill (gcc -O2, one core) on the mock test's call mix"""
import random, sys, time
sys.path.insert(0, ".")
from oracle.oracle import OracleMSA
rng = random.Random(23)
ref = bytes(rng.choice(b"ACGT") for _ in range(200000))
om = OracleMSA(601, 2000)
calls = []
for i in range(2000):
    st = rng.randrange(1000, 190000)
    rd = bytearray(ref[st:st + 158])
    for _ in range(rng.randrange(4)):
        rd[rng.randrange(150)] = rng.choice(b"ACGT")
    if i % 5 == 1:
        del rd[70:70 + rng.randint(1, 6)]
    rd = bytes(rd[:150])
    calls.append((rd, st - 8 - rng.randrange(8), st + 158 + rng.randrange(24), i % 6 != 0))
ms = int(0.56 * (70 + 100 * 149))
t0 = time.time()
for rd, a, b, lim in calls:
    if lim:
        om.fill_limited_raw(rd, ref, a, b, ms)
    else:
        om.fill_unlimited_raw(rd, ref, a, b)
dt = time.time() - t0
print("oracle (C restatement of jni/MultiStateAligner11tsJNI.c, one core): %.0f calls/s on the same mix" % (len(calls) / dt))
