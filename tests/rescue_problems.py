"""Rescue-scan problems shared by the CPU and GPU tests."""
import random


def make_problems(seed, n, ref_len=20000):
    rng = random.Random(seed)
    body = bytearray(rng.choice(b"ACGT") for _ in range(ref_len))
    for _ in range(4):                                   # a few tandem / dispersed repeats: several equally good starts
        p, q = rng.randrange(500, ref_len - 900), rng.randrange(500, ref_len - 900)
        body[q:q + 300] = body[p:p + 300]
    p = rng.randrange(2000, ref_len - 2000)
    body[p:p + 60] = b"N" * 60
    ref = bytes(b"N" * 300 + body + b"N" * 300)
    probs = []
    for i in range(n):
        L = rng.choice([150, 150, 100, 75, 250, 40])
        true = rng.randrange(400, len(ref) - L - 400)
        rd = bytearray(ref[true:true + L])
        kind = rng.random()
        if kind < 0.3:
            pass
        elif kind < 0.7:
            for _ in range(rng.randint(1, 12)):
                rd[rng.randrange(L)] = rng.choice(b"ACGTN")
        elif kind < 0.8:
            rd = bytearray(rng.choice(b"ACGT") for _ in range(L))       # unrelated read: usually no rescue
        else:
            del rd[L // 2:L // 2 + 2]                                     # small deletion: half the read shifts
            rd += ref[true + L:true + L + 2]
        right = rng.random() < 0.5
        dist = rng.choice([200, 600, 1200, 3000])
        off = rng.randrange(0, dist + 100)
        loc = true - off if right else true + off
        ideal = true + rng.randrange(-80, 80)
        mam = rng.choice([L // 4, L // 8, 3, 0])
        probs.append((bytes(rd), 1, loc, dist, right, ideal, mam))
    # edges: search window clipped by both chromosome ends, read shorter than 10
    probs.append((ref[350:500], 1, 10, 400, True, 350, 5))
    probs.append((ref[len(ref) - 500:len(ref) - 350], 1, len(ref) - 100, 600, False, len(ref) - 500, 5))
    probs.append((ref[1000:1008], 1, 900, 300, True, 1000, 2))
    return ref, probs
