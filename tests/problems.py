"""Seeded problem generators shared by the oracle tests and the GPU parity tests."""
import random

ACGT = b"ACGT"


def rand_seq(rng, n):
    return bytes(rng.choice(ACGT) for _ in range(n))


def mutate(rng, s, max_events=4, n_prob=0.1):
    s = bytearray(s)
    for _ in range(rng.randint(0, max_events)):
        k = rng.random()
        if len(s) < 12:
            break
        pos = rng.randrange(5, len(s) - 5)
        if k < 0.5:
            s[pos] = rng.choice(ACGT)
        elif k < 0.7:
            del s[pos:pos + rng.randint(1, 6)]
        elif k < 1.0 - n_prob:
            s[pos:pos] = bytes(rng.choice(ACGT) for _ in range(rng.randint(1, 4)))
        else:
            s[pos] = ord("N")
    return bytes(s)


def max_quality(n):
    return 70 + 100 * (n - 1)


def survey_problem_stream(seed=123, count=400):
    """The exact 400-problem stream of the survey's stale-matrix experiment.
    Yields (read, ref, a, b, minScore, bandwidth, bandwidthRatio)."""
    rng = random.Random(seed)

    def mut(s):
        s = bytearray(s)
        for _ in range(rng.randint(0, 4)):
            k = rng.random()
            pos = rng.randrange(5, len(s) - 5)
            if k < 0.5:
                s[pos] = rng.choice(b"ACGT")
            elif k < 0.7:
                del s[pos:pos + rng.randint(1, 6)]
            elif k < 0.9:
                s[pos:pos] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 4)))
            else:
                s[pos] = ord("N")
        return s

    for _ in range(count):
        G = bytes(rng.choice(b"ACGT") for _ in range(1200))
        L0 = rng.choice([100, 150, 150, 150, 250])
        st = rng.randrange(300, 600)
        rd = mut(G[st:st + L0])
        if len(rd) < 60:
            continue
        pad = rng.choice([4, 8, 14])
        extra = rng.choice([0, 0, 0, 30, 120])
        a = st - pad
        b = st + L0 + extra + pad - 1
        ms = int(rng.choice([0.4, 0.56, 0.7]) * (70 + 100 * (len(rd) - 1))) - 120
        bw, bwr = rng.choice([(0, 0.0), (0, 0.0), (40, 0.18)])
        yield bytes(rd), G, a, b, ms, bw, bwr


def mixed_problems(seed, count, read_lens=(100, 150, 150, 250), ref_len=1200, allow_fail=True,
                   with_n=True, max_extra=120):
    """Realistic slow-align problems: a read cut from a random reference, mutated, aligned against a
    padded window around its origin (pad 4/8/14 like SLOW_ALIGN_PADDING / SLOW_RESCUE_PADDING), plus a
    share of wrong-site windows that must fail."""
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        G = rand_seq(rng, ref_len)
        L0 = rng.choice(read_lens)
        st = rng.randrange(200, ref_len - L0 - 200)
        rd = mutate(rng, G[st:st + L0], n_prob=0.1 if with_n else 0.0)
        if len(rd) < 40:
            continue
        pad = rng.choice([4, 8, 14])
        extra = rng.choice([0, 0, 0, 30, max_extra])
        if allow_fail and rng.random() < 0.2:
            st = rng.randrange(100, ref_len - L0 - extra - 100)     # wrong site
        a = st - pad
        b = min(ref_len - 1, st + L0 + extra + pad - 1)
        ratio = rng.choice([0.3, 0.4, 0.56, 0.7, 0.9])
        ms = int(ratio * max_quality(len(rd)))
        out.append((rd, G, a, b, ms))
    return out
