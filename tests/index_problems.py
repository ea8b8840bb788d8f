"""Synthetic genomes and reads for the index-probe tests (shared by the CPU and GPU tests)."""
import random

COMP = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}


def revcomp(s):
    return bytes(COMP.get(b, b) for b in reversed(s))


def make_genome(seed, length, repeats=True, pad=400):
    rng = random.Random(seed)
    body = bytearray(rng.choice(b"ACGT") for _ in range(length))
    if repeats:                                   # a few repeat families so that some k-mer lists are long
        for _ in range(6):
            fam = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(150, 600)))
            for _ in range(rng.randint(3, 12)):
                p = rng.randrange(0, length - len(fam))
                cp = bytearray(fam)
                for _ in range(len(fam) // 40):
                    cp[rng.randrange(len(cp))] = rng.choice(b"ACGT")
                body[p:p + len(fam)] = cp
        p = rng.randrange(0, length - 400)
        body[p:p + 300] = b"AC" * 150              # a clumpy low-complexity stretch
        p = rng.randrange(0, length - 400)
        body[p:p + 200] = b"N" * 200               # an internal N run
    return b"N" * pad + bytes(body) + b"N" * pad


def make_reads(seed, genomes, n, read_len=150, k=13, density=1.9):
    """Returns list of (basesP, basesM, baseScores, keyScores, offsets, truth) with truth = (chrom, strand, start)."""
    from oracle.oracle import make_offsets            # KeyRing.makeOffsets restated in the oracle (host-side input generator)
    rng = random.Random(seed)
    out = []
    for i in range(n):
        ci = rng.randrange(len(genomes))
        G = genomes[ci]
        L = rng.choice([read_len, read_len, 100, 75]) if read_len >= 100 else read_len
        lo, hi = min(20, L // 3), max(L - 20, L // 3 + 1)
        st = rng.randrange(300, len(G) - L - 700)
        rd = bytearray(G[st:st + L + 40])
        kind = rng.random()
        if kind < 0.35:
            pass
        elif kind < 0.6:
            for _ in range(rng.randint(1, 3)):
                p = rng.randrange(L)
                rd[p] = rng.choice(b"ACGT")
        elif kind < 0.72:
            p = rng.randrange(lo, hi)
            del rd[p:p + rng.randint(1, 8)]
        elif kind < 0.84:
            p = rng.randrange(lo, hi)
            rd[p:p] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 5)))
        elif kind < 0.9:
            rd[rng.randrange(L)] = ord("N")
        elif kind < 0.95:
            rd = bytearray(rng.choice(b"ACGT") for _ in range(L + 40))      # junk read
        else:
            p = rng.randrange(lo, hi)                                        # long deletion: gap arrays
            extra = bytearray(G[st + L + 40: st + L + 40 + 700])
            rd = rd[:p] + (rd + extra)[p + rng.randint(300, 600):]
        rd = bytes(rd[:L])
        if len(rd) < L or b"N" * 20 in rd:
            continue
        strand = 1 if rng.random() < 0.5 else 0
        bp = revcomp(rd) if strand else rd
        offs = make_offsets(L, k, density)
        mode = rng.random()
        if mode < 0.5:
            ks = [100 * k] * len(offs)
            bs = [0] * L
        else:
            ks = [rng.randint(100 * k // 8, 100 * k) for _ in offs]
            bs = [rng.randint(0, 30) for _ in range(L)]
        out.append((bp, revcomp(bp), bs, ks, offs, (ci + 1, strand, st)))
    return out
