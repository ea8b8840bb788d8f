"""Inputs for the tests of the final alignment stage (genMatchString -> realign_new, clipping, toLocalAlignment).

On ordinary reads nearly all of that stage's branches stay cold: an imperfect read needs ONE fill and that is it (40,000 pairs of the
bench workload: one padded refill, no third fill, no X / Y / C symbol).  So the tests feed the stage ALONE (bbmap_final_batch_device /
oracle.final_reads) with site lists the mapper would rarely produce:
  * reads with insertions, deletions, junk and Ns close to their tips (tip_reads),
  * site lists whose top site is shifted, cut short or far too wide, or is followed by a site that outscores what the top one will
    get (perturb),
  * reads that hang over the ends of the chromosome array, with a site planted there (edge_reads): the only way to X / Y symbols that
    survive to fixXY, 'C' clipping and toLocalAlignment,
  * a small MSA (msaMaxColumns 250): realign_new's padding arithmetic then shrinks windows and reaches its third fill and fillUnlimited.
The oracle counts which branches ran (oracle.final_branch_counts), and the tests assert the counts, so a test that stops exercising a
branch fails instead of passing vacuously."""
import numpy as np

from bbmap_amd import workload as W

ACGT = np.frombuffer(b"ACGT", np.uint8)


def tip_reads(ref, n, L, seed, pad):
    """n reads of L bases drawn from ref (uint8, `pad` bases of padding at both ends), each with one kind of damage near its tips;
    half of them reverse-complemented.  Returns uint8[n, L]."""
    rng = np.random.default_rng(seed)
    reads = np.zeros((n, L), np.uint8)
    lo, hi = pad + 300, len(ref) - pad - 1000
    for i in range(n):
        a = int(rng.integers(lo, hi))
        seg = ref[a:a + L + 200].copy()
        kind = int(rng.integers(0, 8))
        r = seg
        if kind == 0:        # insertion close to the left tip
            p, m = int(rng.integers(0, 14)), int(rng.integers(1, 45))
            r = np.concatenate([seg[:p], ACGT[rng.integers(0, 4, m)], seg[p:]])
        elif kind == 1:      # insertion close to the right tip
            p, m = int(rng.integers(0, 14)), int(rng.integers(1, 45))
            r = np.concatenate([seg[:L - m - p], ACGT[rng.integers(0, 4, m)], seg[L - m - p:]])
        elif kind == 2:      # deletion close to the left tip
            p, m = int(rng.integers(1, 8)), int(rng.integers(1, 70))
            r = np.concatenate([seg[:p], seg[p + m:]])
        elif kind == 3:      # deletion close to the right tip
            p, m = int(rng.integers(1, 8)), int(rng.integers(1, 70))
            r = np.concatenate([seg[:L - p], seg[L - p + m:]])
        elif kind == 4:      # junk tips
            k1, k2 = int(rng.integers(0, 14)), int(rng.integers(0, 14))
            r = seg.copy()
            r[:k1] = ACGT[rng.integers(0, 4, k1)]
            r[L - k2:L] = ACGT[rng.integers(0, 4, k2)]
        elif kind == 5:      # insertion left, deletion right
            p, m = int(rng.integers(0, 4)), int(rng.integers(6, 14))
            r = np.concatenate([seg[:p], ACGT[rng.integers(0, 4, m)], seg[p:]])
            q, d = int(rng.integers(2, 6)), int(rng.integers(20, 60))
            r = np.concatenate([r[:L - q], r[L - q + d:]])
        elif kind == 6:      # Ns in the tip and an insertion
            r = seg.copy()
            r[:int(rng.integers(0, 5))] = ord("N")
            p, m = int(rng.integers(3, 8)), int(rng.integers(6, 12))
            r = np.concatenate([r[:p], ACGT[rng.integers(0, 4, m)], r[p:]])
        else:                # a few substitutions
            r = seg.copy()
            for q in rng.integers(0, L, 3):
                r[q] = ACGT[(int(np.searchsorted(ACGT, r[q])) + 1) % 4]
        r = r[:L].copy()
        if rng.random() < 0.5:
            r = W.revcomp_rows(r.reshape(1, -1))[0]
        reads[i] = r
    return reads


def edge_reads(ref, n, L, seed):
    """Reads that hang 1..29 bases over one end of the chromosome array `ref`, with the site a mapper would have to report for them.
    Returns (reads uint8[n, L], [(strand, start, stop)])."""
    rng = np.random.default_rng(seed)
    reads, info = [], []
    for i in range(n):
        h = int(rng.integers(1, 30))
        if i % 2 == 0:
            r, a, b = np.concatenate([ACGT[rng.integers(0, 4, h)], ref[:L - h]]), 0, L - h - 1
        else:
            r, a, b = np.concatenate([ref[len(ref) - (L - h):], ACGT[rng.integers(0, 4, h)]]), len(ref) - (L - h), len(ref) - 1
        r = r.copy()
        r[rng.integers(0, L, 2)] = ord("A")
        st = 0
        if rng.random() < 0.5:
            r, st = W.revcomp_rows(r.reshape(1, -1))[0], 1
        reads.append(r)
        info.append((st, a, b))
    return np.stack(reads), info


def perturb(sites, nsites, seed, reflen):
    """Damages the top site of every list (see the module text); returns new (sites, nsites)."""
    rng = np.random.default_rng(seed)
    s, ns = sites.copy(), nsites.copy()
    for r in range(len(ns)):
        n = int(ns[r])
        if n < 1:
            continue
        mode = int(rng.integers(0, 6))
        t = s[r, 0].copy()
        if (t["perfect"] or t["ngaps"]) and mode < 4:
            continue
        if mode == 0:      # both limits shifted
            t["start"] += int(rng.integers(-40, 41))
            t["stop"] += int(rng.integers(-40, 41))
        elif mode == 1:    # cut short on the left
            t["start"] += int(rng.integers(5, 60))
        elif mode == 2:    # cut short on the right
            t["stop"] -= int(rng.integers(5, 60))
        elif mode == 3:    # far too wide
            t["start"] -= int(rng.integers(50, 300))
            t["stop"] += int(rng.integers(50, 300))
        elif mode == 4 and n < s.shape[1]:   # a second site that will outscore what the top one gets
            s[r, 1:n + 1] = s[r, 0:n].copy()
            ns[r] = n + 1
            u = s[r, 1].copy()
            u["start"] += 3
            u["stop"] += 3
            u["perfect"] = 0
            u["semiperfect"] = 0
            s[r, 1] = u
            t["slowScore"] += 1
            t["score"] += 1
        if t["stop"] <= t["start"]:
            t["stop"] = t["start"] + 20
        t["start"] = max(int(t["start"]), 0)
        t["stop"] = min(int(t["stop"]), reflen - 1)
        if mode < 4:
            t["perfect"] = 0
            t["semiperfect"] = 0
        s[r, 0] = t
    return s, ns


def plant_edge_sites(sites, nsites, first, info, score=9000):
    """Gives reads first.. the one-site lists of edge_reads."""
    for q, (st, a, b) in enumerate(info):
        r = first + q
        sites[r] = 0
        nsites[r] = 1
        t = sites[r, 0].copy()
        t["chrom"], t["strand"], t["start"], t["stop"], t["hits"] = 1, st, a, b, 5
        t["quickScore"] = t["score"] = t["slowScore"] = score
        t["match_job"] = -1
        sites[r, 0] = t
