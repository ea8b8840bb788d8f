"""Synthetic references, reads and slow-align job lists shaped like BASELINE.json's configs.

No real genome ships with the repository, so the configs are rebuilt from seeds:
 * reference: uniform ACGT of the named length with N padding at both ends
   (current/dna/FastaToChromArrays2.java:565-575 pads chromosomes the same way);
 * reads: the "mutated" mix of the reference's read simulator (current/align2/RandomReads3.java:74-79,
   sh/randomreads.sh:64-78): half the reads perfect, the rest with SNPs / short insertions /
   deletions / N calls;
 * jobs: one slow-align call per read at its true site, window = site +- SLOW_ALIGN_PADDING (4)
   (current/align2/BBMapThread.java:309), minScore = 0.56 * maxQuality (BBMap.java:45-65).
Everything is vectorised numpy so that a million reads build in seconds.
"""
import numpy as np

from .msa import JOB_DTYPE, FILL_AND_SCORE_LIMITED, DO_TRACEBACK

ECOLI_K12_LEN = 4641652
START_PAD = 8000
BASES = np.frombuffer(b"ACGT", np.uint8)


def make_reference(length, seed, pad=START_PAD, repeat_frac=0.0, families=2000, lead_n=0):
    """Uniform ACGT of `length` bases between N pads.  repeat_frac > 0 adds the repeat model of SURVEY.md 8(d): that share of
    the sequence is overwritten with copies of `families` repeat families (300-6000 bp, each copy diverged 1-15 %), so that
    k-mer list lengths are skewed like a real genome's (long lists, greedy trimming, many candidate sites)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    body = BASES[rng.integers(0, 4, size=length, dtype=np.uint8)]
    if repeat_frac > 0:
        fam_len = rng.integers(300, 6001, size=families)
        fams = [BASES[rng.integers(0, 4, size=int(n), dtype=np.uint8)] for n in fam_len]
        target, placed = int(repeat_frac * length), 0
        while placed < target:
            f = fams[int(rng.integers(0, families))]
            n = len(f)
            if n + 1 >= length:
                break
            pos = int(rng.integers(0, length - n))
            cp = f.copy()
            nm = int(n * rng.uniform(0.01, 0.15))
            if nm:
                idx = rng.integers(0, n, size=nm)
                cp[idx] = BASES[rng.integers(0, 4, size=nm, dtype=np.uint8)]
            body[pos:pos + n] = cp
            placed += n
    ref = np.full(length + 2 * pad, ord("N"), np.uint8)
    ref[pad:pad + length] = body
    if lead_n > 0:            # an undefined stretch at the chromosome's start (chr21: 6.6 Mbp of N before the first base)
        ref[pad:pad + min(lead_n, length)] = ord("N")
    return ref


def make_reads_and_jobs(ref, n_reads, read_len=150, seed=2, pad=START_PAD, align_pad=4,
                        min_ratio=0.56, perfect_frac=0.5, flags=FILL_AND_SCORE_LIMITED | DO_TRACEBACK,
                        chunk=131072, max_del=40, max_ins=12, long_del_frac=0.0, long_del=(300, 800), starts=None,
                        hard_frac=0.0, del_model="short", lo=0):
    """Returns (reads_blob uint8[n*read_len... variable], jobs structured array, truth dict).  starts: read start
    coordinates to use instead of drawing them; hard_frac: share of the reads that additionally get 8-12 % substitutions
    (mates the index probe tends to miss and the paired rescue has to find).  del_model: "short" = geometric lengths capped at
    max_del (plus long_del_frac); "randomreads" = the reference generator's own draw, 1 + min(U[0, 399], U[0, 399]) with
    maxdellen 400 (RandomReads3.makeDelsa, current/align2/RandomReads3.java:809-820; sh/randomreads.sh:64-78)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_reads
    body = len(ref) - 2 * pad
    if del_model == "randomreads":
        max_del = 400
    reach = max(max_del, long_del[1] if long_del_frac > 0 else 0)
    # lo: reads start at or after pad + lo (the reference's leading N-run: the generator draws from defined sequence)
    start = rng.integers(pad + lo, pad + body - read_len - reach - 8, size=n, dtype=np.int64)
    if starts is not None:
        start = np.clip(np.asarray(starts, np.int64), pad + lo, pad + body - read_len - reach - 9)
    imperfect = rng.random(n) >= perfect_frac
    # event draws (only applied to imperfect reads)
    n_snp = np.where(imperfect & (rng.random(n) < 0.4), rng.integers(1, 4, size=n), 0)
    has_del = imperfect & (rng.random(n) < 0.2)
    has_ins = imperfect & (rng.random(n) < 0.2) & ~has_del
    has_n = imperfect & (rng.random(n) < 0.2)
    # geometric-ish indel lengths, short ones most common
    del_len = np.where(has_del, np.minimum(max_del, rng.geometric(0.25, size=n)), 0).astype(np.int64)
    if del_model == "randomreads":
        del_len = np.where(has_del, 1 + np.minimum(rng.integers(0, 400, size=n), rng.integers(0, 400, size=n)), 0).astype(np.int64)
    if long_del_frac > 0:                        # long deletions (sh/randomreads.sh maxdellen): the probe reports gap arrays
        is_long = has_del & (rng.random(n) < long_del_frac)
        del_len = np.where(is_long, rng.integers(long_del[0], long_del[1] + 1, size=n), del_len).astype(np.int64)
    ins_len = np.where(has_ins, np.minimum(max_ins, rng.geometric(0.4, size=n)), 0).astype(np.int64)
    ev_pos = rng.integers(10, read_len - 10 - max_ins, size=n, dtype=np.int64)

    reads = np.empty((n, read_len), np.uint8)
    ar = np.arange(read_len, dtype=np.int64)[None, :]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        s = start[lo:hi, None]
        p = ev_pos[lo:hi, None]
        d = del_len[lo:hi, None]
        il = ins_len[lo:hi, None]
        # read index i -> reference offset: deletion skips d bases at p, insertion stalls for il bases at p
        after = ar >= p
        shift = np.where(after, d, 0) - np.where(ar >= p + il, il, np.where(after, ar - p, 0))
        idx = s + ar + shift
        block = ref[idx]
        ins_mask = after & (ar < p + il)
        nins = int(ins_mask.sum())
        if nins:
            block[ins_mask] = BASES[rng.integers(0, 4, size=nins, dtype=np.uint8)]
        # SNPs: up to 3 per read at random positions, always a different base
        m = hi - lo
        for k in range(3):
            rows = np.nonzero(n_snp[lo:hi] > k)[0]
            if rows.size:
                cols = rng.integers(0, read_len, size=rows.size)
                old = block[rows, cols]
                new = BASES[(np.searchsorted(BASES, old) + rng.integers(1, 4, size=rows.size)) % 4]
                new = np.where(old == ord("N"), old, new)
                block[rows, cols] = new
        rows = np.nonzero(has_n[lo:hi])[0]
        if rows.size:
            block[rows, rng.integers(0, read_len, size=rows.size)] = ord("N")
        if hard_frac > 0:
            rows = np.nonzero(rng.random(m) < hard_frac)[0]
            if rows.size:
                nsub = rng.integers(int(0.08 * read_len), int(0.12 * read_len) + 1, size=rows.size)
                for rr_, k_ in zip(rows, nsub):
                    cols = rng.choice(read_len, size=int(k_), replace=False)
                    old = block[rr_, cols]
                    new = BASES[(np.searchsorted(BASES, old) + rng.integers(1, 4, size=cols.size)) % 4]
                    block[rr_, cols] = np.where(old == ord("N"), old, new)
        reads[lo:hi] = block
        del idx, block, shift, after, ins_mask

    span = read_len + del_len - ins_len            # reference bases covered by the read
    jobs = np.zeros(n, JOB_DTYPE)
    jobs["read_off"] = np.arange(n, dtype=np.int64) * read_len
    jobs["ref_off"] = 0
    jobs["read_len"] = read_len
    jobs["ref_len"] = len(ref)
    jobs["refStartLoc"] = start - align_pad
    jobs["refEndLoc"] = start + span - 1 + align_pad
    max_q = 70 + 100 * (read_len - 1)
    jobs["minScore"] = int(min_ratio * max_q)
    jobs["flags"] = flags
    truth = {"start": start, "span": span, "imperfect": imperfect}
    return reads.reshape(-1), jobs, truth


_COMP = np.full(256, 255, np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def revcomp_rows(reads2d):
    """Reverse complement of every row (AminoAcid.reverseComplementBases for the bytes A, C, G, T, N)."""
    return _COMP[reads2d[:, ::-1]]


def make_pairs(ref, n_pairs, read_len=150, seed=3, pad=START_PAD, hard_frac=0.03, middle=(-100, 100), del_model="short",
               perfect_frac=0.5, lo=0):
    """Synthetic read pairs as randomreads.sh makes them (current/align2/RandomReads3.java:1726-1727 mateMiddleMin/Max = -100/100,
    mates on opposite strands): the unsequenced middle between the mates is triangular on [-100, 100] (negative = the mates
    overlap), each mate carries the mutated mix of make_reads_and_jobs, a share hard_frac of the mates is additionally riddled
    with substitutions, and half of the fragments come from the minus strand (read 1 and read 2 swap roles).
    Returns (reads uint8[2 * n_pairs * read_len], mates INTERLEAVED: read 2p is mate 1 of pair p, read 2p+1 its mate 2,
             truth dict: start1, start2 = leftmost reference coordinate of each mate's alignment, strand1, strand2)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    body = len(ref) - 2 * pad
    mid = np.rint(rng.triangular(middle[0], 0.5 * (middle[0] + middle[1]), middle[1], size=n_pairs)).astype(np.int64)
    left = rng.integers(pad + lo, pad + body - 2 * read_len - middle[1] - 64 - (400 if del_model == "randomreads" else 0), size=n_pairs, dtype=np.int64)
    right = left + read_len + mid
    ra, _, ta = make_reads_and_jobs(ref, n_pairs, read_len=read_len, seed=seed * 7919 + 1, pad=pad, starts=left, hard_frac=hard_frac, del_model=del_model, perfect_frac=perfect_frac, lo=lo)
    rb, _, tb = make_reads_and_jobs(ref, n_pairs, read_len=read_len, seed=seed * 7919 + 2, pad=pad, starts=right, hard_frac=hard_frac, del_model=del_model, perfect_frac=perfect_frac, lo=lo)
    ra = ra.reshape(n_pairs, read_len)
    rb = revcomp_rows(rb.reshape(n_pairs, read_len))              # the right-hand mate is read from the other strand
    flip = rng.random(n_pairs) < 0.5                               # fragment from the minus strand: mate 1 is the right-hand one
    m1 = np.where(flip[:, None], rb, ra)
    m2 = np.where(flip[:, None], ra, rb)
    reads = np.empty((2 * n_pairs, read_len), np.uint8)
    reads[0::2] = m1
    reads[1::2] = m2
    truth = {"start1": np.where(flip, tb["start"], ta["start"]), "start2": np.where(flip, ta["start"], tb["start"]),
             "strand1": flip.astype(np.int32), "strand2": (~flip).astype(np.int32)}
    return reads.reshape(-1), truth


def algorithmic_bytes(jobs):
    """SURVEY.md 8(d), fused-traceback DP: per job rows + cols + 20 + 8 + 32 + (rows + cols - 1)."""
    rows = jobs["read_len"].astype(np.int64)
    cols = (jobs["refEndLoc"].astype(np.int64) - jobs["refStartLoc"].astype(np.int64) + 1)
    return int((2 * (rows + cols) + 59).sum())


def make_offsets(readlen, blocksize, density, min_keys=2):
    """KeyRing.makeOffsets(readlen, blocksize, density, minKeysDesired) (current/align2/KeyRing.java:255-297,
    makeOffsetsWithNumberOfKeys :186-229): evenly spaced key offsets.  Host-side input generation: in the
    reference this (and the quality-driven makeOffsets3) runs in Java before findAdvanced is called."""
    if readlen < blocksize:
        return []
    slots = readlen - blocksize + 1
    desired = int(np.ceil(np.float64(np.float32(readlen) * np.float32(density) / np.float32(blocksize))))
    desired = min(slots, max(min_keys, desired))
    if slots == 1 or desired == 1:
        return [slots // 2]
    if slots == 2 or desired == 2:
        return [0, slots - 1]
    if slots == 3 or desired == 3:
        return [0, slots // 2, slots - 1]
    midslots = slots - 2
    middles = min(min(desired, slots) - 2, midslots)
    fsp = max(np.float32(1.0), np.float32(midslots) / np.float32(middles + 1.0))
    offs = [0] * (middles + 2)
    offs[-1] = slots - 1
    for i in range(1, middles + 1):
        offs[i] = int(np.floor(np.float32(fsp * np.float32(i)) + np.float32(0.5)))
    if middles > 2:
        offs[1] = int(fsp)
        offs[middles] = int(np.ceil(np.float64(np.float32(fsp * np.float32(middles)))))
    return offs


def make_pacbio_pieces(chroms, n, seed=5, min_len=6000, max_len=6000, err=(0.13, 0.17), pad=START_PAD, junk_frac=0.0):
    """Pieces of PacBio-like reads as mapPacBio sees them (BASELINE.json configs[4]: 10 kb reads cut at fastareadlen = 6000,
    current/align2/BBMapPacBio.java; error model pbmin 0.13 / pbmax 0.17, current/align2/RandomReads3.java:1714-1715, split into
    deletions / substitutions / insertions 35 : 20 : 45): `n` pieces drawn from `chroms` (padded uint8 arrays), either strand.
    Returns (list of uint8 arrays, truth array of (chrom 1-based, strand, start, stop))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    comp = np.full(256, ord("N"), np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    pieces, truth = [], np.zeros((n, 4), np.int64)
    for i in range(n):
        ci = int(rng.integers(0, len(chroms)))
        G = chroms[ci]
        L = int(rng.integers(min_len, max_len + 1))
        span = L + L // 4 + 64
        s_ = int(rng.integers(pad + 100, len(G) - pad - span - 100))
        e = rng.uniform(*err)
        src = G[s_:s_ + span]
        x = rng.random(len(src))
        keep = x >= e * 0.35
        sub = (x >= e * 0.35) & (x < e * 0.55)
        piece = src.copy()
        piece[sub] = BASES[rng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]
        consumed = np.cumsum(np.ones(len(src), np.int64))          # reference bases used up to each kept base
        piece, consumed = piece[keep], consumed[keep]
        ins_at = np.nonzero(rng.random(len(piece)) < e * 0.45)[0]
        piece = np.insert(piece, ins_at, BASES[rng.integers(0, 4, size=len(ins_at), dtype=np.uint8)])
        consumed = np.insert(consumed, ins_at, consumed[np.minimum(ins_at, len(consumed) - 1)])
        piece, stop = piece[:L], s_ + int(consumed[min(L, len(consumed)) - 1]) - 1
        if rng.random() < junk_frac:
            piece = BASES[rng.integers(0, 4, size=L, dtype=np.uint8)]
        strand = int(rng.integers(0, 2))
        if strand:
            piece = comp[piece[::-1]]
        pieces.append(np.ascontiguousarray(piece))
        truth[i] = (ci + 1, strand, s_, stop)
    return pieces, truth
