"""The one data set the reference itself holds for this path (resources/contents.txt:27-29): the PhiX174 reference and 100
synthetic read pairs made from it by the reference's own read generator, each named `id_chrom_strand_start_stop_origStart_...`
with its true alignment in the padded chromosome coordinates BBMap uses (parser: current/stream/FASTQ.java:590-603;
chromosome array = 8000 N + sequence + 8000 N, current/dna/FastaToChromArrays2.java:565-575).  Copied byte for byte into
tests/golden/ (data, not source)."""
import gzip
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
START_PAD = 8000


def phix_reference():
    seq = bytearray()
    with gzip.open(os.path.join(HERE, "phix174_ill.ref.fa.gz"), "rt") as f:
        for line in f:
            if not line.startswith(">"):
                seq += line.strip().upper().encode()
    body = np.frombuffer(bytes(seq), np.uint8)
    ok = np.isin(body, np.frombuffer(b"ACGT", np.uint8))
    body = np.where(ok, body, ord("N")).astype(np.uint8)            # ChromosomeArray.set: degenerate bases become N
    ref = np.full(len(body) + 2 * START_PAD, ord("N"), np.uint8)
    ref[START_PAD:START_PAD + len(body)] = body
    return ref


def sample_reads(which):
    """Returns (reads uint8[n, 100], truth dict of int arrays: strand, start, stop) for sample1 / sample2."""
    bases, strand, start, stop = [], [], [], []
    with gzip.open(os.path.join(HERE, "sample%d.fq.gz" % which), "rt") as f:
        lines = [ln.rstrip("\n") for ln in f]
    for i in range(0, len(lines), 4):
        name = lines[i][1:].split("_")
        strand.append(int(name[2]))
        start.append(int(name[3]))
        stop.append(int(name[4]))
        bases.append(np.frombuffer(lines[i + 1].upper().encode(), np.uint8))
    lens = {len(b) for b in bases}
    assert lens == {100}, lens
    return np.stack(bases), dict(strand=np.array(strand), start=np.array(start), stop=np.array(stop))
