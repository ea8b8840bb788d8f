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


def fix_qualities(bases, qual):
    """What stream.Read does to a FASTQ record's qualities on input (CHANGE_QUALITY, current/stream/Read.java:164-177): a called
    base's quality is capped to [MIN_CALLED_QUALITY 2, MAX_CALLED_QUALITY 41], an undefined base's becomes 0."""
    b = np.asarray(bases, np.uint8)
    q = np.asarray(qual, np.uint8).copy()
    defined = np.isin(b & ~np.uint8(32), np.frombuffer(b"ACGTU", np.uint8))
    q[defined] = np.clip(q[defined], 2, 41)
    q[~defined] = 0
    return q


def sample_qualities(which):
    """Numeric phred qualities (ASCII - 33) of sample1 / sample2, as stream.Read holds them after input: uint8[n, 100]."""
    with gzip.open(os.path.join(HERE, "sample%d.fq.gz" % which), "rt") as f:
        lines = [ln.rstrip("\n") for ln in f]
    out = []
    for i in range(0, len(lines), 4):
        b = np.frombuffer(lines[i + 1].upper().encode(), np.uint8)
        q = np.frombuffer(lines[i + 3].encode(), np.uint8) - 33
        out.append(fix_qualities(b, q))
    return np.stack(out)


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


def _revcomp(b):
    comp = np.full(256, ord("N"), np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    return comp[np.asarray(b, np.uint8)[::-1]]


def fixture_inputs(mode, use_qualities):
    """(reads list, qualities list or None, paired) of one run: mode "se1" / "se2" = sample1 / sample2 single-ended, "pe" = the pairs."""
    r1, _ = sample_reads(1)
    r2, _ = sample_reads(2)
    q1, q2 = sample_qualities(1), sample_qualities(2)
    if mode == "pe":
        reads = [x for p in zip(r1, r2) for x in p]
        quals = [x for p in zip(q1, q2) for x in p]
    else:
        reads, quals = (list(r1), list(q1)) if mode == "se1" else (list(r2), list(q2))
    return reads, (quals if use_qualities else None), mode == "pe"


def fixture_runs():
    """name -> {"inputs": (recs, blob, baseScores, keyinfo, paired), "oracle": callable} for the six runs of the fixture.  Keys are
    placed by the product's host code (bbkeys_make_batch = AbstractMapThread.quickMap's key stage) with bbmap.sh's densities."""
    from bbmap_amd import keys as K
    from oracle import oracle as O
    ref = phix_reference()
    runs = {}
    for mode in ("se1", "se2", "pe"):
        for use_q in (False, True):
            reads, quals, paired = fixture_inputs(mode, use_q)
            recs, blob, bs, ki = K.make_batch(reads, quals)

            def oracle(recs=recs, blob=blob, bs=bs, ki=ki, paired=paired, final_stage=1):
                oi = O.OracleIndex([ref], k=13)
                oi.s.p.quitAfterTwoPerfects = 0 if paired else 1            # BBMap.java:434
                return O.map_reads(oi, recs, blob, ki, base_scores=bs, paired=paired, cap=64, params=O.map_default_params(finalStage=final_stage))
            runs["%s_%s" % (mode, "qual" if use_q else "noqual")] = {"inputs": (recs, blob, bs, ki, paired), "oracle": oracle}
    return runs


def truth_window_jobs(which):
    """Per read of sample `which`: (bases on the truth strand, refStartLoc, refEndLoc, minScore) of the fill scoreSlow would issue for a
    site at exactly the coordinates in the read's name: window +- SLOW_ALIGN_PADDING 4, minScore = the initial minMsaLimit
    (-CLEARZONE1e + (int)(0.56 * maxSwScore), current/align2/BBMapThread.java:262-264)."""
    reads, truth = sample_reads(which)
    max_sw = 70 + 99 * 100
    floor = -258 + int(np.float32(0.56) * np.float32(max_sw))
    jobs = []
    for i in range(len(reads)):
        rd = _revcomp(reads[i]) if truth["strand"][i] else reads[i]
        jobs.append((rd.tobytes(), int(truth["start"][i]) - 4, int(truth["stop"][i]) + 4, floor))
    return jobs


def truth_window_scores(which):
    """score2's {score, start, stop} of each truth-window fill on the CPU oracle (None where fillAndScoreLimited returns null)."""
    from oracle import oracle as O
    ref = phix_reference().tobytes()
    msa = O.OracleMSA(601, 3000)
    out = []
    for rd, a, b, floor in truth_window_jobs(which):
        sv, _ = msa.fillAndScoreLimited(rd, ref, a, b, floor)
        out.append(None if sv is None else [int(sv[0]), int(sv[1]), int(sv[2])])
    return out


# ---------------------------------------------------------------------------------------------- the same fixture through mapPacBio's classes
PACBIO_MSA = dict(msaMaxRows=160, msaMaxColumns=7600)         # BBMapThreadPacBio's 7600 columns; rows cut to the fixture's reads (memory)


def fixture_runs_pacbio():
    """The fixture's reads mapped as mapPacBio.sh would map them (100-base reads are legal input to BBMapPacBio): BBIndexPacBio's
    constants (k 12, key density 3.5 / 2.8 / 4.5), BBMapThreadPacBio's single-ended flow, MultiStateAligner9PacBio.  Four runs:
    sample1 / sample2, keys placed from the qualities or as for quality-less input.  name -> {"inputs", "oracle"}."""
    from bbmap_amd import keys as K
    from oracle import oracle as O
    ref = phix_reference()
    cfg = K.default_config(K.PROFILE_PACBIO)
    runs = {}
    for mode in ("se1", "se2"):
        for use_q in (False, True):
            reads, quals, _ = fixture_inputs(mode, use_q)
            recs, blob, bs, ki = K.make_batch(reads, quals, cfg)

            def oracle(recs=recs, blob=blob, bs=bs, ki=ki):
                oi = O.OracleIndex([ref], profile="pacbio")
                return O.map_reads(oi, recs, blob, ki, base_scores=bs, paired=False, cap=64, params=O.map_default_params("pacbio", **PACBIO_MSA))
            runs["%s_%s" % (mode, "qual" if use_q else "noqual")] = {"inputs": (recs, blob, bs, ki, False), "oracle": oracle}
    return runs


def truth_window_jobs_pacbio(which):
    """As truth_window_jobs with BBMapThreadPacBio's numbers: SLOW_ALIGN_PADDING 8, minMsaLimit = -CLEARZONE1e + (int)(0.46 * maxSwScore)
    with CLEARZONE1e = 2 * 100 - 90 + 137 + 1 = 248 and maxSwScore = 90 + 99 * 100 (MultiStateAligner9PacBio.java:2377-2380)."""
    reads, truth = sample_reads(which)
    max_sw = 90 + 99 * 100
    floor = -248 + int(np.float32(0.46) * np.float32(max_sw))
    jobs = []
    for i in range(len(reads)):
        rd = _revcomp(reads[i]) if truth["strand"][i] else reads[i]
        jobs.append((rd.tobytes(), int(truth["start"][i]) - 8, int(truth["stop"][i]) + 8, floor))
    return jobs


def truth_window_scores_pacbio(which):
    from oracle import oracle as O
    ref = phix_reference().tobytes()
    msa = O.OracleMSA(160, 7600, scheme="9pacbio")
    out = []
    for rd, a, b, floor in truth_window_jobs_pacbio(which):
        sv, _ = msa.fillAndScoreLimited(rd, ref, a, b, floor)
        out.append(None if sv is None else [int(sv[0]), int(sv[1]), int(sv[2])])
    return out
