"""DP microbenchmark for BASELINE.json configs[4] (mapPacBio: 10 kb reads pre-split into <= 6,000-base pieces, fastareadlen=6000,
current/align2/BBMapPacBio.java:47-69): N pieces with 13-17 % PacBio errors (pbmin/pbmax, current/align2/RandomReads3.java:1714-1715),
each aligned with the MultiStateAligner9PacBio scheme against its window +- padding (fillAndScoreLimited + traceback) by the
strip-tiled wavefront kernel.  Prints one JSON line: pieces/s, visited cells/s.  (The PacBio INDEX probe -- BBIndexPacBio's
constants -- is not built, so this is the DP half of that configuration only.)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from bbmap_amd import msa as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rng = np.random.Generator(np.random.PCG64(5))
BASES = np.frombuffer(b"ACGT", np.uint8)
genome = BASES[rng.integers(0, 4, size=4_000_000, dtype=np.uint8)]
reads, jobs = [], np.zeros(n, M.JOB_DTYPE)
off = 0
for i in range(n):
    s = int(rng.integers(1000, len(genome) - L - 2000))
    src = genome[s:s + L]
    err = rng.uniform(0.13, 0.17)
    x = rng.random(L)
    keep = x >= err * 0.35                                   # deletions
    sub = (x >= err * 0.35) & (x < err * 0.55)
    piece = src.copy()
    piece[sub] = BASES[rng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]
    piece = piece[keep]
    ins_at = np.nonzero(rng.random(len(piece)) < err * 0.45)[0]
    piece = np.insert(piece, ins_at, BASES[rng.integers(0, 4, size=len(ins_at), dtype=np.uint8)])[:6019]
    reads.append(piece)
    jobs[i] = (off, 0, len(piece), len(genome), s - 40, s + L + 40, int(0.3 * (90 + 100 * (len(piece) - 1))),
               M.FILL_AND_SCORE_LIMITED | M.DO_TRACEBACK)
    off += len(piece)
blob = np.concatenate(reads)
ctx = M.MSAContext(maxRows=6019, maxColumns=7600, scheme=M.SCHEME_9PACBIO)
stride = 6019 + 7600 + 64
t = time.time()
res, match = ctx.align_batch(jobs, blob, genome, match_stride=stride)
dt1 = time.time() - t
t = time.time()
res, match = ctx.align_batch(jobs, blob, genome, match_stride=stride)
dt = time.time() - t
k3 = ctx.last_kernel_ms3()
ok = int((res["score_len"] > 0).sum())
cells = int(res["iterations"].sum())
print(json.dumps({"metric": "pacbio_dp_pieces_per_sec", "value": n / (k3[1] * 1e-3), "unit": "pieces/s", "pieces": n, "piece_len": L,
                  "strip_kernel_ms": k3[1], "generic_kernel_ms": k3[2], "host_call_s_incl_copies": dt, "first_call_s": dt1,
                  "aligned": ok, "visited_cells": cells, "gcups_visited": cells / (k3[1] * 1e-3) / 1e9,
                  "gcups_swept": float((jobs["read_len"].astype(np.int64) * (jobs["refEndLoc"] - jobs["refStartLoc"] + 1)).sum()) / (k3[1] * 1e-3) / 1e9}))
